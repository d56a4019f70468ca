"""TEST / BASELINE INFRASTRUCTURE: where the reference's own Python model files come from at run time.

`unpack()` returns a directory holding models/{mrcnn,retina_unet,backbone}.py, utils/{model_utils,exp_utils}.py and plotting.py of the
reference, UNMODIFIED: /root/reference itself when it exists (build container), else oracle/_ref/ref_models.tar.gz -- the archive
`make -C oracle _ref_py` packs from the reference where it lies (git-ignored, shipped to the GPU box with the snapshot) -- unpacked into a
fresh temporary directory.  No reference source file lives in this repository's tree."""
import atexit
import os
import shutil
import tarfile
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ARCHIVE = os.path.join(HERE, "_ref", "ref_models.tar.gz")
FILES = ("models/mrcnn.py", "models/retina_unet.py", "models/backbone.py", "utils/model_utils.py", "utils/exp_utils.py", "plotting.py")
_dir = None


def available():
    return os.path.exists(ARCHIVE) or all(os.path.exists(os.path.join("/root/reference", f)) for f in FILES)


def unpack(prefer_archive=False):
    """directory with the reference files (cached per process); raises FileNotFoundError when neither source exists"""
    global _dir
    if _dir is not None:
        return _dir
    if not prefer_archive and all(os.path.exists(os.path.join("/root/reference", f)) for f in FILES):
        _dir = "/root/reference"
        return _dir
    if not os.path.exists(ARCHIVE):
        raise FileNotFoundError("oracle/_ref/ref_models.tar.gz is missing: run `python -c 'import __graft_entry__ as g; g.build()'` in the build "
                                "container (make -C oracle _ref_py); it ships to the GPU box with the snapshot")
    d = tempfile.mkdtemp(prefix="mdt_ref_models_")
    atexit.register(shutil.rmtree, d, True)
    with tarfile.open(ARCHIVE) as t:
        members = [m for m in t.getmembers() if m.name in FILES]
        t.extractall(d, members=members)
    _dir = d
    return _dir
