"""The 3D conv backbone rides MIOpen (north star).  Its solver search ("find") for this model's odd channel
counts takes ~2 minutes per fresh process, and without it the immediate-mode heuristics fall back to naive
kernels (42x slower).  setup() points MIOpen's user find-db and compiled-kernel cache at a directory inside
the repository so that the result of one search travels with the tree (e.g. onto a fresh GPU box)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CACHE_DIR = os.path.join(HERE, "miopen_cache")


def setup(cache_dir=None):
    d = cache_dir or os.environ.get("MDT_MIOPEN_CACHE", CACHE_DIR)
    if cache_dir is None and "MDT_MIOPEN_CACHE" not in os.environ and not os.environ.get("MDT_MIOPEN_CACHE_INPLACE"):
        # every process works on its OWN copy of the in-tree cache (seeded from it): N ranks never write the same
        # find-db / kernel-cache files concurrently, and a run never dirties the committed files.
        # MDT_MIOPEN_CACHE_INPLACE=1 writes into the tree (to refresh the committed cache after a new find).
        import shutil
        import tempfile
        rank = os.environ.get("RANK", "0")
        dst = os.path.join(tempfile.gettempdir(), "mdt_miopen_cache_rank%s_%d" % (rank, os.getuid()))
        try:
            if os.path.isdir(d) and not os.path.isdir(dst):
                shutil.copytree(d, dst)
            d = dst
        except OSError:
            pass
    try:
        os.makedirs(os.path.join(d, "db"), exist_ok=True)
        os.makedirs(os.path.join(d, "kernels"), exist_ok=True)
    except OSError:
        return None
    if os.environ.get("MDT_MIOPEN_SKIP_NAIVE"):
        # leave the naive direct solvers out of the find: they are never the fastest for these shapes but take seconds
        # per trial on 128^3 maps (Retina U-Net at batch 8: find > 20 min with them, ~1 min without)
        for k in ("FWD", "BWD", "WRW"):
            os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + k, "0")
    os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(d, "db"))
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(d, "kernels"))
    return d
