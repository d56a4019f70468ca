"""The 3D conv backbone rides MIOpen (north star).  Its solver search ("find") for this model's odd channel
counts takes ~2 minutes per fresh process, and without it the immediate-mode heuristics fall back to naive
kernels (42x slower).  setup() points MIOpen's user find-db and compiled-kernel cache at a directory inside
the repository so that the result of one search travels with the tree (e.g. onto a fresh GPU box)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CACHE_DIR = os.path.join(HERE, "miopen_cache")


def _tree_hash(d):
    """identity of the committed cache: names, sizes and mtimes-independent content hash of the find-db text files (the kernel
    cache is derived data)"""
    import hashlib
    h = hashlib.sha256()
    db = os.path.join(d, "db")
    for name in sorted(os.listdir(db)) if os.path.isdir(db) else []:
        h.update(name.encode())
        try:
            with open(os.path.join(db, name), "rb") as f:
                h.update(f.read())
        except OSError:
            pass
    return h.hexdigest()[:12]


def _sweep_stale_pid_copies():
    """round-3 scheme left one /tmp/mdt_miopen_cache_u<uid>_rank<k>_pid<pid> per process; ranks torn down with SIGTERM never ran
    their atexit handler.  Remove the copies whose process is gone."""
    import re
    import shutil
    import tempfile
    tmp = tempfile.gettempdir()
    try:
        names = os.listdir(tmp)
    except OSError:
        return
    for name in names:
        m = re.match(r"mdt_miopen_cache_u%d_rank\w+_pid(\d+)$" % os.getuid(), name)
        if m and not os.path.exists("/proc/%s" % m.group(1)):
            shutil.rmtree(os.path.join(tmp, name), ignore_errors=True)


def _overlay_root():
    for base in (os.environ.get("XDG_CACHE_HOME"), os.path.join(os.path.expanduser("~"), ".cache")):
        if base:
            try:
                os.makedirs(os.path.join(base, "mdt_miopen"), exist_ok=True)
                if os.access(os.path.join(base, "mdt_miopen"), os.W_OK):
                    return os.path.join(base, "mdt_miopen")
            except OSError:
                pass
    import tempfile
    return os.path.join(tempfile.gettempdir(), "mdt_miopen_u%d" % os.getuid())


def setup(cache_dir=None):
    d = cache_dir or os.environ.get("MDT_MIOPEN_CACHE", CACHE_DIR)
    if cache_dir is None and "MDT_MIOPEN_CACHE" not in os.environ and not os.environ.get("MDT_MIOPEN_CACHE_INPLACE"):
        # The committed cache is never written (a run does not dirty the tree) and N ranks never share files: each rank works on
        # a PERSISTENT per-user overlay  <cache root>/mdt_miopen/<hash of the committed find-db>/rank<k>, seeded from the tree by
        # copy-then-rename (an interrupted copy is never taken for a cache).  Persistent, because a find result for a shape the
        # committed db lacks (a user's own patch size, a new layer) must survive the process -- the round-3 per-pid copies were
        # deleted at exit, so every start repeated the ~1 min find; keyed by the tree's hash, so a refreshed
        # committed cache is never shadowed by a stale overlay.  MIOpen's own file locking covers two processes of one rank
        # (pytest + bench).  MDT_MIOPEN_CACHE_INPLACE=1 writes into the tree (to refresh the committed cache after a new find).
        import shutil
        import tempfile
        rank = os.environ.get("RANK", "0")
        _sweep_stale_pid_copies()
        try:
            if os.path.isdir(d):
                root = os.path.join(_overlay_root(), _tree_hash(d))
                dst = os.path.join(root, "rank%s" % rank)
                if not os.path.isdir(dst):
                    os.makedirs(root, exist_ok=True)
                    scratch = tempfile.mkdtemp(prefix="seed_", dir=root)
                    shutil.copytree(d, os.path.join(scratch, "c"))
                    try:
                        os.rename(os.path.join(scratch, "c"), dst)
                    except OSError:          # another process of this rank won the race: use its copy
                        pass
                    shutil.rmtree(scratch, ignore_errors=True)
                if os.path.isdir(dst):
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
    for key, sub in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "kernels")):
        # a value inherited from a parent process that ran setup() (mp.spawn workers, bench.py's self-launched ranks) points
        # at the PARENT's private copy: replace it; a value the user exported is respected
        v = os.environ.get(key)
        if v is None or "/mdt_miopen/" in v or "/mdt_miopen_u" in v or "mdt_miopen_cache_u" in v:
            os.environ[key] = os.path.join(d, sub)
    return d
