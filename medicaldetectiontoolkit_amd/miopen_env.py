"""The 3D conv backbone rides MIOpen (north star).  Its solver search ("find") for this model's odd channel
counts takes ~2 minutes per fresh process, and without it the immediate-mode heuristics fall back to naive
kernels (42x slower).  setup() points MIOpen's user find-db and compiled-kernel cache at a directory inside
the repository so that the result of one search travels with the tree (e.g. onto a fresh GPU box)."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CACHE_DIR = os.path.join(HERE, "miopen_cache")


def _cleanup(path, pid):
    import shutil
    if os.getpid() == pid:        # forked children (mp.spawn uses spawn, DataLoader may fork) must not delete the parent's copy
        shutil.rmtree(path, ignore_errors=True)


def setup(cache_dir=None):
    d = cache_dir or os.environ.get("MDT_MIOPEN_CACHE", CACHE_DIR)
    if cache_dir is None and "MDT_MIOPEN_CACHE" not in os.environ and not os.environ.get("MDT_MIOPEN_CACHE_INPLACE"):
        # every process works on its OWN copy of the in-tree cache (seeded from it): N ranks never write the same
        # find-db / kernel-cache files concurrently, and a run never dirties the committed files.
        # MDT_MIOPEN_CACHE_INPLACE=1 writes into the tree (to refresh the committed cache after a new find).
        # The copy is private to THIS process (pid in the name, removed at exit) and made atomically (copy to a scratch
        # name, then rename): a refreshed in-tree cache is never shadowed by a stale temp copy, an interrupted copy is
        # never taken for a valid cache, and two independent single-GPU processes (pytest + bench) never share files.
        import atexit
        import shutil
        import tempfile
        rank = os.environ.get("RANK", "0")
        dst = os.path.join(tempfile.gettempdir(), "mdt_miopen_cache_u%d_rank%s_pid%d" % (os.getuid(), rank, os.getpid()))
        try:
            if os.path.isdir(d):
                if not os.path.isdir(dst):
                    scratch = tempfile.mkdtemp(prefix="mdt_miopen_cache_tmp_")
                    shutil.copytree(d, os.path.join(scratch, "c"))
                    os.rename(os.path.join(scratch, "c"), dst)
                    shutil.rmtree(scratch, ignore_errors=True)
                    atexit.register(_cleanup, dst, os.getpid())
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
        if key not in os.environ or "mdt_miopen_cache_" in os.environ[key]:
            os.environ[key] = os.path.join(d, sub)
    return d
