import os
import sys

# before anything initialises the HIP runtime (the `cuda` fixture's torch.cuda.is_available() does): the runtime flag hipGraph replays of
# the training step need (see medicaldetectiontoolkit_amd/__init__.py: importing the package sets NOTHING, the entry point does)
if "torch" not in sys.modules and os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") is None:
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    os.environ["MDT_GRAPH_ENV_BEFORE_HIP"] = "1"

# MIOpen's exhaustive find skips the naive direct solvers (never the fastest for these layers, seconds per trial on 128^3 maps: the Retina
# U-Net step at the benchmarked size took 7 min of find with them, ~1 min without) -- what bench.py sets for its child processes
os.environ.setdefault("MDT_MIOPEN_SKIP_NAIVE", "1")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    # MIOpen reads its find-db / kernel-cache paths and solver switches ONCE, at its first convolution: point it at the repository's cache before
    # any test runs one (a test module that convolved first used to leave every later `miopen_env.setup()` without effect -- the Retina U-Net
    # step at the benchmarked size then spent 7.5 min in an uncached exhaustive find)
    from medicaldetectiontoolkit_amd import miopen_env
    miopen_env.setup()
    return torch.device("cuda:0")
