"""MI355X-native hot path of MIC-DKFZ/medicaldetectiontoolkit.

Hand-written HIP kernels for gfx950 behind a C ABI (include/mdt_hip.h), bound at
the reference's own Python call sites:
    cuda_functions.nms_{2D,3D}.pth_nms.nms_gpu
    cuda_functions.roi_align_{2D,3D}.roi_align.crop_and_resize.CropAndResizeFunction
plus device versions of anchor generation / matching, box decode and weighted
box clustering (utils.model_utils, predictor).
"""
import os
import sys

__version__ = "0.1.0"

# hipGraph replays of the training step (training.GraphedTrainStep): the HIP runtime's "graph packet capture" fast path (pre-built AQL
# packets per kernel node, DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default in ROCm 7.0) replays the matching + RPN-loss part of the step
# INCORRECTLY at small sizes -- first replay right, second replay a GPU memory fault; with the flag off (or with AMD_SERIALIZE_KERNEL=3)
# every replay is right (round 4: tools/graph_small_bisect.py, tools/r04_calls/r04_run_o.sh).  The runtime reads the flag ONCE, when it
# initialises -- and torch.cuda.is_available() / device_count() already initialise it without torch.cuda.is_initialized() saying so.
# Importing this package therefore changes NOTHING in the process (round 5): the entry point that wants a graphed step puts
#     DEBUG_CLR_GRAPH_PACKET_CAPTURE=0  and  MDT_GRAPH_ENV_BEFORE_HIP=1
# into the environment BEFORE it imports torch (train.py --graph 1, bench.py and tests/conftest.py do; `graph_env_setup()` below is the
# helper), or the variable is exported in the shell.  `graph_runtime_safe()` is what GraphedTrainStep asks before it captures.


def graph_env_setup():
    """Call BEFORE `import torch` in a process that will use training.GraphedTrainStep.  Refuses (returns False, changes nothing) when
    torch is already imported: it may have initialised the HIP runtime, after which the flag is not read any more."""
    if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" and (os.environ.get("MDT_GRAPH_ENV_BEFORE_HIP") == "1" or _in_initial_environ()):
        return True
    if "torch" in sys.modules:
        return False
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    os.environ["MDT_GRAPH_ENV_BEFORE_HIP"] = "1"        # inherited by child processes, where the flag is in the initial environment anyway
    return True


def _in_initial_environ():
    """the variable was in the environment the process was STARTED with (exported in the shell / by the parent)"""
    try:
        with open("/proc/self/environ", "rb") as f:
            return b"DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" in f.read().split(b"\0")
    except OSError:
        return False


def graph_runtime_safe():
    """True iff the runtime flag was in place before the HIP runtime could have come up: exported to the process, or set by
    graph_env_setup() before torch was imported.  Never inferred from torch.cuda.is_initialized()."""
    if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0":
        return False
    return _in_initial_environ() or os.environ.get("MDT_GRAPH_ENV_BEFORE_HIP") == "1"


def install_dropin():
    """Register this package's `cuda_functions` under the top-level name the reference
    models import (models/mrcnn.py:24-27), so unmodified reference model files pick up
    the HIP kernels.  Call before importing the reference's models."""
    import importlib
    names = [
        "cuda_functions",
        "cuda_functions.nms_2D", "cuda_functions.nms_2D.pth_nms",
        "cuda_functions.nms_3D", "cuda_functions.nms_3D.pth_nms",
        "cuda_functions.roi_align_2D", "cuda_functions.roi_align_2D.roi_align",
        "cuda_functions.roi_align_2D.roi_align.crop_and_resize",
        "cuda_functions.roi_align_3D", "cuda_functions.roi_align_3D.roi_align",
        "cuda_functions.roi_align_3D.roi_align.crop_and_resize",
    ]
    for n in names:
        sys.modules[n] = importlib.import_module(__name__ + "." + n)
    return names
