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
# every replay is right (round 4: tools/graph_small_bisect.py, tools/r04_calls/r04_run_o.sh).  The runtime reads the flag once, when
# it initialises, so it is set here, at package import, and GRAPH_RUNTIME_SAFE records whether that was early enough.
GRAPH_RUNTIME_SAFE = False


def _configure_hip_runtime():
    global GRAPH_RUNTIME_SAFE
    already = False
    tc = sys.modules.get("torch")
    if tc is not None:
        try:
            already = bool(tc.cuda.is_initialized())
        except Exception:
            already = False
    if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") is None and not already:
        os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    GRAPH_RUNTIME_SAFE = os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0" and not (already and os.environ.get("MDT_PACKET_CAPTURE_SET_EARLY") != "1")
    if GRAPH_RUNTIME_SAFE:
        os.environ["MDT_PACKET_CAPTURE_SET_EARLY"] = "1"       # child processes / later imports: the flag was in place before the runtime came up


_configure_hip_runtime()


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
