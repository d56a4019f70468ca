"""MI355X-native hot path of MIC-DKFZ/medicaldetectiontoolkit.

Hand-written HIP kernels for gfx950 behind a C ABI (include/mdt_hip.h), bound at
the reference's own Python call sites:
    cuda_functions.nms_{2D,3D}.pth_nms.nms_gpu
    cuda_functions.roi_align_{2D,3D}.roi_align.crop_and_resize.CropAndResizeFunction
plus device versions of anchor generation / matching, box decode and weighted
box clustering (utils.model_utils, predictor).
"""
import sys

__version__ = "0.1.0"


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
