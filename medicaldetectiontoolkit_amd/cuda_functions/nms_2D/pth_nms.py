"""Drop-in for the reference's cuda_functions/nms_2D/pth_nms.py (nms_gpu :5-17, nms_cpu :20-37).

dets: [N, 5] = (y1, x1, y2, x2, score) in pixel coordinates.
Returns a 1-D int64 tensor of indices into the rows of `dets`, best score first.
"""
from .._nms_impl import nms_cpu as _nms_cpu
from .._nms_impl import nms_gpu as _nms_gpu


def nms_gpu(dets, thresh):
    return _nms_gpu(dets, thresh, 2)


def nms_cpu(dets, thresh):
    return _nms_cpu(dets, thresh, 2)
