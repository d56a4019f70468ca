"""numpy/ctypes front end of the CPU oracle (oracle/mdt_oracle.c) and of the
compiled-where-it-lies reference pieces under oracle/_ref/.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _build():
    subprocess.check_call(["make", "-C", _HERE, "libmdt_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmdt_oracle.so")
        if not os.path.exists(path):
            _build()
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_sample_coord.restype = c_float
        _LIB.oracle_sample_coord.argtypes = [c_float, c_float, c_int, c_int, c_int]
        _LIB.oracle_iou_3d.restype = c_float
        _LIB.oracle_iou_2d.restype = c_float
    return _LIB


def _p(a):
    return a.ctypes.data_as(c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def crop_and_resize_forward(image, boxes, box_ind, crop):
    """image [B,C,Y,X(,Z)] f32, boxes [N,2*dim] normalised, box_ind [N] -> crops [N,C,*crop]."""
    image, boxes, box_ind = _f32(image), _f32(boxes), _i32(box_ind)
    dim = image.ndim - 2
    n, B, C = boxes.shape[0], image.shape[0], image.shape[1]
    crops = np.empty((n, C) + tuple(crop), dtype=np.float32)
    if dim == 3:
        lib().oracle_crop_and_resize_3d_forward(
            _p(image), _p(boxes), _p(box_ind), n, B, image.shape[2], image.shape[3], image.shape[4],
            crop[0], crop[1], crop[2], C, c_float(0.0), _p(crops))
    else:
        lib().oracle_crop_and_resize_2d_forward(
            _p(image), _p(boxes), _p(box_ind), n, B, image.shape[2], image.shape[3],
            crop[0], crop[1], C, c_float(0.0), _p(crops))
    return crops


def crop_and_resize_backward(grads, boxes, box_ind, im_size):
    grads, boxes, box_ind = _f32(grads), _f32(boxes), _i32(box_ind)
    dim = len(im_size) - 2
    n = grads.shape[0]
    out = np.empty(tuple(im_size), dtype=np.float32)
    if dim == 3:
        lib().oracle_crop_and_resize_3d_backward(
            _p(grads), _p(boxes), _p(box_ind), n, im_size[0], im_size[2], im_size[3], im_size[4],
            grads.shape[2], grads.shape[3], grads.shape[4], im_size[1], _p(out))
    else:
        lib().oracle_crop_and_resize_2d_backward(
            _p(grads), _p(boxes), _p(box_ind), n, im_size[0], im_size[2], im_size[3],
            grads.shape[2], grads.shape[3], im_size[1], _p(out))
    return out


def sort_order(scores):
    """descending, ties -> lower index first (the product's documented tie rule)."""
    return np.argsort(-np.asarray(scores, dtype=np.float64), kind="stable")


def gpu_nms(dets, thresh, strict_gt=True):
    """The reference GPU algorithm (mask + greedy scan) restated on the CPU.
    dets [N, 5|7] in original order; returns indices into dets, best first (pth_nms.py:5-17)."""
    dets = _f32(dets)
    n = dets.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    order = sort_order(dets[:, -1])
    ds = np.ascontiguousarray(dets[order])
    keep = np.empty(n, dtype=np.int64)
    num = c_int64(0)
    fn = lib().oracle_gpu_nms_3d if dets.shape[1] == 7 else lib().oracle_gpu_nms_2d
    rc = fn(_p(ds), n, c_float(thresh), 1 if strict_gt else 0, _p(keep), ctypes.byref(num))
    assert rc == 0
    return order[keep[:num.value]]


def nms_mask(dets_sorted, thresh):
    ds = _f32(dets_sorted)
    n = ds.shape[0]
    cb = (n + 63) // 64
    mask = np.zeros((n, cb), dtype=np.uint64)
    fn = lib().oracle_nms_mask_3d if ds.shape[1] == 7 else lib().oracle_nms_mask_2d
    fn(_p(ds), n, c_float(thresh), _p(mask))
    return mask


def _areas(dets):
    d = dets.astype(np.float32)
    a = (d[:, 2] - d[:, 0] + np.float32(1)) * (d[:, 3] - d[:, 1] + np.float32(1))
    if d.shape[1] == 7:
        a = a * (d[:, 5] - d[:, 4] + np.float32(1))
    return a.astype(np.float32)


def cpu_nms(dets, thresh):
    """cpu_nms restated (>= rule, nms.c:35-70) incl. the wrapper's areas/order (pth_nms.py:20-37)."""
    dets = _f32(dets)
    n = dets.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    order = np.ascontiguousarray(sort_order(dets[:, -1]), dtype=np.int64)
    areas = _areas(dets)
    keep = np.empty(n, dtype=np.int64)
    num = c_int64(0)
    fn = lib().oracle_cpu_nms_3d if dets.shape[1] == 7 else lib().oracle_cpu_nms_2d
    fn(_p(dets), c_int64(n), c_int64(dets.shape[1]), _p(order), _p(areas), c_float(thresh), _p(keep), ctypes.byref(num))
    return keep[:num.value]


# ---- oracle/_ref: the reference's own nms.c compiled where it lies ------------------
def ref_available(name):
    return os.path.exists(os.path.join(_HERE, "_ref", name))


def ref_cpu_nms(dets, thresh):
    """Runs the REFERENCE cpu_nms (cuda_functions/nms_{2D,3D}/src/nms.c) through the TH shim."""
    dets = _f32(dets)
    n = dets.shape[0]
    name = "libref_nms3d.so" if dets.shape[1] == 7 else "libref_nms2d.so"
    L = ctypes.CDLL(os.path.join(_HERE, "_ref", name))
    order = np.ascontiguousarray(sort_order(dets[:, -1]), dtype=np.int64)
    areas = _areas(dets)
    keep = np.empty(max(n, 1), dtype=np.int64)
    num = np.zeros(1, dtype=np.int64)
    L.ref_cpu_nms(_p(keep), _p(num), _p(dets), ctypes.c_long(n), ctypes.c_long(dets.shape[1]), _p(order), _p(areas),
                  c_float(thresh))
    return keep[:num[0]]
