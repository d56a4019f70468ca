"""Shared implementation behind cuda_functions.nms_{2D,3D}.pth_nms."""
import ctypes

import torch

from .. import _lib


def _sorted_dets(dets, dim):
    if dets.dim() != 2 or dets.size(1) != 2 * dim + 1:
        raise ValueError("dets must be [N, %d] (coords + score), got %s" % (2 * dim + 1, tuple(dets.shape)))
    _lib.require_cuda(dets, "dets")
    dets = dets.detach()
    if dets.dtype != torch.float32:
        dets = dets.float()
    scores = dets[:, -1]
    # pth_nms.py:10-12 uses an unstable sort; ties are defined here as "lower index first".
    order = torch.sort(scores, dim=0, descending=True, stable=True)[1]
    return order, dets[order].contiguous()


def nms_sorted(dets_sorted, thresh, dim, rule=_lib.NMS_RULE_GT, max_keep=0):
    """dets_sorted: [N, 2*dim+1] fp32 CUDA, sorted by descending score.
    Returns (keep[int64, device, padded with -1], num_out[int32, device])."""
    L = _lib.lib()
    n = dets_sorted.size(0)
    dev = dets_sorted.device
    stride = max_keep if 0 < max_keep < n else n
    keep = torch.empty(max(stride, 1), dtype=torch.int64, device=dev)
    num_out = torch.zeros(1, dtype=torch.int32, device=dev)
    if n == 0:
        return keep[:0], num_out
    ws_bytes = L.mdt_nms_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    fn = L.mdt_nms_3d if dim == 3 else L.mdt_nms_2d
    with torch.cuda.device(dev):
        rc = fn(_lib.ptr(dets_sorted), n, ctypes.c_float(thresh), rule, max_keep, _lib.ptr(keep),
                _lib.ptr(num_out), _lib.ptr(ws), ws_bytes, _lib.current_stream_ptr())
    _lib.check(rc, "mdt_nms_%dd" % dim)
    return keep, num_out


def nms_gpu(dets, thresh, dim):
    order, dets_sorted = _sorted_dets(dets, dim)
    keep, num_out = nms_sorted(dets_sorted, float(thresh), dim, _lib.NMS_RULE_GT, 0)
    n = int(num_out.item())  # the reference synchronises here too (nms_cuda.c:34, pth_nms.py:17)
    return order[keep[:n]].contiguous()


def nms_cpu(dets, thresh, dim):
    """The reference's CPU rule (suppress on IoU >= thresh, nms.c:64); computed on the GPU,
    returned on the CPU like pth_nms.py:20-37."""
    dev_dets = dets if dets.is_cuda else dets.cuda()
    order, dets_sorted = _sorted_dets(dev_dets, dim)
    keep, num_out = nms_sorted(dets_sorted, float(thresh), dim, _lib.NMS_RULE_GE, 0)
    n = int(num_out.item())
    return order[keep[:n]].contiguous().cpu()
