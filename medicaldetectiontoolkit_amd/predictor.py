"""Mirror of the box-consolidation part of the reference's predictor.py, on the gfx950 WBC kernel."""
import ctypes

import numpy as np
import torch

from . import _lib


def weighted_box_clustering_device(dets_sorted, patch_ids, thresh, n_ens, n_patch_ids=None):
    """dets_sorted [n, 2*dim+3] f64 device tensor sorted by descending score; patch_ids [n] i32 device.
    Returns (scores [k] f64, coords [k, 2*dim] f64) device tensors."""
    L = _lib.lib()
    dev = dets_sorted.device
    n, dim = dets_sorted.size(0), (dets_sorted.size(1) - 3) // 2
    out_s = torch.empty(max(n, 1), dtype=torch.float64, device=dev)
    out_c = torch.empty((max(n, 1), 2 * dim), dtype=torch.float64, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    if n == 0:
        return out_s[:0], out_c[:0]
    if n_patch_ids is None:
        n_patch_ids = int(patch_ids.max().item()) + 1
    wsb = L.mdt_wbc_workspace_bytes(n, n_patch_ids)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.mdt_weighted_box_clustering(_lib.ptr(dets_sorted.contiguous()), _lib.ptr(patch_ids.contiguous()), n, dim,
                                           n_patch_ids, ctypes.c_double(thresh), ctypes.c_double(n_ens),
                                           _lib.ptr(out_s), _lib.ptr(out_c), _lib.ptr(num), _lib.ptr(ws), wsb,
                                           _lib.current_stream_ptr())
    _lib.check(rc, "mdt_weighted_box_clustering")
    k = int(num.item())
    return out_s[:k], out_c[:k]


def weighted_box_clustering(dets, box_patch_id, thresh, n_ens, device=None):
    """Drop-in for predictor.weighted_box_clustering (predictor.py:597-706): numpy in, lists out.
    dets: (n, (y1, x1, y2, x2, (z1), (z2), score, patch_center_factor, n_overlaps)); box_patch_id: array of
    hashables (strings in the reference); returns (keep_scores, keep_coords)."""
    dets = np.asarray(dets, dtype=np.float64)
    if dets.shape[0] == 0:
        return [], []
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    _, pid_int = np.unique(np.asarray(box_patch_id), return_inverse=True)
    dim = 2 if dets.shape[1] == 7 else 3
    order = np.argsort(-dets[:, 2 * dim], kind="stable")   # reference: unstable argsort()[::-1]; ties -> lower index
    d = torch.from_numpy(np.ascontiguousarray(dets[order])).to(device)
    p = torch.from_numpy(np.ascontiguousarray(pid_int[order].astype(np.int32))).to(device)
    s, c = weighted_box_clustering_device(d, p, float(thresh), float(n_ens), int(pid_int.max()) + 1)
    return list(s.cpu().numpy()), [list(r) for r in c.cpu().numpy()]
