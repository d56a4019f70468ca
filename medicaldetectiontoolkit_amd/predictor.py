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


def nms_2to3D(dets, thresh, device=None):
    """Drop-in for predictor.nms_2to3D (predictor.py:710-773): dets (n, (y1, x1, y2, x2, score, slice_id)) numpy;
    returns (keep, keep_z): indices into `dets` of the cluster cores and their [z1, z2] extents.
    Score ties are ordered "lower index first" (the reference's argsort()[::-1] is unstable)."""
    dets = np.asarray(dets, dtype=np.float64)
    n = dets.shape[0]
    if n == 0:
        return [], []
    L = _lib.lib()
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    order = np.argsort(-dets[:, 4], kind="stable")
    d = torch.from_numpy(np.ascontiguousarray(dets[order])).to(device)
    n_slices = int(dets[:, 5].max()) + 1
    keep = torch.empty(n, dtype=torch.int64, device=device)
    keep_z = torch.empty((n, 2), dtype=torch.float64, device=device)
    num = torch.zeros(1, dtype=torch.int32, device=device)
    wsb = L.mdt_nms_2to3d_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        rc = L.mdt_nms_2to3d(_lib.ptr(d), n, n_slices, ctypes.c_double(float(thresh)), _lib.ptr(keep), _lib.ptr(keep_z),
                             _lib.ptr(num), _lib.ptr(ws), wsb, _lib.current_stream_ptr())
    _lib.check(rc, "mdt_nms_2to3d")
    k = int(num.item())
    return [int(order[j]) for j in keep[:k].cpu().numpy()], [list(z) for z in keep_z[:k].cpu().numpy()]


def merge_2D_to_3D_preds_per_patient(in_patient_results_list, class_dict, merge_3D_iou, device=None):
    """predictor.py:554-593: 2D patient results (slices in the batch dimension) -> 3D results (dummy batch dim 1)."""
    out = []
    for cl in list(class_dict.keys()):
        boxes, slice_ids = [], []
        for bix, b in enumerate(in_patient_results_list):
            det = [box for box in b if box["box_type"] == "det" and box["box_pred_class_id"] == cl]
            boxes += det
            slice_ids += [bix] * len(det)
        if not boxes:
            continue
        coords = np.array([b["box_coords"] for b in boxes], dtype=np.float64)
        scores = np.array([b["box_score"] for b in boxes], dtype=np.float64)
        keep_ix, keep_z = nms_2to3D(np.concatenate((coords, scores[:, None], np.array(slice_ids, dtype=np.float64)[:, None]), axis=1),
                                    merge_3D_iou, device=device)
        for kix, kz in zip(keep_ix, keep_z):
            out.append({"box_type": "det", "box_coords": list(coords[kix]) + kz, "box_score": scores[kix], "box_pred_class_id": cl})
    out += [box for b in in_patient_results_list for box in b if box["box_type"] == "gt"]
    return [out]


# ----------------------------------------------------------------------------------------------------------
# patch-tiled prediction of one patient (predictor.py:370-455 spatial_tiling_forward + :458-510
# batch_tiling_forward + :514-550 apply_wbc_to_patient), patches sharded over ranks
# ----------------------------------------------------------------------------------------------------------
def box_patch_center_factor(box_coords, patch_size):
    """predictor.py:428-430: mean over axes of norm.pdf(centre, loc=ps/2, scale=0.8*ps/2) * sqrt(2 pi) * 0.8*ps/2,
    i.e. exp(-0.5 * ((centre - ps/2) / (0.8 * ps/2))**2)."""
    dim = len(patch_size)
    c = np.asarray(box_coords, dtype=np.float64)
    centres = [(c[0] + c[2]) / 2, (c[1] + c[3]) / 2] + ([(c[4] + c[5]) / 2] if dim == 3 else [])
    half = np.asarray(patch_size, dtype=np.float64) / 2
    return float(np.mean([np.exp(-0.5 * ((bc - pc) / (pc * 0.8)) ** 2) for bc, pc in zip(centres, half)]))


def _axis_cover_prefix(intervals, extent):
    """prefix sums P[i] = sum_{k < i} cover[k] of the number of [lo, hi) intervals covering index k along one axis"""
    cover = np.zeros(extent + 1, dtype=np.int64)
    for lo, hi in intervals:
        cover[lo] += 1
        cover[hi] -= 1
    cover = np.cumsum(cover)[:extent]
    return np.concatenate([[0], np.cumsum(cover)])


def _python_slice_bounds(start, stop, extent):
    """vectorised slice(start, stop).indices(extent): negative values wrap once, everything is clamped to [0, extent]"""
    a = np.where(start < 0, start + extent, start)
    b = np.where(stop < 0, stop + extent, stop)
    return np.clip(a, 0, extent), np.clip(b, 0, extent)


def box_n_overlaps(int_coords, crop_coords, spatial):
    """predictor.py:432-434 for all boxes at once.  The reference builds the patch-overlap map of the pass (:392-401) and takes
    np.mean(map[ic[1]:ic[3], ic[0]:ic[2], ic[4]:ic[5]]) per box -- the y axis sliced with the x coordinates and vice versa,
    negative starts wrapping like any numpy slice, NaN when the slice is empty.  The patches are a
    cartesian product of per-axis intervals (dataloader_utils.get_patch_crop_coords), so the map is the outer product of
    three per-axis cover counts and a box mean is (sum_y)(sum_x)(sum_z) / (n_y n_x n_z) from three prefix tables: exact integer
    sums, the same float64 quotient np.mean forms.  int_coords [n, 6] int64 (floor / ceil already applied)."""
    c = np.asarray(int_coords, dtype=np.int64).reshape(-1, 6)
    axes = [np.unique(crop_coords[:, 2 * a:2 * a + 2], axis=0) for a in range(3)]
    if len(axes[0]) * len(axes[1]) * len(axes[2]) != crop_coords.shape[0]:      # not a product grid: the literal map
        overlap = np.zeros(spatial, dtype=np.uint8)
        for pc in crop_coords:
            overlap[pc[0]:pc[1], pc[2]:pc[3], pc[4]:pc[5]] += 1
        out = np.empty(c.shape[0])
        for i, ic in enumerate(c):
            region = overlap[ic[1]:ic[3], ic[0]:ic[2], ic[4]:ic[5]]
            out[i] = float(np.mean(region)) if region.size else float("nan")
        return out
    prefix = [_axis_cover_prefix(axes[a], spatial[a]) for a in range(3)]
    num = np.ones(c.shape[0], dtype=np.int64)
    den = np.ones(c.shape[0], dtype=np.int64)
    for axis, (s_col, e_col) in enumerate(((1, 3), (0, 2), (4, 5))):                 # axis 0 is sliced with the x columns
        a, b = _python_slice_bounds(c[:, s_col], c[:, e_col], spatial[axis])
        n = np.maximum(b - a, 0)
        num *= np.where(n > 0, prefix[axis][np.maximum(b, a)] - prefix[axis][a], 0)
        den *= n
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(den > 0, num.astype(np.float64) / den.astype(np.float64), np.nan)


def box_patch_center_factors(coords, patch_size):
    """box_patch_center_factor for [n, 2 * dim] patch-frame boxes at once"""
    c = np.asarray(coords, dtype=np.float64).reshape(-1, 2 * len(patch_size))
    dim = len(patch_size)
    centres = [(c[:, 0] + c[:, 2]) / 2, (c[:, 1] + c[:, 3]) / 2] + ([(c[:, 4] + c[:, 5]) / 2] if dim == 3 else [])
    half = np.asarray(patch_size, dtype=np.float64) / 2
    return np.mean([np.exp(-0.5 * ((bc - pc) / (pc * 0.8)) ** 2) for bc, pc in zip(centres, half)], axis=0)


def _patch_tensor(vol_t, pc, fy, fx):
    """patch `pc` (crop coordinates in the MIRRORED frame of the pass) of the device-resident volume [C, Y, X, Z]: the
    mirrored volume is never materialised -- the patch is cut from the original at the mirrored-back coordinates and flipped"""
    Y, X = vol_t.shape[1], vol_t.shape[2]
    y0, y1 = (Y - pc[1], Y - pc[0]) if fy else (pc[0], pc[1])
    x0, x1 = (X - pc[3], X - pc[2]) if fx else (pc[2], pc[3])
    p = vol_t[:, y0:y1, x0:x1, pc[4]:pc[5]]
    dims = ([1] if fy else []) + ([2] if fx else [])
    return p.flip(dims) if dims else p


def _forward_chunk_dicts(net, batch_t, amp_dtype):
    """generic nets (anything with test_forward): box dicts per chunk -> (boxes [n, 6], batch_ix, class, score) numpy"""
    batch = {"data": batch_t}
    kw = {"return_masks": False} if "return_masks" in net.test_forward.__code__.co_varnames else {}
    if amp_dtype is not None:
        with torch.autocast("cuda", dtype=amp_dtype):
            res = net.test_forward(batch, **kw)
    else:
        res = net.test_forward(batch, **kw)
    rows = [list(np.asarray(box["box_coords"], dtype=np.float64)) + [float(k), float(box["box_pred_class_id"]), float(box["box_score"])]
            for k, bl in enumerate(res["boxes"]) for box in bl if box["box_type"] == "det"]
    return np.asarray(rows, dtype=np.float64).reshape(-1, 9), res


def collect_raw_boxes(net, data, cf, amp_dtype=None, rank_ix="0", test_aug=False, with_seg=False):
    """Patch-tiled forward of one patient WITHOUT consolidation (predictor.py:279-455): returns
    (raw_boxes, info) where raw_boxes is the reference's per-patient box-dict list in patient coordinates, each det
    carrying 'patch_id' = "<rank_ix>_<aug>_<patch>", 'box_patch_center_factor' and 'box_n_overlaps'
    (this is what the reference pickles as raw_pred_boxes_list, predictor.py:192-194).  Patches (x mirrored passes)
    are sharded round-robin over ranks and the rows all_gathered.

    Device-resident: the volume goes up ONCE, patches (and their mirrored versions) are cut on the device, nets that offer
    `test_forward_detections` keep their detections on the device through all chunks and passes (ONE read-out per patient),
    and patch-centre factor / overlap count / un-mirroring are computed for all boxes at once; box dicts are built at the end."""
    from . import distributed as mdist
    from .utils.dataloader_utils import get_patch_crop_coords
    dim = cf.dim
    assert dim == 3, "patch-tiled 3D prediction (2D slices go through merge_2D_to_3D_preds_per_patient)"
    dev = net.device_
    vol_t = data.to(dev) if torch.is_tensor(data) else torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(dev)
    # volumes smaller than the training patch are edge-padded to it, the surplus split low//2 | rest like
    # dataloader_utils.pad_nd_image (experiments/lidc_exp/data_loader.py:346-349); like the reference, predictions
    # stay in the PADDED frame (info["pad_below"] lets a caller shift them back)
    pad_below = [0, 0, 0]
    if any(vol_t.shape[d + 1] < cf.patch_size[d] for d in range(3)):
        diff = [max(cf.patch_size[d] - vol_t.shape[d + 1], 0) for d in range(3)]
        pad_below = [v // 2 for v in diff]
        pads = []
        for v in reversed(diff):                      # F.pad lists the last axis first
            pads += [v // 2, v // 2 + v % 2]
        vol_t = torch.nn.functional.pad(vol_t[None], pads, mode="replicate")[0]
    spatial = tuple(int(v) for v in vol_t.shape[1:])
    Y, X = spatial[0], spatial[1]
    coords = get_patch_crop_coords(np.zeros(spatial, dtype=np.uint8), cf.patch_size)
    n_patches = coords.shape[0]
    augs = [(False, False)] + ([(True, False), (False, True), (True, True)] if test_aug else [])
    items = [(a, p) for a in range(len(augs)) for p in range(n_patches)]
    mine = [items[i] for i in mdist.shard_indices(len(items))]
    fast = hasattr(net, "test_forward_detections") and not with_seg
    dev_rows, dev_keep, host_rows, row_pass, row_patch = [], [], [], [], []
    pass_coords = []
    seg_sum = np.zeros(spatial, dtype=np.float32) if with_seg else None
    for a, (fy, fx) in enumerate(augs):
        ids = [p for (aa, p) in mine if aa == a]
        c_a = coords.copy()
        if fy:      # mirrored image + mirrored crop coordinates (get_mirrored_patch_crops, predictor.py:777-816)
            c_a[:, 0], c_a[:, 1] = Y - coords[:, 1], Y - coords[:, 0]
        if fx:
            c_a[:, 2], c_a[:, 3] = X - coords[:, 3], X - coords[:, 2]
        pass_coords.append(c_a)
        for i in range(0, len(ids), cf.batch_size):
            chunk = ids[i:i + cf.batch_size]
            batch_t = torch.stack([_patch_tensor(vol_t, c_a[p], fy, fx) for p in chunk])
            if fast:
                if amp_dtype is not None:
                    with torch.autocast("cuda", dtype=amp_dtype):
                        rows, keep = net.test_forward_detections(batch_t)
                else:
                    rows, keep = net.test_forward_detections(batch_t)
                dev_rows.append(rows)
                dev_keep.append(keep)
                per = rows.shape[0] // len(chunk)                       # M rows per batch element, element-major
                row_pass += [a] * rows.shape[0]
                row_patch += [p for p in chunk for _ in range(per)]
            else:
                rows, res = _forward_chunk_dicts(net, batch_t, amp_dtype)
                host_rows.append(rows)
                row_pass += [a] * rows.shape[0]
                row_patch += [chunk[int(k)] for k in rows[:, 6]]
                if with_seg and a == 0:
                    for k, p in enumerate(chunk):
                        pc = c_a[p]
                        seg_sum[pc[0]:pc[1], pc[2]:pc[3], pc[4]:pc[5]] += np.asarray(res["seg_preds"][k][0], dtype=np.float32)
    row_pass, row_patch = np.asarray(row_pass, dtype=np.int64), np.asarray(row_patch, dtype=np.int64)
    if fast:
        if dev_rows:
            det = torch.cat(dev_rows).double().cpu().numpy()                # the one read-out of the patient
            keep = torch.cat(dev_keep).cpu().numpy()
            det, row_pass, row_patch = det[keep], row_pass[keep], row_patch[keep]
        else:
            det = np.zeros((0, 9))
    else:
        det = np.concatenate(host_rows, 0) if host_rows else np.zeros((0, 9))
    # ---- all boxes at once: centre factor in the patch frame, shift to the (mirrored) patient frame, overlap count, un-mirror
    local = np.zeros((det.shape[0], 11), dtype=np.float64)
    if det.shape[0]:
        c = det[:, :6].copy()
        fac = box_patch_center_factors(c, cf.patch_size)
        n_ov = np.empty(det.shape[0])
        for a, (fy, fx) in enumerate(augs):
            sel = np.nonzero(row_pass == a)[0]
            if sel.size == 0:
                continue
            pc = pass_coords[a][row_patch[sel]]
            c[sel] += pc[:, [0, 2, 0, 2, 4, 4]]
            ic = c[sel].copy()
            ic[:, [0, 1, 4]] = np.floor(ic[:, [0, 1, 4]])
            ic[:, [2, 3, 5]] = np.ceil(ic[:, [2, 3, 5]])
            n_ov[sel] = box_n_overlaps(ic.astype(np.int64), pass_coords[a], spatial)       # in the frame of the pass, BEFORE un-mirroring
            if fy:
                c[sel, 0], c[sel, 2] = Y - c[sel, 2], Y - c[sel, 0]
            if fx:
                c[sel, 1], c[sel, 3] = X - c[sel, 3], X - c[sel, 1]
        local[:, :6] = c
        local[:, 6] = det[:, 8]                      # score
        local[:, 7] = det[:, 7]                      # class id
        local[:, 8] = fac
        local[:, 9] = row_pass * n_patches + row_patch
        local[:, 10] = n_ov
    allrows = mdist.gather_rows(torch.from_numpy(local).to(dev)).cpu().numpy() if mdist.world()[1] > 1 else local
    raw = []
    for r in allrows:
        q = int(r[9])
        raw.append({"box_type": "det", "box_coords": r[:6].copy(), "box_score": float(r[6]), "box_pred_class_id": int(r[7]),
                    "patch_id": "%s_%d_%d" % (rank_ix, q // n_patches, q % n_patches), "box_patch_center_factor": float(r[8]),
                    "box_n_overlaps": float(r[10])})
    info = {"n_patches": n_patches, "n_passes": len(augs), "pad_below": pad_below, "padded_shape": tuple(spatial)}
    if with_seg:
        overlap0 = np.zeros(spatial, dtype=np.uint8)
        for pc in coords:
            overlap0[pc[0]:pc[1], pc[2]:pc[3], pc[4]:pc[5]] += 1
        m = overlap0 > 0
        seg_sum[m] /= overlap0[m]
        info["seg_preds"] = seg_sum[None, None]
    return raw, info


def apply_wbc_to_patient(raw_boxes, cf, n_ens, device=None):
    """predictor.py:514-550 for one (3D) patient: per foreground class weighted box clustering of the raw boxes;
    ground-truth boxes are passed through."""
    out = []
    for cl in sorted(cf.class_dict.keys()):
        sel = [b for b in raw_boxes if b["box_type"] == "det" and b["box_pred_class_id"] == cl]
        if not sel:
            continue
        dets = np.array([list(b["box_coords"]) + [b["box_score"], b["box_patch_center_factor"], b["box_n_overlaps"]] for b in sel])
        pid = np.array([b["patch_id"] for b in sel])
        ks, kc = weighted_box_clustering(dets, pid, cf.wcs_iou, n_ens, device=device)
        for s_, c_ in zip(ks, kc):
            out.append({"box_type": "det", "box_coords": np.array(c_), "box_score": s_, "box_pred_class_id": cl})
    out.extend([b for b in raw_boxes if b["box_type"] == "gt"])
    return out


def predict_patient(net, data, cf, n_ens=None, amp_dtype=None, rank_ix="0", test_aug=False, with_seg=False):
    """data: numpy [C, Y, X, Z] whole-patient volume -> {'boxes': [[box dicts]], ...} like predictor.predict_patient:
    collect_raw_boxes (tiling, optional mirrored passes, gather over ranks) + apply_wbc_to_patient.
    n_ens defaults to the number of passes (expected predictions per position)."""
    raw, info = collect_raw_boxes(net, data, cf, amp_dtype=amp_dtype, rank_ix=rank_ix, test_aug=test_aug, with_seg=with_seg)
    if n_ens is None:
        n_ens = info["n_passes"]
    res = {"boxes": [apply_wbc_to_patient(raw, cf, n_ens, device=net.device_)], "n_patches": info["n_patches"],
           "n_raw_boxes": len(raw), "n_passes": info["n_passes"]}
    if with_seg:
        res["seg_preds"] = info["seg_preds"]
    return res


def predict_test_set(net, patients, cf, checkpoint_paths=None, out_dir=None, test_aug=None, amp_dtype=None):
    """predictor.py:122-216 in miniature: temporal ensembling over saved epochs (each checkpoint = one ensemble member,
    patch ids prefixed with its rank), raw predictions pickled in the reference's format
    ('raw_pred_boxes_list.pickle' = [[boxes_per_batch_element, pid], ...]), then WBC per patient with
    n_ens = n_checkpoints x n_mirrored_passes.  patients: iterable of (data [C,Y,X,Z], pid)."""
    import os
    import pickle
    from .utils import exp_utils
    test_aug = cf.test_aug if test_aug is None else test_aug
    members = list(checkpoint_paths) if checkpoint_paths else [None]
    patients = list(patients)                       # iterated once per ensemble member
    raw_per_patient = {}
    order = []
    n_passes = 1
    for rank_ix, ckpt in enumerate(members):
        if ckpt is not None:
            exp_utils.load_checkpoint(ckpt, net, map_location=net.device_)
        for data, pid in patients:
            raw, info = collect_raw_boxes(net, data, cf, amp_dtype=amp_dtype, rank_ix=str(rank_ix), test_aug=test_aug)
            n_passes = info["n_passes"]
            if pid not in raw_per_patient:
                raw_per_patient[pid] = []
                order.append(pid)
            raw_per_patient[pid] += raw
    raw_list = [[[raw_per_patient[pid]], pid] for pid in order]
    if out_dir is not None and exp_utils._is_rank0():
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "raw_pred_boxes_list.pickle"), "wb") as handle:
            pickle.dump(raw_list, handle)
    n_ens = len(members) * n_passes
    return [[[apply_wbc_to_patient(boxes[0], cf, n_ens, device=net.device_)], pid] for boxes, pid in raw_list]
