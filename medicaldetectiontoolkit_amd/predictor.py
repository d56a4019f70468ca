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


def _forward_patches(net, data, coords, chunk_ids, cf, amp_dtype, with_seg):
    patches = np.stack([data[:, coords[p][0]:coords[p][1], coords[p][2]:coords[p][3], coords[p][4]:coords[p][5]] for p in chunk_ids])
    batch = {"data": np.ascontiguousarray(patches, dtype=np.float32)}
    kw = {"return_masks": False} if "return_masks" in net.test_forward.__code__.co_varnames else {}
    if amp_dtype is not None:
        with torch.autocast("cuda", dtype=amp_dtype):
            return net.test_forward(batch, **kw)
    return net.test_forward(batch, **kw)


def collect_raw_boxes(net, data, cf, amp_dtype=None, rank_ix="0", test_aug=False, with_seg=False):
    """Patch-tiled forward of one patient WITHOUT consolidation (predictor.py:279-455): returns
    (raw_boxes, info) where raw_boxes is the reference's per-patient box-dict list in patient coordinates, each det
    carrying 'patch_id' = "<rank_ix>_<aug>_<patch>", 'box_patch_center_factor' and 'box_n_overlaps'
    (this is what the reference pickles as raw_pred_boxes_list, predictor.py:192-194).  Patches (x mirrored passes)
    are sharded round-robin over ranks and the rows all_gathered."""
    from . import distributed as mdist
    from .utils.dataloader_utils import get_patch_crop_coords
    dim = cf.dim
    assert dim == 3, "patch-tiled 3D prediction (2D slices go through merge_2D_to_3D_preds_per_patient)"
    dev = net.device_
    # volumes smaller than the training patch are edge-padded to it, the surplus split low//2 | rest like
    # dataloader_utils.pad_nd_image (experiments/lidc_exp/data_loader.py:346-349); like the reference, predictions
    # stay in the PADDED frame (info["pad_below"] lets a caller shift them back)
    pad_below = [0, 0, 0]
    if any(data.shape[d + 1] < cf.patch_size[d] for d in range(3)):
        diff = [max(cf.patch_size[d] - data.shape[d + 1], 0) for d in range(3)]
        pad_below = [v // 2 for v in diff]
        data = np.pad(data, [(0, 0)] + [(v // 2, v // 2 + v % 2) for v in diff], mode="edge")
    spatial = data.shape[1:]
    Y, X = spatial[0], spatial[1]
    coords = get_patch_crop_coords(np.zeros(spatial, dtype=np.uint8), cf.patch_size)
    n_patches = coords.shape[0]
    augs = [(False, False)] + ([(True, False), (False, True), (True, True)] if test_aug else [])
    items = [(a, p) for a in range(len(augs)) for p in range(n_patches)]
    mine = [items[i] for i in mdist.shard_indices(len(items))]
    rows = []
    overlap0 = None
    seg_sum = np.zeros(spatial, dtype=np.float32) if with_seg else None
    for a, (fy, fx) in enumerate(augs):
        ids = [p for (aa, p) in mine if aa == a]
        d = data
        c_a = coords.copy()
        if fy:      # mirrored image + mirrored crop coordinates (get_mirrored_patch_crops, predictor.py:777-816)
            d = d[:, ::-1]
            c_a[:, 0], c_a[:, 1] = Y - coords[:, 1], Y - coords[:, 0]
        if fx:
            d = d[:, :, ::-1]
            c_a[:, 2], c_a[:, 3] = X - coords[:, 3], X - coords[:, 2]
        # patches covering each voxel, in the frame of THIS pass (spatial_tiling_forward builds it per call, :392-401)
        overlap = np.zeros(spatial, dtype=np.uint8)
        for pc in c_a:
            overlap[pc[0]:pc[1], pc[2]:pc[3], pc[4]:pc[5]] += 1
        if a == 0:
            overlap0 = overlap
        for i in range(0, len(ids), cf.batch_size):
            chunk = ids[i:i + cf.batch_size]
            res = _forward_patches(net, d, c_a, chunk, cf, amp_dtype, with_seg)
            for k, p in enumerate(chunk):
                pc = c_a[p]
                if with_seg and a == 0:
                    seg_sum[pc[0]:pc[1], pc[2]:pc[3], pc[4]:pc[5]] += res["seg_preds"][k][0]
                for box in res["boxes"][k]:
                    if box["box_type"] != "det":
                        continue
                    c = np.asarray(box["box_coords"], dtype=np.float64)
                    fac = box_patch_center_factor(c, cf.patch_size)
                    c = c + np.array([pc[0], pc[2], pc[0], pc[2], pc[4], pc[4]])
                    # overlap count under the box, evaluated in the (possibly mirrored) frame of the pass BEFORE the box is
                    # mirrored back.  The reference slices the y axis with the x coordinates and vice versa, lets numpy
                    # wrap negative starts and takes np.mean of what is left -- NaN when the slice is empty, which happens
                    # on non-square volumes (predictor.py:432-434).  Reproduced literally.
                    ic = [int(np.floor(v)) if ix % 2 == 0 else int(np.ceil(v)) for ix, v in enumerate(c)]
                    region = overlap[ic[1]:ic[3], ic[0]:ic[2], ic[4]:ic[5]]
                    n_ov = float(np.mean(region)) if region.size else float("nan")
                    if fy:
                        c[0], c[2] = Y - c[2], Y - c[0]
                    if fx:
                        c[1], c[3] = X - c[3], X - c[1]
                    rows.append(list(c) + [float(box["box_score"]), float(box["box_pred_class_id"]), fac, float(a * n_patches + p), n_ov])
    local = torch.tensor(rows, dtype=torch.float64, device=dev).view(-1, 11)
    allrows = mdist.gather_rows(local).cpu().numpy()
    raw = []
    for r in allrows:
        q = int(r[9])
        raw.append({"box_type": "det", "box_coords": r[:6].copy(), "box_score": float(r[6]), "box_pred_class_id": int(r[7]),
                    "patch_id": "%s_%d_%d" % (rank_ix, q // n_patches, q % n_patches), "box_patch_center_factor": float(r[8]),
                    "box_n_overlaps": float(r[10])})
    info = {"n_patches": n_patches, "n_passes": len(augs), "pad_below": pad_below, "padded_shape": tuple(spatial)}
    if with_seg:
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
