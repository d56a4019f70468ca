"""Generates tests/golden/*.npz by IMPORTING AND RUNNING THE REFERENCE's own Python functions
(/root/reference, read-only).  Run once in the build container:  python tests/golden/make_golden.py
The .npz files are committed; nothing at test time reads /root/reference.

Pinned functions (paths relative to the reference):
  utils/model_utils.py: generate_pyramid_anchors :275, gt_anchor_matching :505, compute_overlaps :83,
                        apply_box_deltas_{2D,3D} :318/:343, clip_boxes_{2D,3D} :374/:386,
                        bbox_overlaps_{2D,3D} :429/:466, box_refinement :114
  predictor.py:         weighted_box_clustering :597
  utils/dataloader_utils.py: get_patch_crop_coords :140
"""
import faulthandler
import hashlib
import logging
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
faulthandler.dump_traceback_later(int(os.environ.get("GOLDEN_TIMEOUT", "600")), exit=True)
sys.path.insert(0, REF)
import utils.model_utils as mutils          # noqa: E402  (the reference's)
import utils.dataloader_utils as dutils     # noqa: E402
import predictor as ref_predictor           # noqa: E402
sys.path.remove(REF)

from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402

log = logging.getLogger("golden")
log.addHandler(logging.NullHandler())


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gt_boxes(rng, n, patch, dim):
    c = rng.uniform(0.2, 0.8, size=(n, dim)) * np.array(patch)
    s = rng.uniform(6, 28, size=(n, dim))
    if dim == 3:
        s[:, 2] = rng.uniform(3, 14, size=n)
    lo, hi = c - s / 2, c + s / 2
    if dim == 3:
        return np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1)
    return np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1]], 1)


def main():
    out = {}
    rng = np.random.default_rng(0)

    # ---------------- anchors ----------------
    cases = {
        "a3_mrcnn_small": Configs(dim=3, model="mrcnn", patch_size=[64, 64, 32]),
        "a3_retina_small": Configs(dim=3, model="retina_unet", patch_size=[64, 64, 32]),
        "a2_mrcnn_small": Configs(dim=2, model="mrcnn", patch_size=[64, 64]),
        "a2_retina_toy": Configs(dim=2, model="retina_net", patch_size=[64, 64]),
    }
    full = {
        "a3_mrcnn_128": Configs(dim=3, model="mrcnn"),
        "a3_retina_128": Configs(dim=3, model="retina_unet"),
        "a2_mrcnn_288": Configs(dim=2, model="mrcnn"),
    }
    anchors = {}
    for name, cf in list(cases.items()) + list(full.items()):
        a = mutils.generate_pyramid_anchors(log, cf)
        anchors[name] = a
        if name in cases:
            out[name] = a
        out[name + "_sha256"] = np.array(sha(a))
        out[name + "_shape"] = np.array(a.shape)
        out[name + "_rows"] = a[:: max(1, a.shape[0] // 257)][:300]   # sampled rows of the full-size tables

    # ---------------- anchor matching (no subsampling: rpn_train_anchors_per_image huge) ----------------
    for name, dim, G in (("a3_mrcnn_small", 3, 1), ("a3_mrcnn_small", 3, 3), ("a3_retina_small", 3, 8),
                         ("a2_mrcnn_small", 2, 3), ("a2_retina_toy", 2, 2)):
        cf = cases[name]
        cf.rpn_train_anchors_per_image = 10 ** 6
        gt = gt_boxes(rng, G, cf.patch_size, dim)
        # make sure at least one anchor exceeds the positive threshold: copy an anchor, jitter slightly
        a = anchors[name]
        gt[0] = a[rng.integers(0, len(a))] + rng.normal(0, 0.2, size=2 * dim)
        cls = rng.integers(1, 3, size=G) if "retina" in name else None
        np.random.seed(0)
        m, d = mutils.gt_anchor_matching(cf, a, gt, cls)
        ov = mutils.compute_overlaps(a, gt)
        key = "match_%s_G%d" % (name, G)
        out[key + "_gt"] = gt
        out[key + "_cls"] = np.array([]) if cls is None else cls
        out[key + "_matches"] = m
        n_pos = int((m > 0).sum())
        out[key + "_deltas"] = d[:n_pos]
        out[key + "_iou_max"] = ov.max(1)
        out[key + "_iou_argmax"] = ov.argmax(1)
        out[key + "_gt_best"] = ov.argmax(0)
    # full-size: hashes only
    for name, dim, G in (("a3_mrcnn_128", 3, 3), ("a3_retina_128", 3, 8)):
        cf = full[name]
        cf.rpn_train_anchors_per_image = 10 ** 6
        a = anchors[name]
        gt = gt_boxes(rng, G, cf.patch_size, dim)
        gt[0] = a[rng.integers(0, len(a))] + rng.normal(0, 0.2, size=2 * dim)
        cls = rng.integers(1, 3, size=G) if "retina" in name else None
        m, d = mutils.gt_anchor_matching(cf, a, gt, cls)
        key = "match_%s_G%d" % (name, G)
        out[key + "_gt"] = gt
        out[key + "_cls"] = np.array([]) if cls is None else cls
        out[key + "_matches_sha256"] = np.array(sha(m.astype(np.int32)))
        out[key + "_nonzero_idx"] = np.nonzero(m > 0)[0]
        out[key + "_n_neg"] = np.array(int((m == -1).sum()))
        out[key + "_deltas"] = d[: int((m > 0).sum())]

    # ---------------- box decode / clip / overlaps / refinement (torch fp32 on CPU) ----------------
    for dim in (2, 3):
        n = 500
        boxes = torch.from_numpy(gt_boxes(rng, n, [128] * dim, dim)).float()
        deltas = torch.from_numpy(rng.normal(0, 1.0, size=(n, 2 * dim))).float()
        std = torch.from_numpy(np.array([0.1, 0.1, 0.1, 0.2, 0.2, 0.2][: 2 * dim] if dim == 3 else [0.1, 0.1, 0.2, 0.2])).float()
        window = np.array([0, 0, 128, 128, 0, 96][: 2 * dim])
        if dim == 3:
            dec = mutils.apply_box_deltas_3D(boxes.clone(), deltas * std)
            clp = mutils.clip_boxes_3D(dec, window)
            ov = mutils.bbox_overlaps_3D(boxes[:40], boxes[40:47])
        else:
            dec = mutils.apply_box_deltas_2D(boxes.clone(), deltas * std)
            clp = mutils.clip_boxes_2D(dec, window)
            ov = mutils.bbox_overlaps_2D(boxes[:40], boxes[40:47])
        ref = mutils.box_refinement(boxes[:40], boxes[40:80])
        out["decode%d_boxes" % dim] = boxes.numpy()
        out["decode%d_deltas" % dim] = deltas.numpy()
        out["decode%d_std" % dim] = std.numpy()
        out["decode%d_window" % dim] = window.astype(np.float32)
        out["decode%d_decoded" % dim] = dec.numpy()
        out["decode%d_clipped" % dim] = clp.numpy()
        out["overlaps%d" % dim] = ov.numpy()
        out["refine%d" % dim] = ref.numpy()

    # ---------------- weighted box clustering ----------------
    for dim, n, n_true, n_ens in ((3, 200, 6, 4), (3, 2000, 20, 20), (2, 300, 8, 4)):
        true = gt_boxes(rng, n_true, [256] * dim, dim)
        which = rng.integers(0, n_true, size=n)
        coords = true[which] + rng.normal(0, 1.5, size=(n, 2 * dim))
        far = rng.random(n) < 0.1                       # some isolated false positives
        coords[far] += rng.uniform(-100, 100, size=(int(far.sum()), 1))
        # keep extents positive: the reference loops forever on a box with (hi - lo + 1) <= 0, whose IoU with
        # itself is 0 so it never leaves `order` (predictor.py:651,701-703)
        for lo_c, hi_c in ((0, 2), (1, 3)) + (((4, 5),) if dim == 3 else ()):
            coords[:, hi_c] = np.maximum(coords[:, hi_c], coords[:, lo_c] + 1.0)
        scores = rng.permutation(np.linspace(0.02, 0.99, n))
        pc = rng.uniform(0.2, 1.0, size=n)
        novs = rng.integers(1, 5, size=n).astype(np.float64)
        n_patch = 75 * 4
        pid_int = rng.integers(0, n_patch, size=n)
        pid_str = np.array(["%d_%d_%d" % (p // 300, (p // 75) % 4, p % 75) for p in pid_int])
        dets = np.concatenate([coords, scores[:, None], pc[:, None], novs[:, None]], 1)
        ks, kc = ref_predictor.weighted_box_clustering(dets.copy(), pid_str, 1e-5, n_ens)
        key = "wbc%d_n%d" % (dim, n)
        out[key + "_dets"] = dets
        out[key + "_pid"] = pid_int.astype(np.int32)
        out[key + "_n_ens"] = np.array(n_ens)
        out[key + "_scores"] = np.array(ks)
        out[key + "_coords"] = np.array(kc).reshape(len(ks), 2 * dim)

    # ---------------- 2D -> 3D merge ----------------
    for n, n_obj in ((120, 5), (900, 25)):
        ctr = rng.uniform(30, 220, size=(n_obj, 2))
        zc = rng.integers(5, 60, size=n_obj)
        zr = rng.integers(1, 6, size=n_obj)
        which = rng.integers(0, n_obj, size=n)
        c = ctr[which] + rng.normal(0, 1.5, size=(n, 2))
        sz = rng.uniform(8, 24, size=(n, 2))
        sl = zc[which] + rng.integers(-8, 9, size=n) * (rng.random(n) < 0.9) * np.minimum(1, zr[which])   # mostly near the core, with holes
        sl = np.clip(zc[which] + np.round(rng.normal(0, zr[which])).astype(int) + (rng.random(n) < 0.1) * rng.integers(4, 9, size=n), 0, 79)
        dets = np.stack([c[:, 0] - sz[:, 0] / 2, c[:, 1] - sz[:, 1] / 2, c[:, 0] + sz[:, 0] / 2, c[:, 1] + sz[:, 1] / 2,
                         rng.permutation(np.linspace(0.05, 0.99, n)), sl.astype(np.float64)], 1)
        keep, keep_z = ref_predictor.nms_2to3D(dets.copy(), 0.1)
        out["m2to3_n%d_dets" % n] = dets
        out["m2to3_n%d_keep" % n] = np.array(keep, dtype=np.int64)
        out["m2to3_n%d_keep_z" % n] = np.array(keep_z, dtype=np.float64)

    # ---------------- patch tiling ----------------
    for shape, ps in (((512, 512, 256), (128, 128, 128)), ((512, 512, 256), (128, 128, 64)), ((300, 260), (128, 128)),
                      ((128, 128, 128), (128, 128, 128)), ((200, 180, 70), (128, 128, 64))):
        img = np.zeros(shape, dtype=np.uint8)
        cc = dutils.get_patch_crop_coords(img, list(ps))
        out["patch_%s_%s" % ("x".join(map(str, shape)), "x".join(map(str, ps)))] = cc

    path = os.path.join(HERE, "reference_python.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f kB" % (os.path.getsize(path) / 1e3), len(out), "arrays")


if __name__ == "__main__":
    main()
