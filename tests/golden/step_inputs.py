"""Seeded inputs of the ASSEMBLED training-step parity cases (VERDICT r2 "whole-step parity").
Imported by BOTH the generator (tests/golden/make_step_golden.py, which runs the reference's own `net.train_forward`) and
the GPU tests (tests/test_step_parity_gpu.py).  Weights are filled from RNGs seeded by state-dict NAME
(fill_by_name below) -- the reference and this repo share module names by contract -- and the images are seeded
noise; the GT boxes are chosen by the generator from the reference's own proposals (so that positive RoI samples exist)
and travel inside the golden file.

Determinism of the reference step (no random draw changes a sampled SET): shem_poolsize = 1 (pool == sample,
model_utils.py:674-691), rpn_train_anchors_per_image large (no surplus-positive sub-sampling, model_utils.py:566-571),
train_rois_per_image large (randperm only permutes the positives, mrcnn.py:529-533)."""
import zlib

import numpy as np

PATCH = [64, 64, 32]
B = 2
# second case (round 4): a patch whose C2 map has 65 536 voxels (1 x 32 x 32 x 64), the size from which this
# repo's fp32-MFMA convolution kernels (conv3x3x3_small fwd / dgrad / wgrad, conv1x1_wgrad, conv_stem_fwd / _wgrad) are dispatched
# (utils/fused_epilogue.py use-rules) -- the step the bench times runs them, so the reference parity must see them too
# third case: THE BENCHMARKED CONFIGURATION itself (BASELINE config 3: 128^3, batch 8), Mask R-CNN only (the reference step takes ~15 min
# of CPU time at this size)
CASES = {"small": ([64, 64, 32], 2), "large": ([128, 128, 64], 1), "bench": ([128, 128, 128], 8)}


def fill_by_name(module, gain=1.0):
    """every parameter from an RNG seeded by its state-dict NAME; std = gain / sqrt(fan_in): with the He factor of
    backbone_inputs.fill_by_name the norm-free ResNet reaches activations of ~2000 and the RPN deltas overflow; this keeps
    features ~10, logits and deltas O(1) through FPN, RPN and heads"""
    import torch
    with torch.no_grad():
        for name, p in sorted(module.state_dict().items()):
            rng = np.random.default_rng(zlib.crc32(name.encode()))
            shape = tuple(p.shape)
            if len(shape) > 1:
                fan_in = int(np.prod(shape[1:]))
                v = rng.standard_normal(shape) * gain * np.sqrt(1.0 / fan_in)
            else:
                v = rng.uniform(-0.05, 0.05, size=shape)
            p.copy_(torch.from_numpy(v.astype(np.float32)))


def make_cf(model, case="small"):
    from medicaldetectiontoolkit_amd.configs import Configs
    patch, nb = CASES[case]
    kw = dict(dim=3, model=model, patch_size=list(patch), batch_size=nb, shem_poolsize=1, rpn_train_anchors_per_image=256)
    if model == "mrcnn":
        kw.update(post_nms_rois_training=40, pre_nms_limit=3000, train_rois_per_image=40)
    else:
        kw.update(retina_shem_poolsize=1)
    return Configs(**kw)


def make_image(seed=31, case="small"):
    patch, nb = CASES[case]
    rng = np.random.default_rng(seed)
    return rng.standard_normal([nb, 1] + list(patch)).astype(np.float32)


def make_batch(img, gt_boxes, gt_labels):
    """the reference's batch dict: solid ellipsoids inscribed in the GT boxes as masks / seg"""
    B = int(img.shape[0])
    Y, X, Z = (int(v) for v in img.shape[2:])
    yy, xx, zz = np.meshgrid(np.arange(Y), np.arange(X), np.arange(Z), indexing="ij")
    seg = np.zeros((B, 1, Y, X, Z), dtype=np.uint8)
    roi_masks = []
    for b in range(B):
        ms = []
        for box in gt_boxes[b]:
            c = [(box[0] + box[2]) / 2.0, (box[1] + box[3]) / 2.0, (box[4] + box[5]) / 2.0]
            r = [max((box[2] - box[0]) / 2.0, 0.5), max((box[3] - box[1]) / 2.0, 0.5), max((box[5] - box[4]) / 2.0, 0.5)]
            m = (((yy + 0.5 - c[0]) / r[0]) ** 2 + ((xx + 0.5 - c[1]) / r[1]) ** 2 + ((zz + 0.5 - c[2]) / r[2]) ** 2) <= 1.0
            ms.append(m[None].astype(np.uint8))
            seg[b, 0][m] = 1
        roi_masks.append(np.array(ms, dtype=np.uint8).reshape((-1, 1, Y, X, Z)))
    return {"data": img, "seg": seg, "pid": ["step_%d" % b for b in range(B)],
            "bb_target": [np.asarray(g, dtype=np.float32).reshape(-1, 6) for g in gt_boxes],
            "roi_labels": [np.asarray(l, dtype=np.int64) for l in gt_labels],
            "roi_masks": roi_masks, "class_target": [[int(v) - 1 for v in l] for l in gt_labels]}


def module_of(name):
    return name.split(".")[0]
