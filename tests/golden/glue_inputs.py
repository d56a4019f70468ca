"""Seeded inputs of the call-site glue parity cases.  Imported by BOTH the generator
(tests/golden/make_glue_golden.py, which feeds them to the reference's own functions) and the GPU tests
(tests/test_glue_parity_gpu.py, which feed them to the re-designed glue): only the outputs are stored.
numpy's default_rng (PCG64) stream is stable across platforms and versions."""
import numpy as np

PATCH = [64, 64, 32]
B = 2
PC = 24          # proposals per batch element in the target / refinement cases
C_FEAT = 4


def make_cf(model="mrcnn"):
    from medicaldetectiontoolkit_amd.configs import Configs
    # shem_poolsize = 1 makes SHEM deterministic (pool == sample, model_utils.py:674-691): the sampled SETS can be compared
    return Configs(dim=3, model=model, patch_size=PATCH, batch_size=B, shem_poolsize=1, rpn_train_anchors_per_image=128,
                   post_nms_rois_training=40, pre_nms_limit=3000 if model == "mrcnn" else 2000)


def softmax(x, axis=-1):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def proposal_layer_inputs(n_anchors, seed=11):
    rng = np.random.default_rng(seed)
    logits = rng.normal(0, 2.0, size=(B, n_anchors, 2))
    probs = softmax(logits).astype(np.float32)
    deltas = rng.normal(0, 0.4, size=(B, n_anchors, 6)).astype(np.float32)
    return probs, deltas


def pyramid_inputs(cf, seed=12):
    """rois on all four levels (box sides away from the half-level boundaries of mrcnn.py:403)"""
    rng = np.random.default_rng(seed)
    fmaps = [rng.normal(size=(B, C_FEAT) + tuple(int(v) for v in s)).astype(np.float32) for s in cf.backbone_shapes]
    rois = []
    for side in (4.2, 8.4, 17.0, 34.0):           # -> levels 0, 1, 2, 3 at a 64-px patch
        for _ in range(7):
            hw = side * rng.uniform(0.93, 1.07, size=2) / 64.0
            d = rng.uniform(3, 20) / 32.0
            c = rng.uniform(0.3, 0.7, size=3)
            rois.append([c[0] - hw[0] / 2, c[1] - hw[1] / 2, c[0] + hw[0] / 2, c[1] + hw[1] / 2, c[2] - d / 2, c[2] + d / 2,
                         float(rng.integers(0, B))])
    rois = np.asarray(rois, dtype=np.float32)
    rois = rois[rng.permutation(len(rois))]
    return fmaps, rois


def _box(c, s):
    return [c[0] - s[0] / 2, c[1] - s[1] / 2, c[0] + s[0] / 2, c[1] + s[1] / 2, c[2] - s[2] / 2, c[2] + s[2] / 2]


def target_layer_inputs(cf, seed=13):
    """per element: 2 GT objects (pixel boxes + binary masks + class ids), PC proposals = <= 3 jittered GT copies
    (IoU >= 0.3: never more positives than int(train_rois_per_image * roi_positive_ratio), so the reference's randperm
    only permutes them) + far-away boxes (negatives) + a few partially overlapping ones (neutral)."""
    rng = np.random.default_rng(seed)
    Y, X, Z = PATCH
    scale = np.array([Y, X, Y, X, Z, Z], dtype=np.float32)
    gt_boxes, gt_cls, gt_masks, props = [], [], [], []
    for b in range(B):
        centres = np.array([[18.0, 20.0, 10.0], [44.0, 40.0, 22.0]]) + rng.uniform(-2, 2, size=(2, 3))
        sizes = rng.uniform([10, 10, 6], [16, 16, 10], size=(2, 3))
        gb = np.array([_box(c, s) for c, s in zip(centres, sizes)], dtype=np.float32)
        gt_boxes.append(gb)
        gt_cls.append(np.array([1, 2], dtype=np.int64) if b == 0 else np.array([2, 1], dtype=np.int64))
        m = np.zeros((2, Y, X, Z), dtype=np.float32)
        yy, xx, zz = np.meshgrid(np.arange(Y), np.arange(X), np.arange(Z), indexing="ij")
        for g in range(2):   # solid ellipsoids inside the boxes
            r = sizes[g] / 2
            m[g] = (((yy - centres[g, 0]) / r[0]) ** 2 + ((xx - centres[g, 1]) / r[1]) ** 2 + ((zz - centres[g, 2]) / r[2]) ** 2 <= 1.0)
        gt_masks.append(m)
        rows = []
        for g, k in ((0, 2), (1, 1)):                                  # 3 positives: 2 around object 0, 1 around object 1
            for _ in range(k):
                rows.append(_box(centres[g] + rng.uniform(-1.0, 1.0, size=3), sizes[g] * rng.uniform(0.9, 1.1, size=3)))
        for _ in range(3):                                             # neutral: small overlap with object 0
            rows.append(_box(centres[0] + np.array([7.5, 7.5, 4.5]) * rng.choice([-1, 1], size=3), sizes[0]))
        while len(rows) < PC:                                          # negatives: corners far from both objects
            c = np.array([rng.uniform(4, 60), rng.uniform(4, 60), rng.uniform(3, 29)])
            s = rng.uniform([4, 4, 3], [10, 10, 6])
            bb = np.array(_box(c, s))
            far = all(bb[2] < g_[0] - 1 or bb[0] > g_[2] + 1 or bb[3] < g_[1] - 1 or bb[1] > g_[3] + 1 or bb[5] < g_[4] - 1 or bb[4] > g_[5] + 1 for g_ in gb)
            if far:
                rows.append(list(bb))
        rows = np.asarray(rows, dtype=np.float32)[rng.permutation(PC)]
        props.append(np.concatenate([rows / scale, np.full((PC, 1), b, dtype=np.float32)], 1))
    batch_proposals = np.concatenate(props, 0).astype(np.float32)
    scores = softmax(rng.normal(0, 1.5, size=(B * PC, cf.head_classes))).astype(np.float32)
    return batch_proposals, scores, gt_cls, gt_boxes, gt_masks


def refine_inputs(cf, seed=14):
    rng = np.random.default_rng(seed)
    n = B * PC
    c = rng.uniform(0.15, 0.85, size=(n, 3))
    s = rng.uniform([0.1, 0.1, 0.15], [0.3, 0.3, 0.4], size=(n, 3))
    rois = np.stack([c[:, 0] - s[:, 0] / 2, c[:, 1] - s[:, 1] / 2, c[:, 0] + s[:, 0] / 2, c[:, 1] + s[:, 1] / 2,
                     c[:, 2] - s[:, 2] / 2, c[:, 2] + s[:, 2] / 2], 1).astype(np.float32)
    probs = softmax(rng.normal(0, 1.5, size=(n, cf.head_classes))).astype(np.float32)
    deltas = rng.normal(0, 0.5, size=(n, cf.head_classes, 6)).astype(np.float32)
    batch_ixs = np.repeat(np.arange(B), PC).astype(np.int64)
    return rois, probs, deltas, batch_ixs


def retina_refine_inputs(cf, n_anchors, seed=15):
    rng = np.random.default_rng(seed)
    logits = rng.normal(0, 1.0, size=(B * n_anchors, cf.head_classes))
    logits[:, 0] += 3.0                                  # mostly background, like a detector
    probs = softmax(logits).astype(np.float32)
    deltas = rng.normal(0, 0.4, size=(B * n_anchors, 6)).astype(np.float32)
    batch_ixs = np.repeat(np.arange(B), n_anchors).astype(np.int64)
    return probs, deltas, batch_ixs


def head_loss_inputs(cf, n, seed=16):
    rng = np.random.default_rng(seed)
    logits = rng.normal(size=(n, cf.head_classes)).astype(np.float32)
    pred_deltas = rng.normal(0, 0.5, size=(n, cf.head_classes, 6)).astype(np.float32)
    pred_masks = rng.uniform(0.05, 0.95, size=(n, cf.head_classes) + tuple(cf.mask_shape)).astype(np.float32)
    return logits, pred_deltas, pred_masks


def rpn_loss_inputs(n_anchors, seed=17):
    """one batch element: 2 GT boxes (pixels) for the anchor matching, RPN logits and predicted deltas.
    Use with make_cf(...).rpn_train_anchors_per_image raised so that the surplus-positive sub-sampling of
    gt_anchor_matching (model_utils.py:566-571, np.random.choice) never triggers."""
    rng = np.random.default_rng(seed)
    gt = np.array([[10.0, 12.0, 26.0, 28.0, 6.0, 14.0], [36.0, 30.0, 52.0, 52.0, 16.0, 26.0]], dtype=np.float64)
    logits = rng.normal(size=(n_anchors, 2)).astype(np.float32)
    pred_deltas = rng.normal(0, 0.5, size=(n_anchors, 6)).astype(np.float32)
    return gt, logits, pred_deltas
