"""CPU restatement (numpy, float64) of the reference's host-side anchor matching -- TEST INFRASTRUCTURE ONLY.

Like everything under oracle/, this is the checker, not the product: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  Pinned against the outputs of the reference's own
utils/model_utils.py:gt_anchor_matching stored in tests/golden/reference_python.npz (tests/test_oracle_cpu.py).

Follows /root/reference/utils/model_utils.py:
  overlaps()         compute_overlaps :83-111 + compute_iou_{2D,3D} :32-79 (same operation order, float64)
  anchor_matching()  gt_anchor_matching :505-619 (negatives, one anchor per GT, positives above the IoU threshold,
                     surplus-positive sub-sampling, delta targets of the kept positives)
"""
import numpy as np


def overlaps(anchors, gt_boxes):
    """IoU [num_anchors, num_gt]; one GT column at a time against all anchors, like the reference's loop"""
    a = np.asarray(anchors, dtype=np.float64)
    g = np.asarray(gt_boxes, dtype=np.float64)
    dim = a.shape[1] // 2
    ext_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ext_g = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    if dim == 3:
        ext_a = ext_a * (a[:, 5] - a[:, 4])
        ext_g = ext_g * (g[:, 5] - g[:, 4])
    out = np.zeros((a.shape[0], g.shape[0]))
    for k in range(g.shape[0]):
        dy = np.maximum(np.minimum(g[k, 2], a[:, 2]) - np.maximum(g[k, 0], a[:, 0]), 0)
        dx = np.maximum(np.minimum(g[k, 3], a[:, 3]) - np.maximum(g[k, 1], a[:, 1]), 0)
        inter = dx * dy
        if dim == 3:
            inter = inter * np.maximum(np.minimum(g[k, 5], a[:, 5]) - np.maximum(g[k, 4], a[:, 4]), 0)
        out[:, k] = inter / (ext_g[k] + ext_a - inter)
    return out


def anchor_matching(anchors, gt_boxes, gt_class_ids, pos_iou, n_train_anchors, std_dev, rng=None):
    """-> (matches [A] int32: >0 positive (class id, 1 for an RPN), -1 negative, 0 neutral;
           deltas [n_train_anchors, 2*dim] float64: rows of the kept positives in anchor order, zeros after)."""
    a = np.asarray(anchors, dtype=np.float64)
    dim = a.shape[1] // 2
    matches = np.zeros(a.shape[0], dtype=np.int32)
    deltas = np.zeros((n_train_anchors, 2 * dim))
    if gt_boxes is None or len(gt_boxes) == 0:
        return np.full(a.shape[0], -1, dtype=np.int32), deltas
    g = np.asarray(gt_boxes, dtype=np.float64)
    cls = np.ones(len(g), dtype=np.int64) if gt_class_ids is None else np.asarray(gt_class_ids)
    iou = overlaps(a, g)
    best_gt = iou.argmax(1)
    best_iou = iou[np.arange(a.shape[0]), best_gt]
    matches[best_iou < (0.1 if dim == 2 else 0.01)] = -1
    for k, anchor_ix in enumerate(iou.argmax(0)):            # no GT stays unmatched; later GTs overwrite earlier ones
        matches[anchor_ix] = cls[k]
    above = best_iou >= pos_iou
    matches[above] = cls[best_gt[above]]
    pos = np.where(matches > 0)[0]
    surplus = len(pos) - n_train_anchors // 2
    if surplus > 0:
        drop = (np.random if rng is None else rng).choice(pos, surplus, replace=False)
        matches[drop] = 0
        pos = np.where(matches > 0)[0]
    pa, pg = a[pos], g[best_gt[pos]]
    size_a = [pa[:, 2] - pa[:, 0], pa[:, 3] - pa[:, 1]] + ([pa[:, 5] - pa[:, 4]] if dim == 3 else [])
    size_g = [pg[:, 2] - pg[:, 0], pg[:, 3] - pg[:, 1]] + ([pg[:, 5] - pg[:, 4]] if dim == 3 else [])
    lo = [0, 1] + ([4] if dim == 3 else [])
    cols = [((pg[:, lo[k]] + 0.5 * size_g[k]) - (pa[:, lo[k]] + 0.5 * size_a[k])) / size_a[k] for k in range(dim)]
    cols += [np.log(size_g[k] / size_a[k]) for k in range(dim)]
    if len(pos):
        deltas[:len(pos)] = np.stack(cols, 1) / np.asarray(std_dev, dtype=np.float64)
    return matches, deltas
