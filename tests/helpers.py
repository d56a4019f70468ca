"""Shared synthetic-input builders for the tests and the bench."""
import numpy as np


def random_boxes_3d(rng, n, patch=128.0, xy=(8.0, 64.0), z=(2.0, 16.0), spill=False):
    """normalised (y1,x1,y2,x2,z1,z2); centre U(0,1), size log-uniform; clipped to [0,1] unless spill."""
    c = rng.uniform(0, 1, size=(n, 3))
    sxy = np.exp(rng.uniform(np.log(xy[0]), np.log(xy[1]), size=(n, 2))) / patch
    sz = np.exp(rng.uniform(np.log(z[0]), np.log(z[1]), size=(n, 1))) / patch
    half = np.concatenate([sxy, sz], 1) / 2
    lo, hi = c - half, c + half
    if not spill:
        lo, hi = np.clip(lo, 0, 1), np.clip(hi, 0, 1)
    b = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1)
    return b.astype(np.float32)


def random_boxes_2d(rng, n, patch=288.0, size=(8.0, 128.0), spill=False):
    c = rng.uniform(0, 1, size=(n, 2))
    s = np.exp(rng.uniform(np.log(size[0]), np.log(size[1]), size=(n, 2))) / patch
    lo, hi = c - s / 2, c + s / 2
    if not spill:
        lo, hi = np.clip(lo, 0, 1), np.clip(hi, 0, 1)
    return np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1]], 1).astype(np.float32)


def nms_boxes(rng, n, dim=3, patch=128.0, tie_free=True):
    """pixel-coordinate detections [n, 2*dim+1] clustered around a few centres, tie-free scores."""
    k = max(1, n // 40)
    centres = rng.uniform(0.1 * patch, 0.9 * patch, size=(k, dim))
    which = rng.integers(0, k, size=n)
    c = centres[which] + rng.normal(0, 3.0, size=(n, dim))
    s = np.exp(rng.uniform(np.log(4), np.log(32), size=(n, dim)))
    lo = np.clip(c - s / 2, 0, patch)
    hi = np.clip(c + s / 2, 0, patch)
    if dim == 3:
        b = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1)
    else:
        b = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1]], 1)
    scores = rng.permutation(np.linspace(0.0, 1.0, n)) if tie_free else np.round(rng.uniform(0, 1, n), 1)
    return np.concatenate([b, scores[:, None]], 1).astype(np.float32)


def trainlike_rois_3d(rng, batch, per_element=6, side=8.0, patch=128.0):
    """RoIs as a training step hands them to one pyramid level (SURVEY.md 8(d) "train-realistic", forced onto one
    level): `per_element` sampled RoIs per batch element (train_rois_per_image, lidc configs.py:258) scattered around
    one object per element, box sides 0.75..1.4 x `side` px -- the sizes the level rule of mrcnn.py:403 routes to
    the level whose anchor scale is `side` (8 px = P2).  Returns normalised boxes [batch*per_element, 6] f32 and
    box_ind [batch*per_element] i32."""
    ctr = rng.uniform(0.25, 0.75, size=(batch, 3))
    rows = []
    for b in range(batch):
        for _ in range(per_element):
            c = ctr[b] + rng.normal(0, 0.02, size=3)
            s = rng.uniform(0.75 * side, 1.4 * side, size=3) / patch
            rows.append([c[0] - s[0] / 2, c[1] - s[1] / 2, c[0] + s[0] / 2, c[1] + s[1] / 2, c[2] - s[2] / 2, c[2] + s[2] / 2])
    boxes = np.clip(np.asarray(rows), 0.0, 1.0).astype(np.float32)
    box_ind = (np.arange(batch * per_element) // per_element).astype(np.int32)
    return boxes, box_ind
