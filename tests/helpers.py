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
