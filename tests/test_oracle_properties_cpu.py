"""Property tests of the CPU oracle (size-independent invariants the GPU tests rely on at full size):
NMS keeps an independent set that is maximal w.r.t. the greedy order; RoIAlign forward is linear in the image and
its backward is the exact transpose."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import oracle
from tests.helpers import nms_boxes, random_boxes_3d


def _iou3(a, b):
    f = np.float32
    inter, sa, sb = f(1), f(1), f(1)
    for lo, hi in ((0, 2), (1, 3), (4, 5)):
        inter = f(inter * max(f(f(min(a[hi], b[hi]) - max(a[lo], b[lo])) + f(1)), f(0)))
        sa = f(sa * f(f(a[hi] - a[lo]) + f(1)))
        sb = f(sb * f(f(b[hi] - b[lo]) + f(1)))
    return f(inter / f(f(sa + sb) - inter))


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10 ** 6), n=st.integers(1, 120), thresh=st.sampled_from([1e-5, 0.1, 0.5, 0.7]))
def test_nms_keeps_maximal_independent_set(seed, n, thresh):
    dets = nms_boxes(np.random.default_rng(seed), n, dim=3, patch=64.0)
    keep = oracle.gpu_nms(dets, thresh, True)
    kept = set(keep.tolist())
    t = np.float32(thresh)
    # descending score order, no two kept boxes overlap above the threshold
    assert np.all(np.diff(dets[keep, -1]) <= 0)
    for i, a in enumerate(keep):
        for b in keep[i + 1:]:
            assert not (_iou3(dets[a], dets[b]) > t)
    # every dropped box is suppressed by a kept box with a higher score
    for j in range(n):
        if j not in kept:
            assert any(dets[k, -1] >= dets[j, -1] and _iou3(dets[k], dets[j]) > t for k in keep)


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10 ** 6), alpha=st.floats(-2, 2), beta=st.floats(-2, 2))
def test_roialign_forward_is_linear_and_backward_is_its_transpose(seed, alpha, beta):
    rng = np.random.default_rng(seed)
    shape = (2, 2, 6, 7, 8)
    x1 = rng.normal(size=shape).astype(np.float32)
    x2 = rng.normal(size=shape).astype(np.float32)
    boxes = random_boxes_3d(rng, 5, patch=16.0, xy=(2, 14), z=(2, 14), spill=True)
    ind = rng.integers(0, 2, size=5).astype(np.int32)
    crop = (3, 4, 2)
    f = lambda x: oracle.crop_and_resize_forward(x, boxes, ind, crop).astype(np.float64)
    lhs = f((np.float32(alpha) * x1 + np.float32(beta) * x2).astype(np.float32))
    rhs = alpha * f(x1) + beta * f(x2)
    assert np.abs(lhs - rhs).max() < 1e-4
    g = rng.normal(size=lhs.shape).astype(np.float32)
    gi = oracle.crop_and_resize_backward(g, boxes, ind, shape).astype(np.float64)
    assert abs((f(x1) * g).sum() - (x1.astype(np.float64) * gi).sum()) < 1e-3
