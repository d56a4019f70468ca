"""Seeded inputs shared by tests/golden/make_predictor_golden.py and the tests (only outputs are stored)."""
import numpy as np


def wbc_case(n, n_true, seed, extent=256.0, dim=3):
    """n detections jittered around n_true objects (+10% isolated false positives): dets [n, 2*dim+3] f64
    (coords, score, patch_center_factor, n_overlaps) and integer patch ids"""
    rng = np.random.default_rng(seed)
    c = rng.uniform(0.2, 0.8, size=(n_true, dim)) * extent
    s = rng.uniform(6, 28, size=(n_true, dim))
    if dim == 3:
        s[:, 2] = rng.uniform(3, 14, size=n_true)
    lo, hi = c - s / 2, c + s / 2
    true = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1]] + ([lo[:, 2], hi[:, 2]] if dim == 3 else []), 1)
    which = rng.integers(0, n_true, size=n)
    coords = true[which] + rng.normal(0, 1.5, size=(n, 2 * dim))
    far = rng.random(n) < 0.1
    coords[far] += rng.uniform(-100, 100, size=(int(far.sum()), 1))
    for lo_c, hi_c in ((0, 2), (1, 3)) + (((4, 5),) if dim == 3 else ()):     # positive extents (the reference never
        coords[:, hi_c] = np.maximum(coords[:, hi_c], coords[:, lo_c] + 1.0)   # terminates otherwise, predictor.py:651)
    scores = rng.permutation(np.linspace(0.02, 0.99, n))
    pc = rng.uniform(0.2, 1.0, size=n)
    novs = rng.integers(1, 5, size=n).astype(np.float64)
    pid = rng.integers(0, 75 * 4 * 5, size=n)
    return np.concatenate([coords, scores[:, None], pc[:, None], novs[:, None]], 1), pid.astype(np.int32)


def tiler_cases(n=200, seed=3):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(n):
        dim = int(rng.integers(2, 4))
        shape = tuple(int(v) for v in rng.integers(20, 400, size=dim))
        ps = [int(v) for v in rng.choice([31, 32, 48, 63, 64, 96, 127, 128], size=dim)]
        if dim == 3 and rng.random() < 0.2:
            ps[2] = 1
        cases.append((shape, ps))
    return cases
