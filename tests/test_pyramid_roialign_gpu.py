"""mdt_pyramid_roi_align_{forward,backward} (include/mdt_hip.h): all pyramid levels in one launch, against
  * the CPU oracle applied level by level (oracle/mdt_oracle.c restates crop_and_resize_kernel.cu), and
  * the per-level entry points (one mdt_crop_and_resize_*_{forward,backward} call per level with box_ind = -1 off-level),
on RoIs spread over all levels incl. skipped rows (batch_ix = -1) and an out-of-range level.
Bars: forward bit-exact; backward <= 2e-6 * sum|terms| per voxel (the default backward's contract), every map fully written."""
import numpy as np
import pytest
import torch

from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl as impl
from oracle import oracle

pytestmark = pytest.mark.gpu


def _case(dim, B, C, base, n, seed, n_levels=4):
    rng = np.random.default_rng(seed)
    shapes = []
    for l in range(n_levels):
        s = [max(1, v >> l) for v in base[:2]] + ([max(1, base[2] >> min(l, 2))] if dim == 3 else [])
        shapes.append((B, C) + tuple(s))
    maps = [rng.normal(size=s).astype(np.float32) for s in shapes]
    c = rng.uniform(0.2, 0.8, size=(n, dim))
    level = rng.integers(0, n_levels, size=n).astype(np.int32)
    half = (0.03 * (2.0 ** level))[:, None] * rng.uniform(0.7, 1.4, size=(n, dim))
    if dim == 3:
        boxes = np.stack([c[:, 0] - half[:, 0], c[:, 1] - half[:, 1], c[:, 0] + half[:, 0], c[:, 1] + half[:, 1],
                          c[:, 2] - half[:, 2], c[:, 2] + half[:, 2]], 1).astype(np.float32)
    else:
        boxes = np.stack([c[:, 0] - half[:, 0], c[:, 1] - half[:, 1], c[:, 0] + half[:, 0], c[:, 1] + half[:, 1]], 1).astype(np.float32)
    bix = rng.integers(0, B, size=n).astype(np.int32)
    bix[rng.integers(0, n)] = -1
    if n > 4:
        level[rng.integers(0, n)] = n_levels + 1          # no such level: zero row, no gradient
    return shapes, maps, boxes, bix, level


CASES = [
    # dim, B, C, finest map, n RoIs, pool
    (3, 2, 4, (64, 64, 32), 28, (7, 7, 3)),
    (3, 2, 4, (64, 64, 32), 12, (14, 14, 5)),
    (3, 8, 36, (64, 64, 32), 48, (7, 7, 3)),
    (2, 2, 8, (128, 128), 40, (7, 7)),
    (2, 3, 5, (64, 96), 17, (14, 14)),
    (3, 1, 3, (24, 20, 12), 9, (7, 7, 3)),      # contiguous extents 12/12/3..: not a multiple of 8 -> per-level fallback inside
]


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_pyramid_forward_backward(case, cuda):
    dim, B, C, base, n, pool = case
    shapes, maps, boxes, bix, level = _case(dim, B, C, base, n, seed=sum(base) + n)
    t = lambda a: torch.from_numpy(a).to(cuda)
    maps_t = [t(m).requires_grad_(True) for m in maps]
    out = impl.pyramid_crop_and_resize(maps_t, t(boxes), t(bix), t(level), pool)
    # forward: level by level through the oracle
    want = np.zeros((n, C) + pool, dtype=np.float32)
    for l in range(len(maps)):
        ind = np.where(level == l, bix, -1).astype(np.int32)
        want += oracle.crop_and_resize_forward(maps[l], boxes, ind, pool)
    assert np.array_equal(out.detach().cpu().numpy(), want)

    g = np.random.default_rng(7).normal(size=want.shape).astype(np.float32)
    out.backward(t(g))
    for l in range(len(maps)):
        ind = np.where(level == l, bix, -1).astype(np.int32)
        ref = oracle.crop_and_resize_backward(g, boxes, ind, shapes[l])
        mag = oracle.crop_and_resize_backward(np.abs(g), boxes, ind, shapes[l])
        got = maps_t[l].grad.cpu().numpy()
        assert got.shape == ref.shape
        assert np.all(np.abs(got - ref) <= 2e-6 * mag + 1e-30), (l, np.abs(got - ref).max())
        # and the per-level entry point
        per_level = impl.crop_backward(t(g), t(boxes), t(ind), shapes[l]).cpu().numpy()
        assert np.all(np.abs(got - per_level) <= 4e-6 * mag + 1e-30)


def test_pyramid_backward_is_deterministic_and_overwrites(cuda):
    dim, B, C, base, n, pool = 3, 2, 6, (64, 64, 32), 30, (7, 7, 3)
    shapes, maps, boxes, bix, level = _case(dim, B, C, base, n, seed=3)
    t = lambda a: torch.from_numpy(a).to(cuda)
    g = t(np.random.default_rng(1).normal(size=(n, C) + pool).astype(np.float32))
    first = impl.pyramid_backward(g, t(boxes), t(bix), t(level), shapes)
    first = [f.clone() for f in first]
    for _ in range(3):
        again = impl.pyramid_backward(g, t(boxes), t(bix), t(level), shapes)
        for a, b in zip(first, again):
            assert torch.equal(a, b)


def test_pyramid_bf16_maps(cuda):
    dim, B, C, base, n, pool = 3, 2, 4, (64, 64, 32), 20, (7, 7, 3)
    shapes, maps, boxes, bix, level = _case(dim, B, C, base, n, seed=5)
    t = lambda a: torch.from_numpy(a).to(cuda)
    maps_bf = [t(m).bfloat16() for m in maps]
    with torch.no_grad():
        got = impl.pyramid_crop_and_resize(maps_bf, t(boxes), t(bix), t(level), pool)
        want = impl.pyramid_crop_and_resize([m.float() for m in maps_bf], t(boxes), t(bix), t(level), pool)
    assert torch.equal(got, want)
