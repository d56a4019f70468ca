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
from tests.helpers import random_boxes_3d

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
    (3, 16, 4, (64, 64, 32), 96, (7, 7, 3)),    # batch 16 x 6 RoIs: 64 < N <= 128, multi-chunk RoI scan inside the one launch
    (3, 4, 6, (64, 64, 32), 128, (14, 14, 5)),  # the dispatch limit, ~32 RoIs per element: several rounds per volume
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


def test_pyramid_full_size_adjoint_and_single_level_equivalence(cuda):
    """BASELINE config 3 shapes (128^3 patch, batch 8, 36 channels: maps 32x32x128 .. 4x4x16, 48 sampled RoIs, pool
    (14,14,5)) through size-independent properties: <forward(x), g> == <x, backward(g)> (the backward is the exact
    adjoint of the forward), and every level's gradient map equals what the single-level entry point writes."""
    from tests.helpers import trainlike_rois_3d
    rng = np.random.default_rng(11)
    B, C, pool = 8, 36, (14, 14, 5)
    shapes = [(B, C, 32, 32, 128), (B, C, 16, 16, 64), (B, C, 8, 8, 32), (B, C, 4, 4, 16)]
    per = []
    for li, (side, n) in enumerate(((8.0, 24), (16.0, 12), (32.0, 8), (64.0, 4))):
        tb, ti = trainlike_rois_3d(rng, B, 6, side)
        keep = rng.permutation(len(tb))[:n]
        per.append((tb[keep], ti[keep], np.full(n, li, dtype=np.int32)))
    order = rng.permutation(48)
    boxes = torch.from_numpy(np.concatenate([p[0] for p in per])[order]).to(cuda)
    bix = torch.from_numpy(np.concatenate([p[1] for p in per])[order]).to(cuda)
    level = torch.from_numpy(np.concatenate([p[2] for p in per])[order]).to(cuda)
    gen = torch.Generator(device=cuda).manual_seed(5)
    maps = [torch.randn(s, device=cuda, generator=gen) for s in shapes]
    g = torch.randn((48, C) + pool, device=cuda, generator=gen)
    out = impl.pyramid_forward(maps, boxes, bix, level, pool)
    grads = impl.pyramid_backward(g, boxes, bix, level, shapes)
    lhs = (out.double() * g.double()).sum().item()
    rhs = sum((m.double() * gm.double()).sum().item() for m, gm in zip(maps, grads))
    scale = sum((m.double().abs() * gm.double().abs()).sum().item() for m, gm in zip(maps, grads))
    assert abs(lhs - rhs) <= 2e-6 * scale
    for l, s in enumerate(shapes):
        ind = torch.where(level == l, bix, torch.full_like(bix, -1))
        single = impl.crop_backward(g, boxes, ind, s)
        mag = impl.crop_backward(g.abs(), boxes, ind, s)
        assert torch.all((grads[l] - single).abs() <= 4e-6 * mag + 1e-30)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("crop", [(7, 7, 3), (14, 14, 5), (1, 1, 1)])
def test_channels_last_pyramid_forward_equals_row_major_kernel(crop, dtype, cuda):
    """mdt_pyramid_roi_align_forward_cl on channels_last_3d maps == mdt_pyramid_roi_align_forward on the same maps in row-major storage, bit
    for bit (fp32 and bf16), incl. rows without a level / batch element; gradients flow to the channels-last maps like to row-major ones"""
    from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl as rai
    rng = np.random.default_rng(31)
    g = torch.Generator(device=cuda).manual_seed(31)
    B, C = 3, 36
    shapes = [(B, C, 16, 16, 32), (B, C, 8, 8, 16), (B, C, 4, 4, 8), (B, C, 2, 2, 4)]
    maps = [torch.randn(s, device=cuda, generator=g).to(dtype) for s in shapes]
    maps_cl = [m.contiguous(memory_format=torch.channels_last_3d) for m in maps]
    N = 90
    boxes = torch.from_numpy(random_boxes_3d(rng, N, spill=True)).to(cuda)
    bix = torch.from_numpy(rng.integers(-1, B + 1, size=N).astype(np.int32)).to(cuda)        # incl. out-of-range elements
    lvl = torch.from_numpy(rng.integers(-1, 5, size=N).astype(np.int32)).to(cuda)            # incl. rows without a level
    assert rai.channels_last_eligible(maps_cl, 3) and not rai.channels_last_eligible(maps, 3)
    want = rai.pyramid_forward(maps, boxes, bix, lvl, crop)
    got = rai.pyramid_forward(maps_cl, boxes, bix, lvl, crop, channels_last=True)
    assert torch.equal(got, want)
    if dtype == torch.float32:
        a = [m.clone().requires_grad_(True) for m in maps]
        b = [m.clone(memory_format=torch.preserve_format).requires_grad_(True) for m in maps_cl]
        assert all(t.is_contiguous(memory_format=torch.channels_last_3d) and not t.is_contiguous() for t in b)
        ok_rows = ((bix >= 0) & (bix < B) & (lvl >= 0) & (lvl < 4))
        bx2, bi2, lv2 = boxes[ok_rows][:48], bix[ok_rows][:48], lvl[ok_rows][:48]
        ya = rai.pyramid_crop_and_resize(a, bx2, bi2, lv2, crop)
        yb = rai.pyramid_crop_and_resize(b, bx2, bi2, lv2, crop)
        assert torch.equal(ya, yb)
        w = torch.randn(ya.shape, device=cuda, generator=g)
        (ya * w).sum().backward()
        (yb * w).sum().backward()
        for ta, tb in zip(a, b):
            assert torch.equal(ta.grad, tb.grad)
