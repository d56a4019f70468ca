"""csrc/glue.hip (round 6): the fused loss / target glue against the tensor expressions it replaces (models/mrcnn.py with FUSED_GLUE off --
themselves pinned against the reference's functions by tests/test_glue_parity_gpu.py and the step goldens).  Both forms draw the same
torch.rand keys, so the sampled index sets must be EQUAL, not just equally distributed."""
import ctypes

import numpy as np
import pytest
import torch

from medicaldetectiontoolkit_amd import _lib
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn
from medicaldetectiontoolkit_amd.utils import model_utils as mutils

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_switch():
    prev = mrcnn.FUSED_GLUE
    yield
    mrcnn.FUSED_GLUE = prev


def _boxes(rng, n, dim, lo=0.0, hi=1.0):
    c = rng.uniform(lo, hi, size=(n, dim))
    h = np.exp(rng.uniform(np.log(0.01), np.log(0.6), size=(n, dim))) / 2
    b = np.stack([c[:, 0] - h[:, 0], c[:, 1] - h[:, 1], c[:, 0] + h[:, 0], c[:, 1] + h[:, 1]] + ([c[:, 2] - h[:, 2], c[:, 2] + h[:, 2]] if dim == 3 else []), 1)
    return b.astype(np.float32)


@pytest.mark.parametrize("dim,levels", [(3, [2, 3, 4, 5]), (2, [2, 3, 4, 5]), (3, [1, 2, 3, 4, 5])])
def test_roi_levels_equal_the_tensor_rule(dim, levels, cuda):
    rng = np.random.default_rng(dim * 10 + len(levels))
    n = 700
    b = _boxes(rng, n, dim)
    b[:5] = 0.0                                   # padding rows of the fixed-size glue: zero box
    b[5, 2] = b[5, 0] - 0.1                       # inverted box: sqrt of a negative area
    b[6, :] = [0.0, 0.0, 1.0, 1.0] + ([0.0, 1.0] if dim == 3 else [])      # the whole patch
    bix = rng.integers(-1, 4, size=n).astype(np.float32)
    rois = torch.from_numpy(np.concatenate([b, bix[:, None]], 1)).to(cuda)
    boxes = torch.empty((n, 2 * dim), dtype=torch.float32, device=cuda)
    ints = torch.empty((2, n), dtype=torch.int32, device=cuda)
    rc = _lib.lib().mdt_roi_levels(_lib.ptr(rois), n, dim, levels[0], levels[-1], 1 if len(levels) == 5 else 0, _lib.ptr(boxes), _lib.ptr(ints[0]), _lib.ptr(ints[1]),
                                   _lib.current_stream_ptr())
    _lib.check(rc, "mdt_roi_levels")
    bx = rois[:, :2 * dim]
    h, w = bx[:, 2] - bx[:, 0], bx[:, 3] - bx[:, 1]
    want = (4 + mutils.log2(torch.sqrt(h * w))).round().int().clamp(levels[0], levels[-1])
    if len(levels) == 5:
        want = torch.where(h * w > 0.65, torch.full_like(want, 5), want)
    assert torch.equal(ints[1], want - levels[0])
    assert torch.equal(ints[0], rois[:, 2 * dim].to(torch.int32)) and torch.equal(boxes, bx)
    assert len(torch.unique(ints[1])) >= 3


@pytest.mark.parametrize("A,n_anchor,poolsize,K", [(449280, 6, 10, 2), (60000, 256, 1, 2), (449280, 256, 1, 2), (70001, 6, 20, 3), (300, 6, 10, 2)])
def test_rpn_sampling_and_losses_equal_the_tensor_form(A, n_anchor, poolsize, K, cuda):
    """compute_rpn_losses with mdt_rpn_sample / mdt_anchor_delta_targets == the tensor form on the same random keys: same anchors, same losses.
    Rows: many positives (sub-sampled), few positives, NO positive (neg_count = 1), no negatives at all."""
    B, dim, G = 4, 3, 5
    rng = np.random.default_rng(A % 1000 + n_anchor)
    cf = Configs(dim=3, model="mrcnn", patch_size=[64, 64, 32], batch_size=B, rpn_train_anchors_per_image=n_anchor, shem_poolsize=poolsize)
    match = np.where(rng.uniform(size=(B, A)) < 0.7, -1, 0).astype(np.int32)
    n_pos = [min(A // 4, 400), 2, 0, 7]
    for b in range(B):
        pos = rng.choice(A, size=n_pos[b], replace=False)
        match[b, pos] = rng.integers(1, K, size=n_pos[b]) if K > 2 else 1
    match[3][match[3] == -1] = 0                                   # an element without negatives
    # foreground probabilities without ties near the top (torch.topk and the kernel order tied values differently -- both are valid pools):
    # class-1 logit = a permutation of an even grid, class 0 at 0, further classes 3 below class 1
    logits = np.zeros((B, A, K), dtype=np.float32)
    for b in range(B):
        logits[b, :, 1] = rng.permutation(np.linspace(-4.0, 4.0, A)).astype(np.float32)
    for k in range(2, K):
        logits[:, :, k] = logits[:, :, 1] - 3.0
    deltas = rng.standard_normal((B, A, 2 * dim)).astype(np.float32)
    anchors = np.concatenate([rng.uniform(0, 40, size=(A, 3)), rng.uniform(41, 64, size=(A, 3))], 1)[:, [0, 1, 3, 4, 2, 5]]
    gtb = np.concatenate([rng.uniform(0, 30, size=(B, G, 3)), rng.uniform(31, 64, size=(B, G, 3))], 2)[:, :, [0, 1, 3, 4, 2, 5]]
    argmax = rng.integers(0, G, size=(B, A)).astype(np.int32)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(cuda) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(cuda, dt)

    class Gt(object):
        px = t(gtb, torch.float64)
    out = {}
    for fused in (True, False):
        mrcnn.FUSED_GLUE = fused
        lg = t(logits).requires_grad_(True)
        dl = t(deltas).requires_grad_(True)
        gen = torch.Generator(device=cuda).manual_seed(1234)
        cl, bl, (pidx, pvalid, nidx, nvalid) = mrcnn.compute_rpn_losses(t(match), t(argmax), lg, dl, t(anchors, torch.float64), None, cf, generator=gen, gt_dev=Gt)
        (cl + bl).backward()
        out[fused] = (float(cl), float(bl), pidx.cpu().numpy(), pvalid.cpu().numpy().astype(bool), nidx.cpu().numpy(), nvalid.cpu().numpy().astype(bool),
                      lg.grad.clone(), dl.grad.clone())
    f, w = out[True], out[False]
    assert np.array_equal(f[3], w[3]) and np.array_equal(f[5], w[5]), "validity masks differ"
    for b in range(B):
        assert np.array_equal(f[2][b][f[3][b]], w[2][b][w[3][b]]), ("positives", b)
        assert np.array_equal(f[4][b][f[5][b]], w[4][b][w[5][b]]), ("negatives", b)
    n_pos_max = max(n_anchor // 2, 1)
    assert f[3].sum(1).tolist() == [min(n_pos[b], n_pos_max) for b in range(B)]
    assert f[5][2].sum() == 1 and f[5][3].sum() == 0                 # no positive: one negative; no negatives: none
    assert abs(f[0] - w[0]) <= 1e-6 * max(1.0, abs(w[0])) and abs(f[1] - w[1]) <= 1e-6 * max(1.0, abs(w[1])), (f[:2], w[:2])
    assert torch.allclose(f[6], w[6], rtol=1e-5, atol=1e-8) and torch.allclose(f[7], w[7], rtol=1e-5, atol=1e-8)


def test_rpn_sample_unsupported_sizes_take_the_tensor_form(cuda):
    L = _lib.lib()
    assert L.mdt_rpn_sample_supported(449280, 3, 30) == 1 and L.mdt_rpn_sample_supported(449280, 128, 128) == 1
    assert L.mdt_rpn_sample_supported(449280, 129, 30) == 0 and L.mdt_rpn_sample_supported(10, 3, 30) == 0
    assert L.mdt_rpn_sample_supported(3000000, 128, 128) == 0          # 3 M anchors x 128 candidates do not fit the merge block


@pytest.mark.parametrize("dim,pc,rois_per_image,poolsize", [(3, 75, 6, 10), (3, 40, 40, 1), (2, 500, 6, 10), (3, 75, 64, 2)])
def test_detection_target_layer_equals_the_tensor_form(dim, pc, rois_per_image, poolsize, cuda):
    B, G = 4, 6
    rng = np.random.default_rng(pc + rois_per_image)
    patch = [64, 64, 32] if dim == 3 else [64, 64]
    cf = Configs(dim=dim, model="mrcnn", patch_size=patch, batch_size=B, train_rois_per_image=rois_per_image, shem_poolsize=poolsize)
    scale = np.asarray(cf.scale, dtype=np.float64)
    gt_boxes, gt_cls = [], []
    for b in range(B):
        n = [3, 1, 0, 6][b]
        gb = _boxes(rng, n, dim, 0.2, 0.8).astype(np.float64) * scale
        gt_boxes.append(gb)
        gt_cls.append(rng.integers(1, 3, size=n))
    props = []
    for b in range(B):
        pb = _boxes(rng, pc, dim)
        k = min(pc // 3, 12 * max(len(gt_boxes[b]), 1))
        for j in range(k if len(gt_boxes[b]) else 0):                      # jittered copies of GT boxes: positives
            pb[j] = (gt_boxes[b][j % len(gt_boxes[b])] / scale + rng.normal(0, 0.01, size=2 * dim)).astype(np.float32)
        props.append(np.concatenate([pb, np.full((pc, 1), b, np.float32)], 1))
    props = torch.from_numpy(np.concatenate(props, 0)).to(cuda)
    scores = torch.softmax(torch.from_numpy(rng.standard_normal((B * pc, 3)).astype(np.float32)), 1).to(cuda)
    shape = tuple(patch)
    masks = torch.from_numpy((rng.uniform(size=(sum(len(g) for g in gt_boxes), 1) + shape) > 0.5).astype(np.uint8)).to(cuda)
    out = {}
    for fused in (True, False):
        mrcnn.FUSED_GLUE = fused
        gen = torch.Generator(device=cuda).manual_seed(99)
        out[fused] = [o.clone() for o in mrcnn.detection_target_layer(props, scores, gt_cls, gt_boxes, masks, cf, B, generator=gen)]
    (si, va, ip, tc, td, tm), (si2, va2, ip2, tc2, td2, tm2) = out[True], out[False]
    assert torch.equal(va, va2) and torch.equal(ip, ip2)
    assert int(ip.sum()) >= 3 and int((va & ~ip).sum()) >= 3
    assert torch.equal(si[va], si2[va2]) and torch.equal(tc[va].long(), tc2[va2].long())
    assert torch.equal(td, td2), float((td - td2).abs().max())
    assert torch.equal(tm[va], tm2[va2]) and float(tm[~ip].abs().max()) == 0.0
    S = va.numel() // B
    assert int(va.view(B, S)[2].sum()) == 1 and int(ip.view(B, S)[2].sum()) == 0       # element without GT: one negative, no positive


@pytest.mark.parametrize("dim,patch", [(3, [64, 64, 32]), (2, [64, 64])])
def test_rpn_at_anchors_fused_gather_equals_the_tensor_form(dim, patch, cuda):
    """rpn_at_anchors through mdt_rpn_patch_gather / _scatter_add == the index-arithmetic tensor form: logits, deltas bit-equal (the same rows
    enter the same matrix products), gradients w.r.t. the maps and the RPN parameters equal (scatter order differs: 1e-6)"""
    B, n = 3, 40
    cf = Configs(dim=dim, model="mrcnn", patch_size=patch, batch_size=B)
    torch.manual_seed(0)
    net = mrcnn.net(cf, device=cuda)
    mf = torch.channels_last_3d if dim == 3 else torch.channels_last
    g = torch.Generator(device=cuda).manual_seed(7)
    shapes = [tuple(int(v) for v in s) for s in cf.backbone_shapes]
    levels = [shapes[i] for i in cf.pyramid_levels]
    A = len(cf.rpn_anchor_ratios) * (1 if dim == 2 else len(cf.rpn_anchor_scales["z"][0])) if False else None
    n_a = net.anchors.shape[0] // sum(int(np.prod(s)) for s in levels)
    total = sum(int(np.prod(s)) for s in levels) * n_a
    idx = torch.randint(0, total, (B, n), device=cuda, generator=g)
    idx[0, :4] = torch.tensor([0, n_a * int(np.prod(levels[0])) - 1, n_a * int(np.prod(levels[0])), total - 1], device=cuda)     # level borders, map corners
    res = {}
    for fused in (True, False):
        mrcnn.FUSED_GLUE = fused
        maps = [torch.randn((B, cf.end_filts) + s, device=cuda, generator=torch.Generator(device=cuda).manual_seed(50 + k)).contiguous(memory_format=mf).requires_grad_(True)
                for k, s in enumerate(levels)]
        net.zero_grad()
        lg, dl = mrcnn.rpn_at_anchors(net.rpn, maps, idx, n_a)
        wl = torch.randn(lg.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(3))
        wd = torch.randn(dl.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(4))
        ((lg * wl).sum() + (dl * wd).sum()).backward()
        res[fused] = (lg.detach(), dl.detach(), [m.grad.clone() for m in maps], [p.grad.clone() for p in net.rpn.parameters()])
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for a, b in zip(res[True][2] + res[True][3], res[False][2] + res[False][3]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max() + 1e-12)), float((a - b).abs().max())
    assert all(float(gm.abs().sum()) > 0 for gm in res[True][2][:2])
    # the scatter is deterministic (one writer per voxel, rows added in row order): a second backward gives the same bits although the 120 sampled
    # neighbourhoods overlap heavily on the small upper levels
    mrcnn.FUSED_GLUE = True
    again = []
    for _ in range(2):
        maps = [torch.randn((B, cf.end_filts) + s, device=cuda, generator=torch.Generator(device=cuda).manual_seed(50 + k)).contiguous(memory_format=mf).requires_grad_(True)
                for k, s in enumerate(levels)]
        lg, dl = mrcnn.rpn_at_anchors(net.rpn, maps, idx, n_a)
        ((lg * wl).sum() + (dl * wd).sum()).backward()
        again.append([m.grad.clone() for m in maps])
    assert all(torch.equal(a, b) for a, b in zip(*again)) and all(torch.equal(a, b) for a, b in zip(again[0], res[True][2]))


@pytest.mark.parametrize("dim,pc,case", [(3, 75, "normal"), (3, 75, "nothing_confident"), (2, 500, "normal"), (3, 40, "one_element_empty")])
def test_refine_detections_fused_equals_the_tensor_form(dim, pc, case, cuda):
    """refine_detections through mdt_refine_detections_pre / the batched NMS / _post == the tensor form: same rows (as sets per element: top-k
    order of equal scores aside there are none), same validity, incl. the reference's keep-index-0 quirk when nothing is confident"""
    B = 3
    rng = np.random.default_rng(pc + len(case))
    patch = [64, 64, 32] if dim == 3 else [64, 64]
    cf = Configs(dim=dim, model="mrcnn", patch_size=patch, batch_size=B)
    rois = torch.from_numpy(_boxes(rng, B * pc, dim, 0.1, 0.9)).to(cuda)
    logits = rng.standard_normal((B * pc, 3)).astype(np.float32) * 1.5
    if case == "nothing_confident":
        logits[:, 0] += 9.0
    if case == "one_element_empty":
        logits[pc:2 * pc, 0] += 9.0
    probs = torch.softmax(torch.from_numpy(logits), 1).to(cuda)
    deltas = torch.from_numpy((rng.standard_normal((B * pc, 3, 2 * dim)) * 0.5).astype(np.float32)).to(cuda)
    bix = torch.arange(B, device=cuda, dtype=torch.float32).repeat_interleave(pc)
    out = {}
    for fused in (True, False):
        mrcnn.FUSED_GLUE = fused
        r, v = mrcnn.refine_detections(rois, probs, deltas, bix, cf, B)
        out[fused] = (r.clone(), v.clone())
    (rf, vf), (rt, vt) = out[True], out[False]
    assert rf.shape == rt.shape and torch.equal(vf, vt)
    M = rf.shape[0] // B
    for b in range(B):
        a = rf[b * M:(b + 1) * M][vf[b * M:(b + 1) * M]].cpu().numpy()
        c = rt[b * M:(b + 1) * M][vt[b * M:(b + 1) * M]].cpu().numpy()
        a = a[np.lexsort(a.T[::-1])]
        c = c[np.lexsort(c.T[::-1])]
        assert np.array_equal(a, c), (b, a[:3], c[:3])
    assert float(rf[~vf].abs().max() if (~vf).any() else 0.0) == 0.0
    if case == "nothing_confident":
        assert int(vf.sum()) == 1 and bool(vf[0]) and float(rf[0, 2 * dim + 1]) == 1.0 and float(rf[0, 2 * dim + 2]) < cf.model_min_confidence
    elif case == "one_element_empty":
        assert int(vf[M:2 * M].sum()) == 0 and int(vf.sum()) > 0
    else:
        assert int(vf.sum()) >= B


def test_shared_pyramid_gradient_equals_autograd_sum(cuda):
    """PyramidGradAccumulator (round 6): the classifier head's and the mask head's RoIAlign backward and the sampled-anchor RPN scatter land in
    ONE buffer per pyramid map (write once, accumulate, scatter) == the three dense gradients autograd adds.  Whole training step, both
    forms from the same weights and batch: every loss term equal, every parameter gradient within fp32 summation order."""
    from medicaldetectiontoolkit_amd.utils.synthetic_data import batch_with_gt_from_proposals, make_batch, to_device
    patch, B = [64, 64, 32], 2
    cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B, channels_last=True)
    torch.manual_seed(0)
    net = mrcnn.net(cf, device=cuda)
    b = batch_with_gt_from_proposals(net, cf, to_device(make_batch(patch, B, seed=5), cuda), cuda)
    prev = mrcnn.SHARED_PYRAMID_GRAD
    res = {}
    try:
        for shared in (True, False):
            mrcnn.SHARED_PYRAMID_GRAD = shared
            torch.manual_seed(11)
            net.zero_grad(set_to_none=True)
            out = net.train_forward(b, monitor=False)
            out["torch_loss"].backward()
            acc = net._pyramid_grad_acc
            assert (acc is not None) == shared
            if shared:
                assert acc.expected == 3 and acc.arrived == 3
                acc.check()
            res[shared] = ({k: float(v) for k, v in out["loss_terms"].items()}, {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None},
                           [int(v) for v in out["sample_counts"]])
    finally:
        mrcnn.SHARED_PYRAMID_GRAD = prev
    assert res[True][0] == res[False][0] and res[True][2] == res[False][2] and res[True][2][1] > 0
    assert set(res[True][1]) == set(res[False][1])
    for n, g in res[True][1].items():
        w = res[False][1][n]
        assert torch.allclose(g, w, rtol=1e-4, atol=1e-6 * float(w.abs().max() + 1e-12)), (n, float((g - w).abs().max()), float(w.abs().max()))
