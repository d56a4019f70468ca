"""Host-side packing logic that needs no GPU."""
import numpy as np
import torch


def test_gt_on_device_packing_matches_padded_tables():
    """models/mrcnn.GtOnDevice (one upload for all GT boxes / class ids) against a direct construction of the padded
    tables the target layer and the RPN losses consume (mrcnn.py:487-500 semantics: elements without a foreground class id
    are treated as having no GT in the target layer but keep their boxes for the anchor matching)."""
    from medicaldetectiontoolkit_amd.models.mrcnn import GtOnDevice
    rng = np.random.default_rng(0)
    boxes = [rng.uniform(0, 60, size=(3, 6)), np.zeros((0, 6)), rng.uniform(0, 60, size=(1, 6)), rng.uniform(0, 60, size=(2, 6))]
    cls = [np.array([1, 2, 1]), np.array([], dtype=np.int64), np.array([0]), np.array([2, 1])]
    g = GtOnDevice(boxes, cls, 3, torch.device("cpu"))
    assert g.n_all == [3, 0, 1, 2] and g.counts == [3, 0, 0, 2]
    assert g.px.shape == (4, 3, 6) and g.px.dtype == torch.float64 and g.px.is_contiguous()
    for b in range(4):
        n = len(boxes[b])
        assert np.array_equal(g.px[b, :n].numpy(), boxes[b])
        assert (g.px[b, n:] == 0).all()
        assert g.cls[b, :n].tolist() == cls[b].tolist() and g.cls_i32.dtype == torch.int32
    assert g.valid.tolist() == [[True, True, True], [False] * 3, [False] * 3, [True, True, False]]
    # index into the stacked GT masks: runs over ALL objects of the batch, -1 where not valid
    assert g.gidx.tolist() == [[0, 1, 2], [-1, -1, -1], [-1, -1, -1], [4, 5, -1]]


def test_const_tensor_is_cached_per_value_dtype_device():
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    a = mutils.const_tensor([0.1, 0.2], torch.float32, torch.device("cpu"))
    b = mutils.const_tensor([0.1, 0.2], torch.float32, torch.device("cpu"))
    c = mutils.const_tensor([0.1, 0.2], torch.float64, torch.device("cpu"))
    d = mutils.const_tensor([[0.1], [0.2]], torch.float32, torch.device("cpu"))
    assert a is b and c is not a and c.dtype == torch.float64 and d.shape == (2, 1) and d is not a
    assert torch.equal(a, torch.tensor([0.1, 0.2]))


def test_conv_reformulations_are_exact_identities_in_float64():
    """the two MIOpen-problem reformulations of utils/fused_epilogue (input gradient as a forward convolution with the
    flipped/transposed filter; stem in space-to-depth form) are identities: float64 on the CPU shows it to 1e-12"""
    import torch.nn.functional as F
    from medicaldetectiontoolkit_amd.utils.fused_epilogue import _ConvStem221, _ConvStride1
    torch.manual_seed(0)
    for nd, ks, pad in ((3, 3, 1), (3, 1, 0), (2, 3, 1), (2, 5, 2)):
        x = torch.randn((2, 5) + (9, 8, 7)[:nd], dtype=torch.float64, requires_grad=True)
        w = torch.randn((4, 5) + (ks,) * nd, dtype=torch.float64, requires_grad=True)
        conv = F.conv3d if nd == 3 else F.conv2d
        y1, y2 = _ConvStride1.apply(x, w, (pad,) * nd), conv(x, w, None, 1, pad)
        g = torch.randn_like(y2)
        a, b = torch.autograd.grad(y1, (x, w), g), torch.autograd.grad(y2, (x, w), g)
        assert torch.equal(y1, y2) and (a[0] - b[0]).abs().max() < 1e-12 and (a[1] - b[1]).abs().max() < 1e-12
    for cin, k in ((1, 7), (2, 3), (3, 5)):
        x = torch.randn(2, cin, 16, 12, 10, dtype=torch.float64, requires_grad=True)
        w = torch.randn(5, cin, k, k, k, dtype=torch.float64, requires_grad=True)
        y1, y2 = _ConvStem221.apply(x, w), F.conv3d(x, w, None, (2, 2, 1), k // 2)
        g = torch.randn_like(y2)
        a, b = torch.autograd.grad(y1, (x, w), g), torch.autograd.grad(y2, (x, w), g)
        assert (y1 - y2).abs().max() < 1e-12 and (a[0] - b[0]).abs().max() < 1e-12 and (a[1] - b[1]).abs().max() < 1e-12


# ---------------------------------------------------------------------------------------------------------------------------
# round 3: staging of numpy batches, flat views of the optimizer, profile post-processing (no GPU: pinned memory and events
# are replaced by plain tensors / counters)
class _FakeEvent(object):
    waits = 0
    records = 0

    def record(self):
        _FakeEvent.records += 1

    def synchronize(self):
        _FakeEvent.waits += 1

    def query(self):              # "not finished yet": the stager must wait (and account the wait as back-pressure)
        return False


def _patch_pinned(monkeypatch):
    import torch
    real_empty = torch.empty

    def empty(*a, **k):
        k.pop("pin_memory", None)
        return real_empty(*a, **k)

    monkeypatch.setattr(torch, "empty", empty)
    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    _FakeEvent.waits = _FakeEvent.records = 0


def test_pinned_stager_reuses_slots_after_their_own_event(monkeypatch):
    """ring of 3: a slot is handed out again only after waiting for the event recorded at its last release; buffers grow, never
    shrink, and are not re-allocated for a smaller request"""
    import torch
    from medicaldetectiontoolkit_amd.utils import model_utils as mu
    _patch_pinned(monkeypatch)
    st = mu.PinnedStager(depth=3)
    ptrs = []
    for i in range(7):
        k, raw = st.acquire(1000 + 10 * i)
        assert raw.dtype == torch.uint8 and raw.numel() == 1000 + 10 * i
        ptrs.append((k, raw.data_ptr()))
        st.release(k)
    assert [k for k, _ in ptrs] == [0, 1, 2, 0, 1, 2, 0]
    assert _FakeEvent.records == 7 and _FakeEvent.waits == 4          # the first round finds no event to wait for
    assert ptrs[3][1] == ptrs[0][1] and ptrs[6][1] == ptrs[0][1]      # 25 % head room: no re-allocation for slightly larger requests
    k, raw = st.acquire(10)
    assert raw.data_ptr() == ptrs[1][1]


def test_stack_into_equals_cat_for_every_dtype_and_ragged_parts(monkeypatch):
    import numpy as np
    import torch
    from medicaldetectiontoolkit_amd.utils import model_utils as mu
    _patch_pinned(monkeypatch)
    rng = np.random.default_rng(0)
    for dtype in (np.uint8, np.float32, np.int64, np.float64):
        parts = [torch.from_numpy(rng.integers(0, 200, size=(n, 1, 5, 6, 7)).astype(dtype)) for n in (3, 1, 4)]
        nbytes = sum(p.numel() * p.element_size() for p in parts)
        raw = torch.empty(nbytes + 64, dtype=torch.uint8)[:nbytes]
        got = mu._stack_into(raw, parts)
        assert got.dtype == parts[0].dtype and torch.equal(got, torch.cat(parts, 0))
    # the parallel path (parts of 32 MB and more are split over the pool) gives the same bytes
    big = torch.from_numpy(rng.standard_normal((9, 1, 100, 100, 100)).astype(np.float32))
    raw = torch.empty(big.numel() * 4, dtype=torch.uint8)
    assert torch.equal(mu._stack_into(raw, [big], pool=mu._stage_pool(), split=3), big)


def test_staged_upload_and_upload_on_cpu_are_plain_copies():
    """device 'cpu' (the CPU test tier): no pinned memory, no thread, same values; empty part lists give None like the
    `masks_list` branch they replace"""
    import numpy as np
    import torch
    from medicaldetectiontoolkit_amd.utils import model_utils as mu
    parts = [np.arange(2 * 24, dtype=np.uint8).reshape(2, 1, 2, 3, 4), np.ones((1, 1, 2, 3, 4), dtype=np.uint8)]
    su = mu.StagedUpload(parts, "cpu")
    assert su.future is None and torch.equal(su.get(), torch.cat([torch.from_numpy(p) for p in parts], 0))
    assert mu.StagedUpload([], "cpu").get() is None
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    assert torch.equal(mu.upload(a, "cpu"), torch.from_numpy(a))


def test_flat_views_keep_shape_strides_and_slot_order():
    """training._view_like: element k of a parameter's storage is slot off + k of the flat buffer, for contiguous and for dense
    permuted (channels-last) parameters alike -- what lets Adam run elementwise over the flat buffers"""
    import torch
    from medicaldetectiontoolkit_amd import training
    w_cl = torch.arange(5 * 2 * 27, dtype=torch.float32).reshape(5, 2, 3, 3, 3).contiguous(memory_format=torch.channels_last_3d)
    w_c = torch.arange(7 * 3, dtype=torch.float32).reshape(7, 3)
    assert training._dense(w_cl) and not w_cl.is_contiguous()
    flat = torch.zeros(w_cl.numel() + w_c.numel() + 5)
    v1 = training._view_like(flat, 3, w_cl)
    v2 = training._view_like(flat, 3 + w_cl.numel(), w_c)
    assert v1.shape == w_cl.shape and v1.stride() == w_cl.stride() and v2.stride() == w_c.stride()
    v1.copy_(w_cl)
    v2.copy_(w_c)
    # storage order of the parameter == slot order in the flat buffer
    assert torch.equal(flat[3:3 + w_cl.numel()], w_cl.permute(0, 2, 3, 4, 1).reshape(-1))
    assert torch.equal(flat[3 + w_cl.numel():3 + w_cl.numel() + w_c.numel()], w_c.reshape(-1))
    assert float(flat[:3].abs().sum()) == 0.0 and float(flat[-2:].abs().sum()) == 0.0
    assert not training._dense(torch.zeros(4, 6)[:, ::2])


def test_steady_state_categories_cover_this_repos_kernels():
    """tools/steady_state.category: every kernel name of libmdt_hip lands in one of the two 'this repo' rows, MIOpen / CK / rocBLAS
    names in the convolution row (the split the DESIGN tables quote)"""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("steady_state", os.path.join(root, "tools", "steady_state.py"))
    ss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ss)
    names = set()
    csrc = os.path.join(root, "medicaldetectiontoolkit_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith(".hip"):
            names |= set(re.findall(r"__global__[^;{]*?void\s+(\w+)\s*\(", open(os.path.join(csrc, f)).read()))
    assert len(names) > 30
    for n in names:
        assert "this repo" in ss.category("(anonymous namespace)::" + n), n
    for n in ("_ZN2ck16tensor_operation6device", "Cijk_Alik_Bljk_S_B_Bias_HA_S_SAV_UserArgs_MT256x48", "miopenSp3AsmConv"):
        assert ss.category(n).startswith("MIOpen")
    assert ss.category("void at::native::vectorized_elementwise_kernel<4, ...>").startswith("torch elementwise")


def test_rpn_at_anchors_equals_the_dense_rpn_in_float64():
    """models/mrcnn.rpn_at_anchors (the RPN of mrcnn.py:40-86 re-evaluated at chosen anchors: 3^dim x C neighbourhood gather -> matrix
    products) == gathering the dense RPN outputs at the same anchors, and so are ALL gradients (feature maps of every level, the six
    RPN parameters): volume corners (zero padding), first / last anchor of a level, duplicates, 2D and 3D; float64, 1e-12"""
    import pytest
    from medicaldetectiontoolkit_amd.configs import Configs
    from medicaldetectiontoolkit_amd.models import mrcnn
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    torch.manual_seed(0)
    for dim, sizes in ((3, [(8, 8, 16), (4, 4, 8), (2, 2, 4), (1, 1, 2)]), (2, [(16, 16), (8, 8), (4, 4), (2, 2)])):
        cf = Configs(dim=dim, model="mrcnn", patch_size=[32] * dim, batch_size=2)
        rpn = mrcnn.RPN(cf, mutils.NDConvGenerator(dim)).double()
        assert mrcnn.rpn_sparse_supported(rpn)
        B, n_apv = 2, len(cf.rpn_anchor_ratios)
        mf = torch.channels_last_3d if dim == 3 else torch.channels_last
        for channels_last in (True, False):
            maps = [torch.randn((B, cf.end_filts) + s, dtype=torch.float64) for s in sizes]
            maps = [(m.contiguous(memory_format=mf) if channels_last else m).requires_grad_(True) for m in maps]
            outs = [rpn(m) for m in maps]
            logits, deltas = torch.cat([o[0] for o in outs], 1), torch.cat([o[2] for o in outs], 1)
            A = logits.shape[1]
            first = n_apv * int(np.prod(sizes[0]))
            idx = torch.randint(0, A, (B, 40))
            idx[0, :4] = torch.tensor([0, A - 1, first - 1, first])
            idx[1, :3] = idx[1, 3]                                                  # duplicates accumulate in the scatter
            ls, ds = mrcnn.rpn_at_anchors(rpn, maps, idx, n_apv)
            dl = torch.gather(logits, 1, idx.unsqueeze(-1).expand(-1, -1, 2))
            dd = torch.gather(deltas, 1, idx.unsqueeze(-1).expand(-1, -1, 2 * dim))
            assert float((ls - dl).abs().max()) < 1e-12 and float((ds - dd).abs().max()) < 1e-12
            wl, wd = torch.randn_like(dl), torch.randn_like(dd)
            wrt = maps + list(rpn.parameters())
            g_dense = torch.autograd.grad((dl * wl).sum() + (dd * wd).sum(), wrt, retain_graph=True)
            g_sparse = torch.autograd.grad((ls * wl).sum() + (ds * wd).sum(), wrt)
            for a, b in zip(g_dense, g_sparse):
                assert a.shape == b.shape and float((a - b).abs().max()) < 1e-11 * max(1.0, float(a.abs().max()))
    # a normalisation layer or a strided conv_shared is outside the restatement: the dense graph is used
    cf = Configs(dim=3, model="mrcnn", patch_size=[32, 32, 32], batch_size=2)
    cf.rpn_anchor_stride = 2
    assert not mrcnn.rpn_sparse_supported(mrcnn.RPN(cf, mutils.NDConvGenerator(3)))


def test_stride_tap_equals_three_consumers_of_a_stage_output_in_float64():
    """utils/fused_epilogue._StrideTap + ResBlock.forward_subsampled (a stage output sub-sampled ONCE for the two strided 1x1 layers of the
    next stage's first block, the FPN lateral reading the tap's alias; the sub-sampled gradient added IN PLACE into the lateral's) ==
    the plain graph of models/backbone.py:128-153 / 183-206 with its three consumers: block output, lateral output, the input gradient
    and every parameter gradient; odd extents, 2D and 3D; float64 1e-13.  Single-consumer cases (either gradient absent) as well."""
    from medicaldetectiontoolkit_amd.models.backbone import ResBlock
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe, model_utils as mutils
    torch.manual_seed(0)
    for dim, shape in ((3, (2, 8, 8, 6, 10)), (3, (1, 8, 7, 5, 9)), (2, (2, 8, 9, 7))):
        conv = mutils.NDConvGenerator(dim)
        blk = ResBlock(8, 4, conv=conv, stride=2, downsample=(8, 2, 2)).double()
        lat = conv(8, 5, ks=1, stride=1, relu=None).double()
        mf = torch.channels_last_3d if dim == 3 else torch.channels_last
        x0 = torch.randn(shape, dtype=torch.float64).contiguous(memory_format=mf)
        res = []
        for tap in (False, True):
            x = x0.clone().requires_grad_(True)
            h = x * 1.0
            blk.zero_grad()
            lat.zero_grad()
            if tap:
                x_lat, x_s = fe.stride_tap(h, blk.conv1[0].stride)
                out = blk.forward_subsampled(x_s)
            else:
                x_lat, out = h, blk(h)
            lo = lat(x_lat)
            ((out * out).sum() + (lo * lo * lo).sum()).backward()
            res.append([out.detach(), lo.detach(), x.grad.clone()] + [p.grad.clone() for p in list(blk.parameters()) + list(lat.parameters())])
        for a, b in zip(*res):
            assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-13 * max(1.0, float(a.abs().max()))
        x = x0.clone().requires_grad_(True)
        _, x_s = fe.stride_tap(x * 1.0, (2,) * dim)
        x_s.sum().backward()
        sl = (slice(None), slice(None)) + (slice(None, None, 2),) * dim
        want = torch.zeros_like(x0)
        want[sl] = 1.0
        assert torch.equal(x.grad, want)
        x.grad = None
        x_lat, _ = fe.stride_tap(x * 1.0, (2,) * dim)
        (x_lat * 3.0).sum().backward()
        assert torch.equal(x.grad, torch.full_like(x0, 3.0))


def test_channels_last_nearest_upsampling_equals_f_interpolate():
    """utils/fused_epilogue.upsample_nearest (channels-last broadcast copy forward, strided sum over the replicas backward: the FPN's
    top-down path without the two layout conversions torch's row-major nearest kernels need) == F.interpolate(mode='nearest') bit for
    bit forward, 1e-14 backward (float64), result and gradient channels-last; factors 2, (2, 2, 1), 3; 2D and 3D; row-major input
    falls through to F.interpolate"""
    import torch.nn.functional as F
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    torch.manual_seed(0)
    for shape, sc in (((2, 5, 3, 4, 6), 2), ((1, 3, 2, 2, 5), (2, 2, 1)), ((2, 5, 3, 4), 2), ((2, 4, 3, 3, 2), 3)):
        mf = torch.channels_last_3d if len(shape) == 5 else torch.channels_last
        x0 = torch.randn(shape, dtype=torch.float64).contiguous(memory_format=mf)
        res = []
        for own in (False, True):
            x = x0.clone().requires_grad_(True)
            h = x * 1.0
            y = fe.upsample_nearest(h, sc) if own else F.interpolate(h, scale_factor=sc)
            assert not own or y.is_contiguous(memory_format=mf)
            w = torch.randn(y.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
            (y * w).sum().backward()
            res.append((y.detach(), x.grad.clone()))
        assert torch.equal(res[0][0], res[1][0])
        assert float((res[0][1] - res[1][1]).abs().max()) < 1e-13 and res[1][1].is_contiguous(memory_format=mf)
    xr = torch.randn(2, 5, 3, 4, 6)
    assert torch.equal(fe.upsample_nearest(xr, 2), F.interpolate(xr, scale_factor=2))


def test_rpn_losses_through_sampled_anchors_equal_the_dense_graph_losses():
    """models/mrcnn.compute_rpn_losses with sparse_eval = rpn_at_anchors (dense RPN outputs detached: they only rank and select) ==
    the same function on the dense RPN graph (mrcnn.py:176-240 on the outputs of :987-1003): class loss, box loss, the sampled
    anchors, and the gradients w.r.t. every pyramid map and every RPN parameter; float64, same generator seed for the random keys"""
    from medicaldetectiontoolkit_amd.configs import Configs
    from medicaldetectiontoolkit_amd.models import mrcnn
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    torch.manual_seed(1)
    dim, B = 3, 2
    cf = Configs(dim=dim, model="mrcnn", patch_size=[32, 32, 32], batch_size=B)
    sizes = [(8, 8, 16), (4, 4, 8), (2, 2, 4), (1, 1, 2)]
    rpn = mrcnn.RPN(cf, mutils.NDConvGenerator(dim)).double()
    n_apv = len(cf.rpn_anchor_ratios)
    A = n_apv * sum(int(np.prod(s)) for s in sizes)
    anchors = torch.rand(A, 2 * dim, dtype=torch.float64) * 10
    anchors[:, [2, 3, 5]] = anchors[:, [0, 1, 4]] + 4 + torch.rand(A, 3, dtype=torch.float64) * 8
    match = torch.full((B, A), 0, dtype=torch.int32)
    match[:, ::7] = -1
    match[0, [5, 700, A - 1]] = 1
    match[1, [3 * int(np.prod(sizes[0])), 11]] = 1
    argmax = torch.zeros((B, A), dtype=torch.int32)
    gt = [np.array([[2.0, 3.0, 14.0, 12.0, 1.0, 9.0]]), np.array([[1.0, 1.0, 9.0, 7.0, 2.0, 8.0]])]
    res = []
    for sparse in (False, True):
        maps = [torch.randn((B, cf.end_filts) + s, dtype=torch.float64, generator=torch.Generator().manual_seed(7 + i))
                .contiguous(memory_format=torch.channels_last_3d).requires_grad_(True) for i, s in enumerate(sizes)]
        rpn.zero_grad()
        with torch.set_grad_enabled(not sparse):
            outs = [rpn(m) for m in maps]
            logits, deltas = torch.cat([o[0] for o in outs], 1), torch.cat([o[2] for o in outs], 1)
        ev = (lambda idx: mrcnn.rpn_at_anchors(rpn, maps, idx, n_apv)) if sparse else None
        gen = torch.Generator().manual_seed(3)
        cl, bl, samples = mrcnn.compute_rpn_losses(match, argmax, logits, deltas, anchors, gt, cf, generator=gen, sparse_eval=ev)
        (cl + 2.0 * bl).backward()
        res.append((float(cl), float(bl), [t.clone() for t in samples], [m.grad.clone() for m in maps] + [p.grad.clone() for p in rpn.parameters()]))
    (ca, ba, sa, ga), (cb, bb, sb, gb) = res
    assert abs(ca - cb) < 1e-12 and abs(ba - bb) < 1e-12 and ca > 0 and ba > 0
    assert all(torch.equal(a, b) for a, b in zip(sa, sb)) and int(sa[1].sum()) == 5
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 1e-12 * max(1.0, float(a.abs().max()))
    assert any(float(g.abs().max()) > 0 for g in ga[:4])


def test_rank_core_plan_is_disjoint_and_numa_local():
    """utils/affinity.plan: ranks sharing a NUMA node split its cores evenly; unknown topology -> even split of the allowed cores; the slices
    of different ranks never overlap; more ranks than cores degrade to sharing instead of an empty set"""
    from medicaldetectiontoolkit_amd.utils import affinity
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(128))
    nodes = {0: list(range(0, 64)), 1: list(range(64, 128))}
    numa = [0, 0, 0, 0, 1, 1, 1, 1]
    slices = [affinity.plan(r, 8, allowed, numa, nodes) for r in range(8)]
    assert all(len(s) == 16 for s in slices)
    assert all(set(s) <= set(nodes[numa[r]]) for r, s in enumerate(slices))
    assert len(set().union(*map(set, slices))) == 128
    # topology not exposed (numa_node = -1 in a container): even split
    slices = [affinity.plan(r, 4, allowed, [None] * 4, nodes) for r in range(4)]
    assert [len(s) for s in slices] == [32] * 4 and len(set().union(*map(set, slices))) == 128
    # a restricted cpuset (cgroup): only allowed cores are handed out
    slices = [affinity.plan(r, 2, [4, 5, 6, 7], [0, 0], nodes) for r in range(2)]
    assert slices == [[4, 5], [6, 7]]
    assert affinity.plan(5, 8, [0, 1], None, None) == [1]
    rec = affinity.pin_rank(0, 1, [0], set_torch_threads=False)
    assert rec["pinned"] and rec["cores"] >= 1


def test_bench_step_form_text_builds_for_both_models():
    """bench.py's JSON line is assembled after minutes of GPU work: the one large string expression in it must at least evaluate (a stray
    operator there once took the whole line down on the GPU box)"""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    ast.parse(src)
    i = src.index('"step_form": (') + len('"step_form": ')
    depth = 0
    for j in range(i, len(src)):
        if src[j] == "(":
            depth += 1
        elif src[j] == ")":
            depth -= 1
            if depth == 0:
                break

    class A(object):
        model, sparse_rpn_loss, step_form = "mrcnn", 1, "exec"
    for model in ("mrcnn", "retina_unet"):
        for form in ("exec", "no-readout"):
            A.model, A.step_form = model, form
            text = eval(src[i:j + 1], {"args": A})
            assert isinstance(text, str) and "exec.py:68-74" in text
            # the headline IS the step exec.py runs -- no "NOT in `value`" clause in its description
            assert "NOT in" not in text
            assert ("no_readout_step" in text) == (form == "exec")
            assert ("WITHOUT the per-batch read-out" in text) == (form == "no-readout")


def _stride_tap_case(mode):
    """x -> _StrideTap -> (lateral 1x1 convolution, strided consumer); returns x.grad, what a hook on the lateral alias kept, the reference
    gradients, and how the node's in-place / cloned counters moved"""
    import torch
    import torch.nn.functional as F
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe

    class PyConv1x1(torch.autograd.Function):          # a Python producer of the lateral's gradient, like fused_epilogue._ConvStride1
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return F.conv3d(x, w)

        @staticmethod
        def backward(ctx, gy):
            x, w = ctx.saved_tensors
            gx = torch.einsum("bodhw,oc->bcdhw", gy, w[:, :, 0, 0, 0]).contiguous(memory_format=torch.channels_last_3d)
            return gx, torch.einsum("bodhw,bcdhw->oc", gy, x)[:, :, None, None, None]

    torch.manual_seed(0)
    x = torch.randn(1, 4, 8, 8, 8).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w1, w2 = torch.randn(3, 4, 1, 1, 1), torch.randn(5, 4, 1, 1, 1)
    fe._tap_owned_refs()
    before = list(fe._TAP_REFS)
    kept = []
    lat, xs = fe.stride_tap(x * 1.0, (2, 2, 2))
    if mode == "hook_keeps_gradient":
        lat.register_hook(lambda g: kept.append(g))
    if mode == "retain_grad":
        lat.retain_grad()
    a = PyConv1x1.apply(lat, w1) if mode == "python_producer" else F.conv3d(lat, w1)
    (a.sum() + F.conv3d(xs, w2).sum() * 3).backward()
    xr = x.detach().clone().requires_grad_(True)
    lat_only, = torch.autograd.grad(F.conv3d(xr, w1).sum(), xr)
    xr2 = x.detach().clone().requires_grad_(True)
    full, = torch.autograd.grad(F.conv3d(xr2, w1).sum() + F.conv3d(xr2[:, :, ::2, ::2, ::2], w2).sum() * 3, xr2)
    seen = kept[0] if kept else (lat.grad if mode == "retain_grad" else None)
    return x.grad, seen, lat_only, full, (fe._TAP_REFS[1] - before[1], fe._TAP_REFS[2] - before[2])


def test_stride_tap_adds_in_place_only_into_a_gradient_nobody_else_holds():
    """_StrideTap.backward adds the strided consumers' gradient INTO the lateral's incoming gradient.  In the model's graph that
    tensor is fresh and this node is its only holder: the add runs in place (one strided pass over 1/8 of the rows).  A tensor hook that
    keeps the gradient it is shown shares the tensor: then the add must run on a clone, and what the hook kept stays the lateral's own
    gradient."""
    import torch
    for mode in ("plain", "python_producer", "retain_grad"):
        gx, seen, lat_only, full, (inplace, cloned) = _stride_tap_case(mode)
        assert torch.allclose(gx, full, rtol=1e-5, atol=1e-5), mode
        assert (inplace, cloned) == (1, 0), (mode, inplace, cloned)
        if seen is not None:
            assert torch.allclose(seen, lat_only, rtol=1e-5, atol=1e-5), mode            # retain_grad clones for itself: untouched
    gx, seen, lat_only, full, (inplace, cloned) = _stride_tap_case("hook_keeps_gradient")
    assert (inplace, cloned) == (0, 1)
    assert torch.allclose(gx, full, rtol=1e-5, atol=1e-5)
    assert torch.allclose(seen, lat_only, rtol=1e-5, atol=1e-5)                          # NOT the sum: the kept tensor was not written to


def test_rank_pinning_never_takes_the_job_down(monkeypatch):
    """a cgroup that refuses the mask, or a sysfs without the expected files: the rank runs unpinned and says why"""
    import os
    from medicaldetectiontoolkit_amd.utils import affinity

    def refuse(pid, cpus):
        raise OSError(22, "Invalid argument")
    monkeypatch.setattr(os, "sched_setaffinity", refuse)
    rec = affinity.pin_rank(0, 2, [0, 1], set_torch_threads=False)
    assert rec["pinned"] is False and "OSError" in rec["why"]
    monkeypatch.undo()

    def broken(d):
        raise RuntimeError("no sysfs")
    monkeypatch.setattr(affinity, "gpu_numa_node", broken)
    rec = affinity.pin_rank(1, 2, [0, 1], set_torch_threads=False)
    assert rec["pinned"] is False and "RuntimeError" in rec["why"]
