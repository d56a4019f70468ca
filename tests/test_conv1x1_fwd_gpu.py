"""mdt_conv1x1_forward (csrc/conv1x1_fwd.hip): the bottleneck 1x1 layers of the reference's ResBlock (models/backbone.py:197-206) with bias /
residual / ReLU inside the convolution's pass.  Against a float64 convolution within fp32 summation-order bounds, against the two-pass form
(MIOpen + mdt_bias_act_forward) through autograd, ragged voxel counts, and the ResBlock as a whole with the switch on and off."""
import pytest
import torch
import torch.nn.functional as F

from medicaldetectiontoolkit_amd import _lib
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe

pytestmark = pytest.mark.gpu

PAIRS = [(18, 72), (72, 18), (36, 144)]


@pytest.mark.parametrize("cin,cout", PAIRS)
@pytest.mark.parametrize("with_res,relu", [(True, True), (False, True), (True, False), (False, False)])
@pytest.mark.parametrize("V", [32 * 40, 1000 + 7, 5])
def test_conv1x1_forward_vs_float64(cin, cout, with_res, relu, V, cuda):
    g = torch.Generator(device=cuda).manual_seed(cin * 1000 + cout + V)
    x = torch.randn(V, cin, device=cuda, generator=g)
    w = torch.randn(cout, cin, device=cuda, generator=g) * 0.2
    wpad = torch.zeros(cout * cin + 1, device=cuda)
    wpad[1:].copy_(w.reshape(-1))
    wv = wpad[1:].view(cout, cin)                       # a filter inside a flat parameter buffer: 4-byte aligned only
    b = torch.randn(cout, device=cuda, generator=g)
    res = torch.randn(V, cout, device=cuda, generator=g) if with_res else None
    out = torch.full((V + 1, cout), 7.0, device=cuda)    # one guard row behind the result
    rc = _lib.lib().mdt_conv1x1_forward(x.data_ptr(), wv.data_ptr(), b.data_ptr(), res.data_ptr() if with_res else None, out.data_ptr(), V, cin, cout,
                                        1 if relu else 0, _lib.raw_stream())
    assert rc == 0
    ref = x.double() @ w.double().t() + b.double()
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    bound = 2e-6 * (x.double().abs() @ w.double().abs().t() + b.double().abs() + (res.double().abs() if with_res else 0)) + 1e-7
    assert bool(((out[:V].double() - ref).abs() <= bound).all())
    assert bool((out[V] == 7.0).all())
    out2 = torch.empty(V, cout, device=cuda)
    _lib.lib().mdt_conv1x1_forward(x.data_ptr(), wv.data_ptr(), b.data_ptr(), res.data_ptr() if with_res else None, out2.data_ptr(), V, cin, cout,
                                   1 if relu else 0, _lib.raw_stream())
    assert torch.equal(out2, out[:V])                    # fixed order: run-to-run identical


def test_conv1x1_forward_declines_other_shapes(cuda):
    L = _lib.lib()
    assert not L.mdt_conv1x1_forward_supported(18, 18) and not L.mdt_conv1x1_forward_supported(72, 288) and not L.mdt_conv1x1_forward_supported(144, 36)
    x = torch.zeros(64, 20, device=cuda)
    rc = L.mdt_conv1x1_forward(x.data_ptr(), x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 64, 20, 20, 0, _lib.raw_stream())
    assert rc == _lib.MDT_ERR_UNSUPPORTED


@pytest.mark.parametrize("cin,cout,shape", [(18, 72, (2, 32, 32, 32)), (36, 144, (2, 32, 32, 16)), (72, 18, (2, 32, 32, 16))])
@pytest.mark.parametrize("with_res,relu", [(True, True), (False, False)])
def test_conv_bias_module_with_the_fused_forward_equals_the_two_pass_form(cin, cout, shape, with_res, relu, cuda):
    """ConvBias3d through autograd: outputs within fp32 rounding of MIOpen + epilogue, all four gradients too (the backward is the same code on
    an output that differs by rounding; the ReLU mask may flip only where |y| is at rounding level)"""
    mf = torch.channels_last_3d
    g = torch.Generator(device=cuda).manual_seed(cin + cout)
    conv = fe.ConvBias3d(cin, cout, 1).to(cuda).to(memory_format=mf)
    x0 = torch.randn((shape[0], cin) + shape[1:], device=cuda, generator=g).contiguous(memory_format=mf)
    r0 = torch.randn((shape[0], cout) + shape[1:], device=cuda, generator=g).contiguous(memory_format=mf) if with_res else None
    gy = torch.randn((shape[0], cout) + shape[1:], device=cuda, generator=g).contiguous(memory_format=mf)

    def run(on):
        fe.CONV1X1_FWD = on
        try:
            conv.zero_grad()
            x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            r = r0.clone(memory_format=torch.preserve_format).requires_grad_(True) if with_res else None
            y = conv(x, residual=(r * 1.0 if with_res else None), relu=relu)
            y.backward(gy)
            return y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone(), (r.grad.clone() if with_res else None)
        finally:
            fe.CONV1X1_FWD = True

    assert fe.conv1x1_forward_applies(conv, x0)
    a, b = run(True), run(False)
    scale = float(b[0].abs().max())
    assert float((a[0] - b[0]).abs().max()) <= 1e-5 * scale
    flipped = ((a[0] > 0) != (b[0] > 0)).float().mean().item() if relu else 0.0
    assert flipped <= 1e-4
    for u, v in zip(a[1:], b[1:]):
        if u is None:
            continue
        assert float((u - v).abs().max()) <= 2e-4 * float(v.abs().max()) + (1e-2 * float(v.abs().max()) if flipped > 0 else 0.0)


def test_identity_resblock_with_fused_1x1_layers_equals_plain_modules(cuda):
    """a C2-shaped identity ResBlock (72 -> 18 -> 18 -> 72): res-tap conv1 + bias + ReLU and conv3 + residual + ReLU on mdt_conv1x1_forward vs the switch off"""
    from medicaldetectiontoolkit_amd.models import backbone
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    mf = torch.channels_last_3d
    conv = mutils.NDConvGenerator(3)
    torch.manual_seed(4)
    blk = backbone.ResBlock(72, 18, conv=conv, norm=None, relu="relu").to(cuda).to(memory_format=mf)
    x0 = torch.randn(2, 72, 32, 32, 64, device=cuda).contiguous(memory_format=mf)
    gy = torch.randn(2, 72, 32, 32, 64, device=cuda).contiguous(memory_format=mf)

    def run(on):
        fe.CONV1X1_FWD = on
        try:
            blk.zero_grad()
            x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            y = blk(x * 1.0)
            y.backward(gy)
            return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
        finally:
            fe.CONV1X1_FWD = True

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) <= 1e-3 * float(v.abs().max()) + 1e-6


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("V", [32 * 64, 1000 + 7, 3])
def test_conv1x1_backward_vs_float64(relu, V, cuda):
    """mdt_conv1x1_backward: g exact (a mask), gx and gbias against float64 within fp32 summation-order bounds, run-to-run identical"""
    cin, cout = 18, 72
    g_ = torch.Generator(device=cuda).manual_seed(V + relu)
    gy = torch.randn(V, cout, device=cuda, generator=g_)
    y = torch.randn(V, cout, device=cuda, generator=g_) if relu else None
    w = torch.randn(cout, cin, device=cuda, generator=g_) * 0.2
    L = _lib.lib()
    assert L.mdt_conv1x1_backward_supported(cin, cout) and not L.mdt_conv1x1_backward_supported(36, 144)
    wsb = L.mdt_conv1x1_backward_workspace_bytes(V, cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)

    def run():
        g = torch.full((V + 1, cout), 7.0, device=cuda) if relu else None
        gx = torch.full((V + 1, cin), 7.0, device=cuda)
        gb = torch.empty(cout, device=cuda)
        rc = L.mdt_conv1x1_backward(gy.data_ptr(), y.data_ptr() if relu else None, w.data_ptr(), g.data_ptr() if relu else None, gx.data_ptr(), gb.data_ptr(), V, cin, cout,
                                    ws.data_ptr(), wsb, _lib.raw_stream())
        assert rc == 0
        return g, gx, gb

    g, gx, gb = run()
    gref = gy * (y > 0) if relu else gy
    if relu:
        assert torch.equal(g[:V], gref) and bool((g[V] == 7.0).all())
    assert bool((gx[V] == 7.0).all())
    ref = gref.double() @ w.double()
    bound = 2e-6 * (gref.double().abs() @ w.double().abs()) + 1e-7
    assert bool(((gx[:V].double() - ref).abs() <= bound).all())
    bref = gref.double().sum(0)
    assert float((gb.double() - bref).abs().max()) <= 2e-6 * float(gref.double().abs().sum(0).max())
    g2, gx2, gb2 = run()
    assert torch.equal(gx2, gx) and torch.equal(gb2, gb)
    # too small a workspace / a ReLU without a place for g are argument errors
    assert L.mdt_conv1x1_backward(gy.data_ptr(), None, w.data_ptr(), None, gx.data_ptr(), gb.data_ptr(), V, cin, cout, ws.data_ptr(), 16, _lib.raw_stream()) == -2
    if relu:
        assert L.mdt_conv1x1_backward(gy.data_ptr(), y.data_ptr(), w.data_ptr(), None, gx.data_ptr(), gb.data_ptr(), V, cin, cout, ws.data_ptr(), wsb, _lib.raw_stream()) == -1


def test_conv_bias_module_backward_in_one_pass_equals_the_separate_passes(cuda):
    """ConvBias3d 18 -> 72 + residual + ReLU: the backward through mdt_conv1x1_backward vs mdt_bias_act_backward + the input-gradient convolution"""
    mf = torch.channels_last_3d
    g = torch.Generator(device=cuda).manual_seed(8)
    conv = fe.ConvBias3d(18, 72, 1).to(cuda).to(memory_format=mf)
    x0 = torch.randn(2, 18, 32, 32, 32, device=cuda, generator=g).contiguous(memory_format=mf)
    r0 = torch.randn(2, 72, 32, 32, 32, device=cuda, generator=g).contiguous(memory_format=mf)
    gy = torch.randn(2, 72, 32, 32, 32, device=cuda, generator=g).contiguous(memory_format=mf)

    def run(on, relu):
        fe.CONV1X1_BWD = on
        try:
            conv.zero_grad()
            x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            r = r0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            conv(x, residual=r * 1.0, relu=relu).backward(gy)
            return x.grad.clone(), r.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()
        finally:
            fe.CONV1X1_BWD = True

    for relu in (True, False):
        a, b = run(True, relu), run(False, relu)
        assert torch.equal(a[1], b[1])                          # the masked gradient itself
        for u, v in zip(a, b):
            assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max())


@pytest.mark.parametrize("shapes", [[(16, 16, 8), (8, 8, 4), (4, 4, 2)], [(5, 7, 3)], [(32, 32, 16), (16, 16, 8), (8, 8, 4), (4, 4, 2)]])
def test_rpn_levels_fused_equals_the_modules_and_the_concatenation(shapes, cuda):
    """mdt_rpn_heads_forward through models/mrcnn.rpn_levels_fused: logits / deltas within fp32 summation-order rounding of RPN.forward per level + torch.cat
    (the reference's mrcnn.py:40-86, :1030-1033), probs the softmax of the logits it returns; ragged voxel counts (tiles straddle batch elements)"""
    from medicaldetectiontoolkit_amd.configs import Configs
    from medicaldetectiontoolkit_amd.models import mrcnn
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    cf = Configs(dim=3, model="mrcnn", patch_size=[64, 64, 32], batch_size=2, channels_last=True)
    torch.manual_seed(3)
    rpn = mrcnn.RPN(cf, mutils.NDConvGenerator(3)).to(cuda).to(memory_format=torch.channels_last_3d)
    flat = torch.zeros(129, device=cuda)                    # conv_shared's bias inside a flat buffer: 4-byte aligned only
    flat[1:].copy_(rpn.conv_shared[0].bias.detach())
    rpn.conv_shared[0].bias.data = flat[1:]
    g = torch.Generator(device=cuda).manual_seed(9)
    maps = [torch.randn((3, cf.end_filts) + s, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d) for s in shapes]
    with torch.no_grad():
        fused = mrcnn.rpn_levels_fused(rpn, maps)
        assert fused is not None
        ref = [torch.cat(list(o), dim=1) for o in zip(*[rpn(m) for m in maps])]
    for k in (0, 2):
        assert fused[k].shape == ref[k].shape
        assert float((fused[k] - ref[k]).abs().max()) <= 2e-5 * float(ref[k].abs().max())
    assert torch.equal(fused[1], F.softmax(fused[0], dim=2))
    assert float((fused[1] - ref[1]).abs().max()) <= 1e-5
    with torch.no_grad():
        again = mrcnn.rpn_levels_fused(rpn, maps)
    assert torch.equal(again[0], fused[0]) and torch.equal(again[2], fused[2])
