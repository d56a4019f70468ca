"""Fused conv epilogue (csrc/epilogue.hip) against plain torch ops: forward and input gradient bit-identical, bias
gradient to fp32 summation-order rounding, both memory orders, with / without residual and ReLU; run-to-run
deterministic."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 18, 9, 10, 12), (3, 36, 8, 8, 16), (2, 288, 4, 4, 8), (2, 144, 6, 6), (1, 7, 5, 3, 3)])
@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("with_res,relu", [(False, True), (True, True), (True, False), (False, False)])
def test_bias_act_matches_torch(shape, channels_last, with_res, relu, cuda):
    g = torch.Generator(device=cuda).manual_seed(hash((shape, channels_last, with_res, relu)) % 2 ** 31)
    mf = (torch.channels_last_3d if len(shape) == 5 else torch.channels_last) if channels_last else torch.contiguous_format
    x0 = torch.randn(shape, device=cuda, generator=g).contiguous(memory_format=mf)
    bias0 = torch.randn(shape[1], device=cuda, generator=g)
    res0 = torch.randn(shape, device=cuda, generator=g).contiguous(memory_format=mf) if with_res else None
    gy = torch.randn(shape, device=cuda, generator=g).contiguous(memory_format=mf)

    def run(fused):
        x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        b = bias0.clone().requires_grad_(True)
        r = res0.clone(memory_format=torch.preserve_format).requires_grad_(True) if with_res else None
        pre = x * 1.0          # a fresh non-leaf tensor, like a conv output (the fused op works in place)
        if fused:
            y = fe.bias_act(pre, b, r, relu)
        else:
            y = pre + b.view([1, -1] + [1] * (len(shape) - 2))
            if r is not None:
                y = y + r
            y = F.relu(y) if relu else y
        y.backward(gy)
        return y.detach(), x.grad, b.grad, (r.grad if r is not None else None)

    yf, gxf, gbf, grf = run(True)
    yt, gxt, gbt, grt = run(False)
    assert torch.equal(yf, yt)
    assert torch.equal(gxf, gxt)
    if with_res:
        assert torch.equal(grf, grt)
    assert torch.allclose(gbf, gbt, rtol=1e-4, atol=1e-4 * float(gbt.abs().max() + 1))
    yf2, gxf2, gbf2, _ = run(True)
    assert torch.equal(gbf, gbf2) and torch.equal(gxf, gxf2)


@pytest.mark.parametrize("shape", [(8, 36, 32, 32, 128), (8, 18, 64, 64, 32), (4, 288, 4, 4, 16), (1, 7, 33, 17, 5)])
def test_bias_grad_second_stage_inside_the_launch(shape, cuda):
    """mdt_bias_act_backward_ticket (round 6): the last block of the launch folds the per-block partial rows -- thousands of blocks spread over
    the eight XCDs, whose L2s are not coherent: a missing release / acquire shows up as a stale partial row.  Against an fp64 sum, twenty
    launches in a row bit-identical (the ticket is left at zero each time), equal to the two-launch form within summation-order rounding."""
    g = torch.Generator(device=cuda).manual_seed(5)
    gy = torch.randn(shape, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    y = torch.randn(shape, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    want = (gy.double() * (y > 0)).sum((0, 2, 3, 4))
    scale = (gy.double().abs() * (y > 0)).sum((0, 2, 3, 4)) + 1.0
    prev_flag, fe.BIAS_GRAD_IN_LAUNCH = fe.BIAS_GRAD_IN_LAUNCH, True
    first = None
    for it in range(20):
        gx, gb = fe._bias_act_bwd(gy, y, True, torch.channels_last_3d)
        if it % 3 == 0:                                  # other traffic between launches: the partial rows must not survive in a cache
            torch.randn(1 << 22, device=cuda).sum()
        assert float(((gb.double() - want).abs() / scale).max()) <= 2e-6, it
        if first is None:
            first = gb.clone()
            assert torch.equal(gx, gy * (y > 0))
        assert torch.equal(gb, first), it
    assert int(fe._TICKET[gy.device].item()) == 0
    fe.BIAS_GRAD_IN_LAUNCH = False
    try:
        _, gb2 = fe._bias_act_bwd(gy, y, True, torch.channels_last_3d)
    finally:
        fe.BIAS_GRAD_IN_LAUNCH = prev_flag
    assert float(((gb2.double() - first.double()).abs() / scale).max()) <= 2e-6


def test_fused_modules_keep_reference_state_dict_layout(cuda):
    from medicaldetectiontoolkit_amd.utils.model_utils import NDConvGenerator
    conv = NDConvGenerator(3)
    a = conv(4, 8, ks=3, pad=1, relu="relu")
    b = conv(4, 8, ks=1, relu=None)
    assert sorted(a.state_dict()) == ["0.bias", "0.weight"] and sorted(b.state_dict()) == ["bias", "weight"]
    a, b = a.to(cuda), b.to(cuda)
    x = torch.randn(2, 4, 6, 6, 6, device=cuda)
    want_a = F.relu(F.conv3d(x, a[0].weight, a[0].bias, padding=1))
    assert torch.allclose(a(x), want_a, atol=1e-5)
    r = torch.randn(2, 8, 6, 6, 6, device=cuda)
    assert torch.allclose(b(x, residual=r, relu=True), F.relu(F.conv3d(x, b.weight, b.bias) + r), atol=1e-5)


@pytest.mark.parametrize("dim,cin,cout,ks,shape", [(3, 18, 18, 3, (2, 12, 10, 16)), (3, 36, 128, 3, (1, 8, 8, 16)), (3, 18, 72, 1, (2, 8, 8, 8)),
                                                   (2, 24, 48, 3, (2, 20, 24)), (3, 5, 7, 3, (1, 5, 6, 7))])
@pytest.mark.parametrize("channels_last", [False, True])
def test_conv_input_gradient_as_forward_conv(dim, cin, cout, ks, shape, channels_last, cuda):
    """utils/fused_epilogue._ConvStride1: output identical (same MIOpen forward call); input gradient = forward convolution
    of the output gradient with the flipped/transposed filter, equal to MIOpen's backward-data to fp32 summation order;
    weight gradient is the same MIOpen call"""
    g = torch.Generator(device=cuda).manual_seed(cin * 131 + cout)
    mf = (torch.channels_last_3d if dim == 3 else torch.channels_last) if channels_last else torch.contiguous_format
    conv_fn = F.conv3d if dim == 3 else F.conv2d
    x0 = torch.randn((shape[0], cin) + shape[1:], device=cuda, generator=g).contiguous(memory_format=mf)
    w0 = (torch.randn((cout, cin) + (ks,) * dim, device=cuda, generator=g) * 0.1).contiguous(memory_format=mf)
    gy = torch.randn((shape[0], cout) + shape[1:], device=cuda, generator=g).contiguous(memory_format=mf)
    pad = (ks // 2,) * dim
    x1, w1 = x0.clone(memory_format=torch.preserve_format).requires_grad_(True), w0.clone(memory_format=torch.preserve_format).requires_grad_(True)
    x2, w2 = x0.clone(memory_format=torch.preserve_format).requires_grad_(True), w0.clone(memory_format=torch.preserve_format).requires_grad_(True)
    y1 = fe._ConvStride1.apply(x1, w1, pad)
    y2 = conv_fn(x2, w2, None, 1, pad)
    assert torch.equal(y1, y2)
    y1.backward(gy)
    y2.backward(gy)
    scale = conv_fn(gy.abs(), w0.abs().flip(*range(2, 2 + dim)).transpose(0, 1).contiguous(), None, 1, pad)       # sum |terms| per input element
    assert torch.all((x1.grad - x2.grad).abs() <= 2e-6 * scale + 1e-30)
    assert torch.allclose(w1.grad, w2.grad, rtol=1e-4, atol=1e-4 * w2.grad.abs().max().item())


@pytest.mark.parametrize("cin,k", [(1, 7), (2, 3), (1, 5)])
def test_stem_space_to_depth_equals_strided_conv(cin, k, cuda):
    """utils/fused_epilogue._ConvStem221 (stride (2, 2, 1), pad k // 2): forward equal to the strided convolution to fp32
    summation order; weight gradient is the strided problem's own MIOpen call"""
    g = torch.Generator(device=cuda).manual_seed(7 * cin + k)
    x0 = torch.randn((2, cin, 32, 24, 20), device=cuda, generator=g)
    w0 = torch.randn((18, cin, k, k, k), device=cuda, generator=g) * 0.05
    x1, w1 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    x2, w2 = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
    y1 = fe._ConvStem221.apply(x1, w1)
    y2 = F.conv3d(x2, w2, None, (2, 2, 1), k // 2)
    assert y1.shape == y2.shape
    scale = F.conv3d(x0.abs(), w0.abs(), None, (2, 2, 1), k // 2)
    assert torch.all((y1 - y2).abs() <= 2e-6 * scale + 1e-30)
    gy = torch.randn(y2.shape, device=cuda, generator=g)
    y1.backward(gy)
    y2.backward(gy)
    assert torch.allclose(w1.grad, w2.grad, rtol=1e-4, atol=1e-4 * w2.grad.abs().max().item())
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-4 * x2.grad.abs().max().item())
    conv = torch.nn.Conv3d(cin, 18, k, stride=(2, 2, 1), padding=k // 2).to(cuda)
    assert fe._is_stem221(conv, x0) and not fe._is_stem221(conv, x0[:, :, :31])


@pytest.mark.parametrize("shape", [(2, 18, 16, 12, 10), (1, 5, 7, 9, 4), (2, 36, 8, 8, 1), (1, 3, 2, 2, 3)])
@pytest.mark.parametrize("ties", [False, True])
def test_maxpool_channels_last_equals_torch(shape, ties, cuda):
    """csrc/pool.hip vs torch.nn.functional.max_pool3d(3, (2, 2, 1), 1): forward values bit-identical (incl. ties after a
    ReLU, -inf and NaN), input gradient equal (same arg-max rule; sums of <= 12 terms in a fixed order)"""
    g = torch.Generator(device=cuda).manual_seed(sum(shape) + int(ties))
    x0 = torch.randn(shape, device=cuda, generator=g)
    if ties:
        x0 = torch.relu(x0)                       # many exact ties at 0, like the activation in front of the stem pooling
        x0[0, 0, 0, 0, 0] = float("nan")
        x0[0, 1] = float("-inf")
    x1 = x0.contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    x2 = x0.clone().requires_grad_(True)
    pool = fe.MaxPool3dStem(kernel_size=3, stride=(2, 2, 1), padding=1)
    y1 = pool(x1)
    y2 = F.max_pool3d(x2, 3, (2, 2, 1), 1)
    assert y1.shape == y2.shape and y1.is_contiguous(memory_format=torch.channels_last_3d)
    assert torch.equal(torch.nan_to_num(y1, nan=123.0), torch.nan_to_num(y2, nan=123.0))
    assert torch.equal(torch.isnan(y1), torch.isnan(y2))
    gy = torch.randn(y2.shape, device=cuda, generator=g)
    y1.backward(gy.contiguous(memory_format=torch.channels_last_3d))
    y2.backward(gy)
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-6, atol=1e-6)
    # run-to-run identical (torch's backward uses atomics, this one does not)
    x3 = x0.contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    pool(x3).backward(gy.contiguous(memory_format=torch.channels_last_3d))
    assert torch.equal(x3.grad, x1.grad)


def test_maxpool_module_falls_back_to_torch_outside_its_case(cuda):
    x = torch.randn(1, 4, 8, 8, 8, device=cuda)                      # contiguous NCDHW: torch path
    assert torch.equal(fe.MaxPool3dStem(3, (2, 2, 1), 1)(x), F.max_pool3d(x, 3, (2, 2, 1), 1))
    xc = x.contiguous(memory_format=torch.channels_last_3d)
    assert torch.equal(fe.MaxPool3dStem(2, 2, 0)(xc), F.max_pool3d(xc, 2, 2, 0))


@pytest.mark.parametrize("shape", [(18, 18, 3, 3, 3), (128, 36, 3, 3, 3), (72, 18, 1, 1, 1), (48, 24, 3, 3), (7, 5, 7, 7, 3)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_filter_flip_transpose_equals_torch(shape, channels_last, cuda):
    g = torch.Generator(device=cuda).manual_seed(len(shape) + shape[0])
    nd = len(shape) - 2
    mf = (torch.channels_last_3d if nd == 3 else torch.channels_last) if channels_last else torch.contiguous_format
    w = torch.randn(shape, device=cuda, generator=g).contiguous(memory_format=mf)
    got = fe.flip_transpose_filter(w, mf)
    want = w.transpose(0, 1).flip(*range(2, 2 + nd))
    assert got.is_contiguous(memory_format=mf) and torch.equal(got, want)


# ------------------------------------------------------------------ 1x1(x1) weight gradient (csrc/conv1x1_wgrad.hip)
WGRAD_CASES = [
    # B, Cin, Cout, spatial
    (2, 18, 72, (16, 16, 32)),       # C2 bottleneck expand (3 x 1 tiles)
    (2, 72, 18, (16, 16, 32)),       # C2 bottleneck reduce (1 x 3 tiles)
    (2, 18, 18, (8, 8, 16)),
    (1, 128, 18, (8, 8, 16)),        # 1 x 4 tiles
    (2, 36, 144, (8, 8, 8)),         # 5 x 2 tiles -> tile groups
    (1, 144, 576, (4, 4, 4)),        # 18 x 5 tiles -> (2, 2) groups
    (1, 5, 3, (3, 5, 7)),            # 105 voxels: tail loop only, odd count
    (3, 7, 33, (5, 5, 5)),           # 375 voxels, Cout just over one tile
    # the production shapes of the benchmarked configuration: the C2 maps of 8 x 128^3 patches
    (8, 18, 72, (32, 32, 128)),
    (8, 72, 18, (32, 32, 128)),
]


def test_batched_filter_flips_follow_the_weights(cuda):
    """_FlipCache (round 6): the flipped / transposed filters of all registered parameters from ONE launch per step.  Registered at first
    sight; refreshed when a raw kernel rewrote the weights (weights_changed(), what training.FlatAdam calls) or a torch op did (version
    counter); filters of deleted parameters are dropped from the table; results equal the torch expression."""
    from medicaldetectiontoolkit_amd import _lib

    def want(w, mf):
        wt = w.detach().transpose(0, 1)
        if w.shape[2:].numel() > 1:
            wt = wt.flip(*range(2, w.dim()))
        return wt.contiguous(memory_format=mf)
    fe._FLIP.clear()
    g = torch.Generator(device=cuda).manual_seed(3)
    shapes = [(18, 18, 3, 3, 3), (36, 36, 3, 3, 3), (72, 18, 1, 1, 1), (36, 128, 3, 3, 3), (7, 5, 7, 7, 3)]
    mfs = [torch.channels_last_3d, torch.channels_last_3d, torch.contiguous_format, torch.channels_last_3d, torch.contiguous_format]
    ws = [torch.nn.Parameter(torch.randn(sh, device=cuda, generator=g).contiguous(memory_format=mf)) for sh, mf in zip(shapes, mfs)]
    for w, mf in zip(ws, mfs):                       # first sight: registered, flipped one by one
        assert torch.equal(fe.flip_transpose_filter(w, mf), want(w, mf))
    _lib.count_calls(True)
    try:
        with torch.no_grad():
            ws[1].mul_(2.0)                          # a torch op: seen through the version counter
        outs = [fe.flip_transpose_filter(w, mf) for w, mf in zip(ws, mfs)]
        for w, mf, o in zip(ws, mfs, outs):
            assert torch.equal(o, want(w, mf))
        c1 = dict(_lib.CALLS)
        assert c1.get("mdt_filter_flip_transpose_batched", 0) == 1 and c1.get("mdt_filter_flip_transpose", 0) == 0, c1
        ws[3].data.add_(1.0)                         # a raw update: .data does not bump the version of the parameter
        fe.weights_changed()
        outs = [fe.flip_transpose_filter(w, mf) for w, mf in zip(ws, mfs)]
        for w, mf, o in zip(ws, mfs, outs):
            assert torch.equal(o, want(w, mf))
        c2 = dict(_lib.CALLS)
        assert c2.get("mdt_filter_flip_transpose_batched", 0) == 2 and c2.get("mdt_filter_flip_transpose", 0) == 0, c2
        # unchanged weights: served from the buffers, no launch at all
        outs = [fe.flip_transpose_filter(w, mf) for w, mf in zip(ws, mfs)]
        assert dict(_lib.CALLS) == c2
        # a parameter dies: its record leaves the table at the next refresh
        wdev = ws[0].device
        del ws[0], outs
        import gc
        gc.collect()
        fe.weights_changed()
        for w, mf in zip(ws, mfs[1:]):
            assert torch.equal(fe.flip_transpose_filter(w, mf), want(w, mf))
        assert len(fe._FLIP[wdev].entries) == 4
    finally:
        _lib.count_calls(False)
    # a non-parameter tensor keeps the single launch
    t = torch.randn(6, 4, 3, 3, 3, device=cuda, generator=g)
    assert torch.equal(fe.flip_transpose_filter(t, torch.contiguous_format), want(t, torch.contiguous_format))


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[str(c) for c in WGRAD_CASES])
def test_conv1x1_wgrad_vs_aten(case, cuda):
    """mdt_conv1x1_wgrad == aten.convolution_backward's weight gradient (MIOpen) to fp32 summation order: 1e-5 relative to the
    summed magnitudes; asymmetric random operands (a row <-> column swap of the MFMA C/D map cannot pass); run-to-run identical"""
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    B, cin, cout, sp = case
    g = torch.Generator(device=cuda).manual_seed(cin * 1000 + cout)
    x = torch.randn((B, cin) + sp, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    gy = torch.randn((B, cout) + sp, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((cout, cin, 1, 1, 1), device=cuda, generator=g)
    got = fe.conv1x1_weight_grad(gy, x, w, force=True)
    assert got is not None and got.shape == w.shape
    X = x.permute(0, 2, 3, 4, 1).reshape(-1, cin).double()
    G = gy.permute(0, 2, 3, 4, 1).reshape(-1, cout).double()
    want = (G.t() @ X)
    mag = (G.abs().t() @ X.abs())
    assert torch.all((got.view(cout, cin).double() - want).abs() <= 1e-5 * mag + 1e-12), float(((got.view(cout, cin).double() - want).abs() / mag).max())
    ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [0, 0, 0], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False])[1]
    assert torch.all((got.double() - ref.double()).abs().view(cout, cin) <= 2e-5 * mag + 1e-12)
    assert torch.equal(got, fe.conv1x1_weight_grad(gy, x, w, force=True))


def test_conv1x1_wgrad_inside_autograd_matches_miopen(cuda):
    """a 1x1x1 ConvBias3d layer trained one step with the kernel on / off: same weight gradient (2D layer too)"""
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    for nd in (3, 2):
        conv = (fe.ConvBias3d if nd == 3 else fe.ConvBias2d)(18, 72, 1).to(cuda)
        mf = torch.channels_last_3d if nd == 3 else torch.channels_last
        conv = conv.to(memory_format=mf)
        x = torch.randn((2, 18) + ((32, 32, 40) if nd == 3 else (160, 256)), device=cuda).contiguous(memory_format=mf).requires_grad_(True)
        grads = []
        for flag in (True, False):
            fe.WGRAD_1X1 = flag
            conv.zero_grad()
            x.grad = None
            conv(x).square().sum().backward()
            grads.append((conv.weight.grad.clone(), x.grad.clone()))
        fe.WGRAD_1X1 = True
        scale = float(grads[1][0].abs().max())
        assert float((grads[0][0] - grads[1][0]).abs().max()) <= 1e-4 * scale
        assert torch.allclose(grads[0][1], grads[1][1])


# ------------------------------------------------------------------ x2 (y, x) linear up-sampling on channels-last storage
@pytest.mark.parametrize("case", [(2, 36, (8, 12, 16), "trilinear"), (1, 5, (3, 4, 7), "trilinear"), (2, 18, (1, 6, 8), "trilinear"),
                                  (2, 48, (10, 14), "bilinear"), (1, 3, (5, 1), "bilinear")])
def test_upsample2x_channels_last_matches_interpolate(case, cuda):
    """csrc/upsample.hip == F.interpolate(scale (2, 2, 1) 'trilinear' / scale 2 'bilinear', align_corners=False), forward 1e-6 and
    the backward (gather-form adjoint) against torch's autograd; deterministic"""
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    B, C, sp, mode = case
    nd = len(sp)
    mf = torch.channels_last_3d if nd == 3 else torch.channels_last
    sf = (2, 2, 1) if nd == 3 else 2
    g = torch.Generator(device=cuda).manual_seed(C)
    x = torch.randn((B, C) + sp, device=cuda, generator=g).contiguous(memory_format=mf).requires_grad_(True)
    y = fe.upsample2x_yx(x, sf, mode)
    assert y is not None and y.is_contiguous(memory_format=mf)
    xr = x.detach().clone().requires_grad_(True)
    want = F.interpolate(xr, scale_factor=sf, mode=mode, align_corners=False)
    assert y.shape == want.shape
    assert torch.allclose(y, want, rtol=1e-6, atol=1e-6), float((y - want).abs().max())
    go = torch.randn(want.shape, device=cuda, generator=g)
    y.backward(go.contiguous(memory_format=mf))
    want.backward(go)
    assert torch.allclose(x.grad, xr.grad, rtol=1e-5, atol=1e-5), float((x.grad - xr.grad).abs().max())
    x2 = x.detach().clone().requires_grad_(True)
    y2 = fe.upsample2x_yx(x2, sf, mode)
    y2.backward(go.contiguous(memory_format=mf))
    assert torch.equal(y2, y) and torch.equal(x2.grad, x.grad)


# ------------------------------------------------------------------ few-channel 3x3x3 convolution (csrc/conv3x3x3_small.hip)
CONV3_CASES = [
    # B, Cin, Cout, (Y, X, Z)
    (2, 18, 18, (16, 16, 128)),      # the C2 bottleneck layer (3 K-steps per trip)
    (1, 18, 18, (5, 7, 32)),         # x not a multiple of 4, single z tile, borders everywhere
    (2, 6, 30, (6, 9, 64)),          # C_out > C_in
    (2, 16, 5, (4, 6, 64)),          # generic K loop (8 K-steps), few outputs
    (1, 12, 20, (3, 5, 32)),         # LDS budget: 27 * 12 * 20 + 18 rows * 34 * 12 floats
    (8, 18, 18, (32, 32, 128)),      # the production shape: ResBlock.conv2 of stage C2 at 8 x 128^3
]


@pytest.mark.parametrize("case", CONV3_CASES, ids=[str(c) for c in CONV3_CASES])
def test_conv3x3x3_small_vs_torch(case, cuda):
    """mdt_conv3x3x3_small_forward == F.conv3d(padding=1) to fp32 summation order (1e-5 of the summed magnitudes), zero padding on
    all six faces; asymmetric random operands"""
    from medicaldetectiontoolkit_amd import _lib
    B, cin, cout, sp = case
    g = torch.Generator(device=cuda).manual_seed(cin * 100 + cout)
    x = torch.randn((B, cin) + sp, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((cout, cin, 3, 3, 3), device=cuda, generator=g)
    assert _lib.lib().mdt_conv3x3x3_small_supported(sp[0], sp[1], sp[2], cin, cout) == 1
    wt = w.permute(2, 3, 4, 1, 0).contiguous()
    y = torch.empty((B, cout) + sp, device=cuda).contiguous(memory_format=torch.channels_last_3d)
    y.fill_(float("nan"))
    rc = _lib.lib().mdt_conv3x3x3_small_forward(x.data_ptr(), wt.data_ptr(), y.data_ptr(), B, sp[0], sp[1], sp[2], cin, cout,
                                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    want = F.conv3d(x.double(), w.double(), None, 1, 1)
    mag = F.conv3d(x.double().abs(), w.double().abs(), None, 1, 1)
    err = (y.double() - want).abs()
    assert torch.all(err <= 1e-5 * mag + 1e-12), float((err / mag).max())      # a NaN left anywhere (unwritten output) fails


@pytest.mark.parametrize("epilogue", ["bias", "bias_relu", "relu"])
def test_conv3x3x3_small_epilogue_vs_torch(epilogue, cuda):
    """mdt_conv3x3x3_small_forward_bias_act == act(F.conv3d + bias) in fp64 to 1e-5 of the summed magnitudes; the sums are the plain
    kernel's (bit-equal after undoing the epilogue is not asserted: the bias add rounds once more)"""
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    g = torch.Generator(device=cuda).manual_seed(77)
    x = torch.randn((2, 18, 8, 9, 64), device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((18, 18, 3, 3, 3), device=cuda, generator=g) * 0.1
    bias = torch.randn(18, device=cuda, generator=g) if epilogue != "relu" else None
    relu = epilogue != "bias"
    prev = fe.CONV3_SMALL
    fe.CONV3_SMALL = True
    try:
        y = fe.conv3x3x3_small(x, w, bias, relu) if True else None
        # (the Python helper's use-rule asks for >= 65 536 voxels: call below it through the C ABI directly)
        if y is None:
            from medicaldetectiontoolkit_amd import _lib
            wt = w.permute(2, 3, 4, 1, 0).contiguous()
            y = torch.full((2, 18, 8, 9, 64), float("nan"), device=cuda).contiguous(memory_format=torch.channels_last_3d)
            rc = _lib.lib().mdt_conv3x3x3_small_forward_bias_act(x.data_ptr(), wt.data_ptr(), bias.data_ptr() if bias is not None else None, 1 if relu else 0,
                                                                 y.data_ptr(), 2, 8, 9, 64, 18, 18, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
    finally:
        fe.CONV3_SMALL = prev
    want = F.conv3d(x.double(), w.double(), bias.double() if bias is not None else None, 1, 1)
    mag = F.conv3d(x.double().abs(), w.double().abs(), bias.double().abs() if bias is not None else None, 1, 1)
    if relu:
        want = torch.relu(want)
    err = (y.double() - want).abs()
    assert torch.all(err <= 1e-5 * mag + 1e-12), float((err / mag).max())


def test_conv3x3x3_small_inside_autograd_matches_miopen(cuda):
    """an 18 -> 18 ConvBiasReLU layer: forward, input gradient (the kernel with the flipped / transposed filter), weight and bias
    gradient with the kernel on (bias + ReLU in its epilogue, round 4, or as a separate pass) / off"""
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    from medicaldetectiontoolkit_amd.utils.model_utils import NDConvGenerator
    torch.manual_seed(5)
    layer = NDConvGenerator(3)(18, 18, ks=3, pad=1, relu="relu").to(cuda).to(memory_format=torch.channels_last_3d)
    x = torch.randn((2, 18, 16, 16, 128), device=cuda).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    res = []
    for flag, epi in ((True, True), (True, False), (False, False)):     # fused bias + ReLU epilogue / kernel + separate epilogue pass / MIOpen
        fe.CONV3_SMALL, fe.CONV3_SMALL_EPILOGUE = flag, epi
        layer.zero_grad()
        x.grad = None
        y = layer(x)
        assert (y.grad_fn.name() == "_Conv3SmallBiasReLUBackward") == (flag and epi), y.grad_fn.name()
        y.square().sum().backward()
        res.append((y.detach().clone(), x.grad.clone(), layer[0].weight.grad.clone(), layer[0].bias.grad.clone()))
    fe.CONV3_SMALL = fe.CONV3_SMALL_EPILOGUE = True
    assert torch.equal(res[0][0], res[1][0])            # same kernel, same sums: the epilogue only moves the bias add + ReLU
    for other in (res[0], res[1]):
        for a, b in zip(other, res[2]):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max())), float((a - b).abs().max())


# ------------------------------------------------------------------ stem weight gradient (csrc/conv_stem_wgrad.hip)
@pytest.mark.parametrize("case", [(2, 18, 7, (32, 32, 32)), (1, 18, 7, (16, 24, 16)), (2, 5, 3, (12, 8, 8)), (1, 32, 5, (8, 8, 24)),
                                  (1, 18, 7, (8, 16, 128)), (2, 7, 7, (16, 8, 64)), (3, 32, 7, (4, 24, 128)), (1, 18, 7, (12, 12, 64)),
                                  (8, 18, 7, (128, 128, 128))],     # last: the production shape (the stem of 8 x 128^3 patches)
                         ids=lambda c: str(c))
def test_stem_wgrad_vs_aten(case, cuda):
    """mdt_conv_stem_wgrad == aten.convolution_backward's weight gradient of a one-channel k^3 convolution with stride (2, 2, 1),
    pad k // 2 (1e-5 of the summed magnitudes; deterministic).  k = 7 with OZ in {32, 64, 128} and OX % 4 == 0 takes the LDS-image
    kernel, everything else (OZ = 16, k = 3 / 5, OX = 6) the gather kernel"""
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    B, cout, k, sp = case
    g = torch.Generator(device=cuda).manual_seed(cout * 10 + k)
    x = torch.randn((B, 1) + sp, device=cuda, generator=g)
    w = torch.randn((cout, 1, k, k, k), device=cuda, generator=g)
    osp = ((sp[0] + 2 * (k // 2) - k) // 2 + 1, (sp[1] + 2 * (k // 2) - k) // 2 + 1, sp[2])
    gy = torch.randn((B, cout) + osp, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    got = fe.stem_weight_grad(gy, x, w, (2, 2, 1))
    assert got is not None and got.shape == w.shape
    p = k // 2
    want = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [2, 2, 1], [p, p, p], [1, 1, 1], False, [0, 0, 0], 1,
                                               [False, True, False])[1]
    mag = torch.ops.aten.convolution_backward(gy.double().abs(), x.double().abs(), w.double(), None, [2, 2, 1], [p, p, p], [1, 1, 1], False,
                                              [0, 0, 0], 1, [False, True, False])[1]
    err = (got.double() - want).abs()
    assert torch.all(err <= 1e-5 * mag + 1e-12), float((err / mag).max())
    assert torch.equal(got, fe.stem_weight_grad(gy, x, w, (2, 2, 1)))


# ------------------------------------------------------------------ stem forward (csrc/conv_stem_fwd.hip)
STEM_FWD_CASES = [(2, 18, (32, 32, 128)), (1, 18, (16, 24, 64)), (2, 7, (8, 8, 32)), (1, 32, (12, 16, 128)), (3, 18, (6, 40, 64)),
                  (8, 18, (128, 128, 128))]        # last: the production shape


@pytest.mark.parametrize("case", STEM_FWD_CASES, ids=[str(c) for c in STEM_FWD_CASES])
@pytest.mark.parametrize("epilogue", ["plain", "bias", "bias_relu"])
def test_stem_forward_vs_conv3d(case, epilogue, cuda):
    """mdt_conv_stem_forward == F.conv3d(x, w, bias, stride (2, 2, 1), padding 3) (+ ReLU) of the one-channel 7^3 stem, to
    1e-5 of the summed magnitudes (fp32 MFMA, 343 terms in a fixed order: deterministic); borders on all six faces are in every case"""
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    B, cout, sp = case
    if B == 8 and epilogue != "bias_relu":
        pytest.skip("the production shape runs once, in the form the model uses (bias + ReLU epilogue): two fp64 reference convolutions of 8 x 128^3 per run")
    g = torch.Generator(device=cuda).manual_seed(cout + sp[0])
    x = torch.randn((B, 1) + sp, device=cuda, generator=g)
    w = torch.randn((cout, 1, 7, 7, 7), device=cuda, generator=g) * 0.1
    bias = torch.randn(cout, device=cuda, generator=g) if epilogue != "plain" else None
    r = fe.stem_forward(x, w, bias, epilogue == "bias_relu")
    assert r is not None, "shape should be supported"
    got, xp = r
    assert got.shape == (B, cout, sp[0] // 2, sp[1] // 2, sp[2]) and got.is_contiguous(memory_format=torch.channels_last_3d)
    assert xp.shape == (B, sp[0] + 6, sp[1] + 6, sp[2] + 6)
    want = F.conv3d(x.double(), w.double(), bias.double() if bias is not None else None, (2, 2, 1), 3)
    if epilogue == "bias_relu":
        want = torch.relu(want)
    mag = F.conv3d(x.double().abs(), w.double().abs(), bias.double().abs() if bias is not None else None, (2, 2, 1), 3)
    err = (got.double() - want).abs()
    assert torch.all(err <= 1e-5 * mag + 1e-12), float((err / mag).max())
    assert torch.equal(got, fe.stem_forward(x, w, bias, epilogue == "bias_relu")[0])


def test_stem_forward_inside_autograd(cuda):
    """_ConvStem221 on a supported shape runs the forward kernel and hands its padded copy to the weight-gradient kernel: output and
    weight gradient equal to the strided convolution's; with the switch off the space-to-depth path gives the same"""
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    g = torch.Generator(device=cuda).manual_seed(3)
    x0 = torch.randn((2, 1, 16, 16, 64), device=cuda, generator=g)
    w0 = torch.randn((18, 1, 7, 7, 7), device=cuda, generator=g) * 0.05
    gy = torch.randn((2, 18, 8, 8, 64), device=cuda, generator=g)
    res = []
    for on in (True, False):
        fe.STEM_FWD = on
        try:
            w = w0.clone().requires_grad_(True)
            y = fe._ConvStem221.apply(x0, w)
            y.backward(gy)
            res.append((y.detach().clone(), w.grad.clone()))
        finally:
            fe.STEM_FWD = True
    w2 = w0.clone().requires_grad_(True)
    y2 = F.conv3d(x0, w2, None, (2, 2, 1), 3)
    y2.backward(gy)
    for y, gw in res:
        assert torch.allclose(y, y2, rtol=1e-4, atol=1e-4 * float(y2.abs().max()))
        assert torch.allclose(gw, w2.grad, rtol=1e-4, atol=1e-4 * float(w2.grad.abs().max()))


def test_stem_conv_bias_relu_module_fused(cuda):
    """ConvBiasReLU over the stem (models/backbone.py C1) takes the convolution kernel's bias + ReLU epilogue: output, weight and
    bias gradients equal to relu(conv3d(x) + b) on torch ops; the switch off gives the unfused path with the same results"""
    import torch.nn as nn
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    g = torch.Generator(device=cuda).manual_seed(11)
    mod = fe.ConvBiasReLU(fe.ConvBias3d(1, 18, kernel_size=7, padding=3, stride=(2, 2, 1)), nn.ReLU(inplace=True)).to(cuda)
    x = torch.randn((2, 1, 16, 24, 64), device=cuda, generator=g)
    gy = torch.randn((2, 18, 8, 12, 64), device=cuda, generator=g)
    w0, b0 = mod[0].weight.detach().clone(), mod[0].bias.detach().clone()
    wr, br = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    yr = torch.relu(F.conv3d(x, wr, br, (2, 2, 1), 3))
    yr.backward(gy)
    for on in (True, False):
        fe.STEM_FWD = on
        try:
            mod.zero_grad()
            y = mod(x)
            assert (type(y.grad_fn).__name__ == "_ConvStemBiasReLUBackward") == on
            y.backward(gy)
        finally:
            fe.STEM_FWD = True
        assert torch.allclose(y, yr, rtol=1e-4, atol=1e-4 * float(yr.abs().max()))
        assert torch.allclose(mod[0].weight.grad, wr.grad, rtol=1e-4, atol=1e-4 * float(wr.grad.abs().max()))
        assert torch.allclose(mod[0].bias.grad, br.grad, rtol=1e-4, atol=1e-4 * float(br.grad.abs().max()))
    with torch.no_grad():
        assert torch.allclose(mod(x), yr, rtol=1e-4, atol=1e-4 * float(yr.abs().max()))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not fe.stem_forward_supported(x, mod[0].weight)


def test_stem_forward_unsupported_shapes_fall_back(cuda):
    from medicaldetectiontoolkit_amd import _lib
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    L = _lib.lib()
    assert L.mdt_conv_stem_forward_supported(64, 64, 128, 18, 7, 2, 2) == 1
    assert L.mdt_conv_stem_forward_supported(64, 64, 96, 18, 7, 2, 2) == 0      # OZ not in {32, 64, 128}
    assert L.mdt_conv_stem_forward_supported(64, 62, 128, 18, 7, 2, 2) == 0     # OX % 4
    assert L.mdt_conv_stem_forward_supported(64, 64, 128, 40, 7, 2, 2) == 0
    assert L.mdt_conv_stem_forward_supported(64, 64, 128, 18, 5, 2, 2) == 0
    x = torch.randn((1, 1, 16, 16, 20), device=cuda)
    assert fe.stem_forward(x, torch.randn((18, 1, 7, 7, 7), device=cuda)) is None


@pytest.mark.parametrize("case", CONV3_CASES, ids=[str(c) for c in CONV3_CASES])
def test_conv3x3x3_small_wgrad_vs_aten(case, cuda):
    """mdt_conv3x3x3_small_wgrad == aten.convolution_backward's weight gradient (1e-5 of the summed magnitudes), deterministic"""
    from medicaldetectiontoolkit_amd import _lib
    B, cin, cout, sp = case
    g = torch.Generator(device=cuda).manual_seed(cin * 100 + cout + 1)
    x = torch.randn((B, cin) + sp, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    gy = torch.randn((B, cout) + sp, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((cout, cin, 3, 3, 3), device=cuda, generator=g)
    L = _lib.lib()
    wsb = L.mdt_conv3x3x3_small_wgrad_workspace_bytes(B, sp[0], sp[1], cin, cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)
    outs = []
    for _ in range(2):
        gw = torch.full((3, 3, 3, cin, cout), float("nan"), device=cuda)
        rc = L.mdt_conv3x3x3_small_wgrad(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), B, sp[0], sp[1], sp[2], cin, cout, ws.data_ptr(), wsb,
                                         torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(gw.permute(4, 3, 0, 1, 2).contiguous())
    assert torch.equal(outs[0], outs[1])
    want = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1,
                                               [False, True, False])[1]
    mag = torch.ops.aten.convolution_backward(gy.double().abs(), x.double().abs(), w.double(), None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False,
                                              [0, 0, 0], 1, [False, True, False])[1]
    err = (outs[0].double() - want).abs()
    assert torch.all(err <= 1e-5 * mag + 1e-12), float((err / mag).max())


# ------------------------------------------------------------------ residual tap: 1x1 input gradient added to the residual gradient (round 4)
@pytest.mark.parametrize("case", [(18, 72, (2, 8, 8, 64)), (36, 144, (1, 5, 7, 9)), (2, 4, (1, 3, 1, 5)), (72, 168, (1, 4, 4, 4))], ids=lambda c: str(c))
def test_conv1x1_dgrad_add_vs_fp64(case, cuda):
    """mdt_conv1x1_dgrad_add: out = res + gy @ W over channels-last rows == the fp64 product to 1e-6 of the summed magnitudes; deterministic;
    every element of the output written"""
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    cout, cin, sp = case
    g = torch.Generator(device=cuda).manual_seed(cout * 7 + cin)
    gy = torch.randn((sp[0], cout) + sp[1:], device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    res = torch.randn((sp[0], cin) + sp[1:], device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((cout, cin, 1, 1, 1), device=cuda, generator=g)
    got = fe.conv1x1_dgrad_add(gy, w, res)
    assert got is not None and got.shape == res.shape and got.is_contiguous(memory_format=torch.channels_last_3d)
    G = gy.permute(0, 2, 3, 4, 1).reshape(-1, cout).double()
    R = res.permute(0, 2, 3, 4, 1).reshape(-1, cin).double()
    W = w.reshape(cout, cin).double()
    want = R + G @ W
    mag = R.abs() + G.abs() @ W.abs()
    err = (got.permute(0, 2, 3, 4, 1).reshape(-1, cin).double() - want).abs()
    assert torch.all(err <= 1e-6 * mag + 1e-12), float((err / mag).max())
    assert torch.equal(got, fe.conv1x1_dgrad_add(gy, w, res))


def test_resblock_residual_tap_equals_plain_autograd(cuda):
    """an identity ResBlock (72 -> 18 -> 18 -> 72 on a 32 x 32 x 64 map) with the residual tap on / off: same output, same input gradient,
    same parameter gradients (fp32 summation order only); with the tap on the block's autograd graph contains the tap node"""
    from medicaldetectiontoolkit_amd.models.backbone import ResBlock
    from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
    from medicaldetectiontoolkit_amd.utils.model_utils import NDConvGenerator
    torch.manual_seed(9)
    blk = ResBlock(72, 18, conv=NDConvGenerator(3)).to(cuda).to(memory_format=torch.channels_last_3d)
    x0 = torch.randn((1, 72, 32, 32, 64), device=cuda).contiguous(memory_format=torch.channels_last_3d)
    pre = torch.nn.Conv3d(72, 72, 1).to(cuda).to(memory_format=torch.channels_last_3d)      # so that the block input needs a gradient, as inside the net
    res = []
    for flag in (True, False):
        fe.RES_TAP = flag
        blk.zero_grad(); pre.zero_grad()
        x = x0.clone().requires_grad_(True)
        xin = pre(x)
        assert fe.res_tap_applies(blk.conv1, xin) == flag
        y = blk(xin)
        (y.square().mean() * 100).backward()
        res.append(([y.detach().clone(), x.grad.clone()], {n: p.grad.clone() for n, p in list(blk.named_parameters()) + list(pre.named_parameters())}))
    fe.RES_TAP = True
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max())), float((a - b).abs().max())
    for n in res[0][1]:
        a, b = res[0][1][n], res[1][1][n]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()) + 1e-9), (n, float((a - b).abs().max()))


@pytest.mark.parametrize("shape,scale", [((2, 36, 8, 12, 6), 2), ((1, 36, 4, 4, 8), (2, 2, 1)), ((3, 8, 6, 10), 2), ((2, 36, 16, 16, 8), (1, 2, 2))])
def test_lateral_with_the_upsampling_inside_the_epilogue(shape, scale, cuda):
    """conv_bias_add_upsampled (the FPN's top-down step, models/backbone.py:147-153) against conv + F.interpolate + add: forward bit-equal
    (same two additions in the same order), input / coarse-map gradients bit-equal to the materialised form's, bias gradient to
    summation-order rounding; the up-sampled map is never allocated"""
    nd = len(shape) - 2
    sc = (scale,) * nd if isinstance(scale, int) else scale
    mf = torch.channels_last_3d if nd == 3 else torch.channels_last
    g = torch.Generator(device=cuda).manual_seed(5 + len(shape))
    cin = 12
    x0 = torch.randn((shape[0], cin) + shape[2:], device=cuda, generator=g).contiguous(memory_format=mf)
    coarse0 = torch.randn(shape[:2] + tuple(s // k for s, k in zip(shape[2:], sc)), device=cuda, generator=g).contiguous(memory_format=mf)
    conv = (fe.ConvBias3d if nd == 3 else fe.ConvBias2d)(cin, shape[1], 1).to(cuda).to(memory_format=mf)
    flat = torch.zeros(shape[1] + 1, device=cuda)
    flat[1:].copy_(conv.bias.detach())
    conv.bias.data = flat[1:]              # a bias inside a flat parameter buffer (training.FlatAdam): 4-byte aligned only
    gy = torch.randn(shape, device=cuda, generator=g).contiguous(memory_format=mf)

    def run(fused):
        fe.LATERAL_UPSAMPLE_FUSED = fused
        try:
            conv.zero_grad()
            x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            c = coarse0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            torch.cuda.reset_peak_memory_stats()
            m0 = torch.cuda.memory_allocated()
            y = fe.conv_bias_add_upsampled(conv, x, c * 1.0, sc)
            peak = torch.cuda.max_memory_allocated() - m0
            y.backward(gy)
            return y.detach().clone(), x.grad.clone(), c.grad.clone(), conv.bias.grad.clone(), conv.weight.grad.clone(), peak
        finally:
            fe.LATERAL_UPSAMPLE_FUSED = True

    yf, gxf, gcf, gbf, gwf, peak_f = run(True)
    yr, gxr, gcr, gbr, gwr, peak_r = run(False)
    ref = F.conv3d(x0, conv.weight, conv.bias) if nd == 3 else F.conv2d(x0, conv.weight, conv.bias)
    ref = ref + F.interpolate(coarse0, scale_factor=tuple(float(v) for v in sc))
    assert torch.equal(yf, yr)
    ref = ref.detach()
    assert float((yf - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(gxf, gxr) and torch.equal(gcf, gcr)
    assert float((gwf - gwr).abs().max()) <= 1e-5 * float(gwr.abs().max())      # (MIOpen's 2D weight gradient is not run-to-run bit-stable)
    assert float((gbf - gbr).abs().max()) <= 1e-5 * float(gy.abs().sum(dim=[0] + list(range(2, 2 + nd))).max())
    assert peak_f < peak_r       # no up-sampled copy of the coarse map


@pytest.mark.parametrize("shape", [(2, 36, 8, 8, 16), (1, 36, 5, 7, 3), (3, 18, 9, 11), (2, 48, 4, 4, 4), (8, 36, 32, 32, 16)])
def test_row_major_output_gradient_of_a_channels_last_bias_layer(shape, cuda):
    """mdt_bias_grad_to_channels_last: the layout change is exact, the bias gradient equals a float64 sum to fp32 rounding and is run-to-run
    identical; inside autograd a row-major gradient into a channels-last ConvBias layer gives what the two-pass form gives"""
    nd = len(shape) - 2
    mf = torch.channels_last_3d if nd == 3 else torch.channels_last
    g = torch.Generator(device=cuda).manual_seed(11 + shape[1] + shape[0])
    gy = torch.randn(shape, device=cuda, generator=g)                     # row-major
    gx, gb = fe.bias_grad_to_channels_last(gy, mf)
    assert gx.is_contiguous(memory_format=mf) and torch.equal(gx, gy)
    ref = gy.double().sum(dim=[0] + list(range(2, 2 + nd)))
    bound = 1e-6 * float(gy.double().abs().sum(dim=[0] + list(range(2, 2 + nd))).max())
    assert float((gb.double() - ref).abs().max()) <= bound
    gx2, gb2 = fe.bias_grad_to_channels_last(gy, mf)
    assert torch.equal(gb, gb2)
    # inside autograd
    cin = 8
    conv = (fe.ConvBias3d if nd == 3 else fe.ConvBias2d)(cin, shape[1], 1).to(cuda).to(memory_format=mf)
    x0 = torch.randn((shape[0], cin) + shape[2:], device=cuda, generator=g).contiguous(memory_format=mf)

    def run(on):
        fe.BIAS_GRAD_TRANSPOSE = on
        try:
            conv.zero_grad()
            x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            conv(x).backward(gy)
            return x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()
        finally:
            fe.BIAS_GRAD_TRANSPOSE = True

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0])
    assert float((a[1] - b[1]).abs().max()) <= 1e-5 * float(b[1].abs().max())      # (MIOpen's 2D weight gradient is not run-to-run bit-stable)
    assert float((a[2] - b[2]).abs().max()) <= bound


def test_bias_only_backward_returns_the_output_gradient_itself(cuda):
    """no activation: the input gradient is gy -- the kernel reduces and stores nothing (gx == NULL in the C-ABI); same numbers as the copying form"""
    g = torch.Generator(device=cuda).manual_seed(3)
    shape = (2, 36, 8, 8, 16)
    x0 = torch.randn(shape, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    b0 = torch.randn(36, device=cuda, generator=g)
    gy = torch.randn(shape, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)

    def run(on):
        fe.BIAS_BWD_NO_COPY = on
        try:
            x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            b = b0.clone().requires_grad_(True)
            fe.bias_act(x * 1.0, b, None, False).backward(gy)
            return x.grad.clone(), b.grad.clone()
        finally:
            fe.BIAS_BWD_NO_COPY = True

    a, c = run(True), run(False)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and torch.equal(a[0], gy)
    # relu without an output buffer is an argument error, not a silent skip
    from medicaldetectiontoolkit_amd import _lib
    ws = torch.empty(4096 * 36 + 64, device=cuda)
    rc = _lib.lib().mdt_bias_act_backward(None, gy.data_ptr(), x0.data_ptr(), b0.data_ptr(), gy.numel(), 36, 1, 1, ws.data_ptr(), ws.numel() * 4, _lib.raw_stream())
    assert rc == -1          # MDT_ERR_INVALID_ARGUMENT (include/mdt_hip.h)


def test_stem_and_pooling_as_one_node_equals_the_two_nodes(cuda):
    """_ConvStemBiasReLUPool: ReLU mask / bias gradient at the pooled resolution before the pooling backward.  Output and weight gradient bit-equal to
    _ConvStemBiasReLU + _MaxPoolK3S221 (a 0/1 mask commutes with the pooling backward's sums), bias gradient to summation-order rounding; an
    FPN built on it gives the same pyramid as with the switch off"""
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    torch.manual_seed(12)
    conv = mutils.NDConvGenerator(3)
    stem = conv(1, 18, ks=7, stride=(2, 2, 1), pad=3, norm=None, relu="relu").to(cuda).to(memory_format=torch.channels_last_3d)
    pool = fe.MaxPool3dStem(kernel_size=3, stride=(2, 2, 1), padding=1)
    x = torch.randn(2, 1, 64, 64, 32, device=cuda)
    assert fe.stem_pool_fused_applies(stem, pool, x)
    gp = torch.randn(2, 18, 16, 16, 32, device=cuda).contiguous(memory_format=torch.channels_last_3d)

    def run(fused):
        stem.zero_grad()
        p = fe.conv_stem_bias_relu_pool(stem, x) if fused else pool(stem(x))
        p.backward(gp)
        return p.detach().clone(), stem[0].weight.grad.clone(), stem[0].bias.grad.clone()

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert float((a[2] - b[2]).abs().max()) <= 1e-5 * float(gp.abs().sum(dim=(0, 2, 3, 4)).max())
    a2 = run(True)
    assert torch.equal(a2[2], a[2])
