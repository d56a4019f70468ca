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
