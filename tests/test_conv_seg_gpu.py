"""csrc/conv_seg.hip: the Retina U-Net's segmentation branch final_conv(P0_conv2(p0)) (models/retina_unet.py:483-486) as ONE 36 -> 2 3x3x3 layer.  Kernel level:
forward / input gradient / weight + bias gradient against float64 within fp32 summation-order bounds, run-to-run identical.  Module level: the composed layer
against the two layers as the reference runs them, outputs and all six gradients (input, both filters, both biases) through autograd."""
import pytest
import torch
import torch.nn.functional as F

from medicaldetectiontoolkit_amd import _lib
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (1, 2, 4, 64), (3, 6, 12, 32)])
def test_conv_seg_kernels_vs_float64(shape, cuda):
    B, Y, X, Z = shape
    g_ = torch.Generator(device=cuda).manual_seed(sum(shape))
    x = torch.randn(B, 36, Y, X, Z, device=cuda, generator=g_).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(2, 36, 3, 3, 3, device=cuda, generator=g_) * 0.1
    b = torch.randn(2, device=cuda, generator=g_)
    gy = torch.randn(B, 2, Y, X, Z, device=cuda, generator=g_).contiguous(memory_format=torch.channels_last_3d)
    L = _lib.lib()
    assert L.mdt_conv_seg_supported(36, 2, Y, X, Z) and not L.mdt_conv_seg_supported(36, 3, Y, X, Z) and not L.mdt_conv_seg_supported(36, 2, Y, X, 48)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    ref = F.conv3d(xr, wr, br, 1, 1)
    ref.backward(gy.double())
    xa = x.double().abs().requires_grad_(True)
    wa = w.double().abs().requires_grad_(True)
    mag = F.conv3d(xa, wa, b.double().abs(), 1, 1)
    mag.backward(gy.double().abs())

    def run():
        xx = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
        ww = w.clone().requires_grad_(True)
        bb = b.clone().requires_grad_(True)
        y = fe._ConvSeg.apply(xx, ww, bb)
        y.backward(gy)
        return y.detach(), xx.grad, ww.grad, bb.grad

    y, gx, gw, gb = run()
    assert y.is_contiguous(memory_format=torch.channels_last_3d)
    assert bool(((y.double() - ref.detach()).abs() <= 2e-6 * mag.detach() + 1e-7).all())
    assert bool(((gx.double() - xr.grad).abs() <= 2e-6 * xa.grad + 1e-7).all())
    assert bool(((gw.double() - wr.grad).abs() <= 4e-6 * wa.grad + 1e-6).all())
    assert float((gb.double() - br.grad).abs().max()) <= 4e-6 * float(gy.double().abs().sum(dim=(0, 2, 3, 4)).max()) + 1e-6
    y2, gx2, gw2, gb2 = run()
    assert torch.equal(y, y2) and torch.equal(gx, gx2) and torch.equal(gw, gw2) and torch.equal(gb, gb2)


def test_composed_segmentation_head_equals_the_two_layers(cuda):
    mf = torch.channels_last_3d
    torch.manual_seed(6)
    conv2 = fe.ConvBias3d(36, 36, 3, padding=1).to(cuda).to(memory_format=mf)
    final = fe.ConvBias3d(36, 2, 1).to(cuda).to(memory_format=mf)
    x0 = torch.randn(2, 36, 16, 16, 32, device=cuda).contiguous(memory_format=mf)
    gy = torch.randn(2, 2, 16, 16, 32, device=cuda).contiguous(memory_format=mf)
    assert fe.seg_head_composed_applies(conv2, final, x0)

    def run(composed):
        conv2.zero_grad(); final.zero_grad()
        x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        y = fe.seg_head_composed(conv2, final, x * 1.0) if composed else final(conv2(x * 1.0))
        y.backward(gy)
        return [y.detach().clone(), x.grad.clone(), conv2.weight.grad.clone(), conv2.bias.grad.clone(), final.weight.grad.clone(), final.bias.grad.clone()]

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) <= 2e-4 * float(v.abs().max()) + 1e-6
