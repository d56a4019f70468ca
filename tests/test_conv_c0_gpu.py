"""csrc/conv_c0.hip: the one-channel 3x3x3 first layer of the stride-1 backbone (models/backbone.py:60-63, the Retina U-Net's C0[0]) -- forward against a float64
convolution, weight / bias gradient against float64 sums within fp32 summation-order bounds, run-to-run identical, and the module path (ConvBiasReLU through
_ConvC0BiasReLU) against MIOpen + the epilogue kernel through autograd."""
import pytest
import torch
import torch.nn.functional as F

from medicaldetectiontoolkit_amd import _lib
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
from medicaldetectiontoolkit_amd.utils import model_utils as mutils

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 16, 16, 32), (1, 5, 3, 64), (3, 7, 9, 128), (1, 1, 1, 32)])
@pytest.mark.parametrize("relu", [1, 0])
def test_conv_c0_forward_and_backward_vs_float64(shape, relu, cuda):
    B, Y, X, Z = shape
    g = torch.Generator(device=cuda).manual_seed(sum(shape) + relu)
    x = torch.randn(B, 1, Y, X, Z, device=cuda, generator=g)
    w = torch.randn(18, 1, 3, 3, 3, device=cuda, generator=g) * 0.3
    b = torch.randn(18, device=cuda, generator=g)
    L = _lib.lib()
    assert L.mdt_conv_c0_supported(1, 18, 3, Z) and not L.mdt_conv_c0_supported(2, 18, 3, Z) and not L.mdt_conv_c0_supported(1, 18, 3, 48)
    y = torch.full((B, 18, Y, X, Z), 7.0, device=cuda).contiguous(memory_format=torch.channels_last_3d)
    wt = w.reshape(18, 27).t().contiguous()
    assert L.mdt_conv_c0_forward(x.data_ptr(), wt.data_ptr(), b.data_ptr(), relu, y.data_ptr(), B, Y, X, Z, 18, _lib.raw_stream()) == 0
    ref = F.conv3d(x.double(), w.double(), b.double(), 1, 1)
    mag = F.conv3d(x.double().abs(), w.double().abs(), b.double().abs(), 1, 1)
    if relu:
        ref = ref.clamp_min(0)
    assert bool(((y.double() - ref).abs() <= 2e-6 * mag + 1e-7).all())
    # backward
    gy = torch.randn(B, 18, Y, X, Z, device=cuda, generator=g).contiguous(memory_format=torch.channels_last_3d)
    wsb = L.mdt_conv_c0_wgrad_workspace_bytes(B, Y, X, Z)
    ws = torch.empty(wsb, dtype=torch.uint8, device=cuda)

    def run():
        gw = torch.empty(18, 27, device=cuda)
        gb = torch.empty(18, device=cuda)
        rc = L.mdt_conv_c0_backward(gy.data_ptr(), y.data_ptr() if relu else None, x.data_ptr(), relu, gw.data_ptr(), gb.data_ptr(), B, Y, X, Z, 18, ws.data_ptr(), wsb,
                                    _lib.raw_stream())
        assert rc == 0
        return gw, gb

    gw, gb = run()
    gm = (gy * (y > 0) if relu else gy).double()
    xr = x.double().requires_grad_(False)
    wr = w.double().clone().requires_grad_(True)
    F.conv3d(xr, wr, None, 1, 1).backward(gm)
    wa = w.double().abs().clone().requires_grad_(True)
    F.conv3d(xr.abs(), wa, None, 1, 1).backward(gm.abs())
    assert bool(((gw.double().view(18, 1, 3, 3, 3) - wr.grad).abs() <= 4e-6 * wa.grad + 1e-6).all())
    assert float((gb.double() - gm.sum(dim=(0, 2, 3, 4))).abs().max()) <= 4e-6 * float(gm.abs().sum(dim=(0, 2, 3, 4)).max()) + 1e-6
    gw2, gb2 = run()
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)


def test_first_layer_module_on_own_kernels_equals_miopen_path(cuda):
    torch.manual_seed(2)
    seq = mutils.NDConvGenerator(3)(1, 18, ks=3, pad=1, norm=None, relu="relu").to(cuda)
    x = torch.randn(2, 1, 32, 32, 64, device=cuda)
    assert fe.conv_c0_applies(seq, x)
    gy = torch.randn(2, 18, 32, 32, 64, device=cuda).contiguous(memory_format=torch.channels_last_3d)

    def run(own):
        seq.zero_grad()
        y = fe.conv_c0_bias_relu(seq, x) if own else seq(x)
        y.backward(gy)
        return y.detach().clone(), seq[0].weight.grad.clone(), seq[0].bias.grad.clone()

    a, b = run(True), run(False)
    assert a[0].is_contiguous(memory_format=torch.channels_last_3d)
    assert float((a[0] - b[0]).abs().max()) <= 1e-5 * float(b[0].abs().max())
    for u, v in zip(a[1:], b[1:]):
        assert float((u - v).abs().max()) <= 2e-4 * float(v.abs().max())
    assert not fe.conv_c0_applies(seq, x.clone().requires_grad_(True))          # an input that needs a gradient stays on the library path
