"""The Retina U-Net's C1 forward at the benchmarked size (8 x 18 x 128^3): this repo's fp32-MFMA kernel against MIOpen on the space-to-depth problem, event-timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MDT_MIOPEN_SKIP_NAIVE", "1")
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
import torch.nn.functional as F
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
x = torch.randn(8, 18, 128, 128, 128, device=dev).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(18, 18, 7, 7, 7, device=dev) * 0.05


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_own = timed(lambda: fe.s221_forward(x, w))
xs, ws = fe.s2d_input(x, 7), fe.s2d_filter(w)
t_ck = timed(lambda: F.conv3d(xs, ws, None, 1, 0))
t_s2d = timed(lambda: fe.s2d_input(x, 7))
y1, y2 = fe.s221_forward(x, w), F.conv3d(xs, ws, None, 1, 0)
print("own forward %.2f ms (%.1f TF/s on 0.93 TFLOP)   MIOpen on the space-to-depth problem %.2f ms + %.2f ms for the space-to-depth copy   max |diff| %.3g of max %.3g" % (
    t_own, 0.93e3 / t_own, t_ck, t_s2d, float((y1 - y2).abs().max()), float(y2.abs().max())))

gy = torch.randn(8, 18, 64, 64, 128, device=dev).contiguous(memory_format=torch.channels_last_3d)
t_own = timed(lambda: fe.s221_input_grad(gy, w, x.shape))
t_ck = timed(lambda: fe.s2d_input_grad_fold(fe.s2d_input_grad_conv(gy, ws), x.shape, 7))
g1, g2 = fe.s221_input_grad(gy, w, x.shape), fe.s2d_input_grad_fold(fe.s2d_input_grad_conv(gy, ws), x.shape, 7)
print("own input gradient %.2f ms (%.1f TF/s on 0.93 TFLOP)   MIOpen forward convolution on the padded output gradient + fold %.2f ms   max |diff| %.3g of max %.3g" % (
    t_own, 0.93e3 / t_own, t_ck, float((g1 - g2).abs().max()), float(g2.abs().max())))
