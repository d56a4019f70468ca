"""The few-channel 3x3x3 layers (18 -> 18 on C2 of the Mask R-CNN backbone, 8 x 32 x 32 x 128; 18 -> 18 on the Retina U-Net's full-resolution C0 map): the
unit-stride window kernel (mdt_conv_win_forward, csrc/conv_s221.hip) against csrc/conv3x3x3_small.hip and MIOpen, event-timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MDT_MIOPEN_SKIP_NAIVE", "1")
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
import torch.nn.functional as F
from medicaldetectiontoolkit_amd import _lib
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
L = _lib.lib()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (B, C, Co, Y, X, Z) in [(8, 18, 18, 32, 32, 128), (8, 18, 18, 128, 128, 128), (8, 36, 36, 32, 32, 128), (8, 36, 18, 64, 64, 128), (8, 36, 32, 32, 32, 128), (8, 36, 18, 32, 32, 128), (8, 36, 32, 16, 16, 64)]:
    x = torch.randn(B, C, Y, X, Z, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Co, C, 3, 3, 3, device=dev) * 0.1
    wt = w.permute(2, 3, 4, 1, 0).contiguous()
    y = torch.empty((B, Co, Y, X, Z), device=dev).contiguous(memory_format=torch.channels_last_3d)
    ok = L.mdt_conv_win_forward_supported(Y, X, Z, C, Co, 3)
    t_win = timed(lambda: L.mdt_conv_win_forward(x.data_ptr(), wt.data_ptr(), None, 0, y.data_ptr(), B, Y, X, Z, C, Co, 3, _lib.raw_stream())) if ok else float("nan")
    prev = fe.CONV_WIN
    fe.CONV_WIN = False
    small = fe.conv3x3x3_small(x, w)
    t_small = timed(lambda: fe.conv3x3x3_small(x, w)) if small is not None else float("nan")
    fe.CONV_WIN = prev
    t_mi = timed(lambda: F.conv3d(x, w, None, 1, 1), reps=5)
    ref = F.conv3d(x, w, None, 1, 1)
    gf = 2.0 * B * Y * X * Z * C * Co * 27 / 1e9
    print("%s %d->%d: window kernel %.0f us (%.1f TF/s)   conv3x3x3_small %.0f us   MIOpen %.0f us   max |diff| %.3g of %.3g" % (
        (B, Y, X, Z), C, Co, t_win, gf / t_win * 1e3 / 1e3, t_small, t_mi, float((y - ref).abs().max()) if ok else float("nan"), float(ref.abs().max())), flush=True)

for (B, C, Co, Y, X, Z) in [(8, 18, 18, 32, 32, 128), (8, 18, 18, 128, 128, 128), (8, 36, 18, 64, 64, 128), (8, 36, 32, 32, 32, 128)]:
    x = torch.randn(B, C, Y, X, Z, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(Co, C, 3, 3, 3, device=dev) * 0.1
    gy = torch.randn(B, Co, Y, X, Z, device=dev).contiguous(memory_format=torch.channels_last_3d)
    g1 = fe.conv_win_weight_grad(gy, x, w)
    t_win = timed(lambda: fe.conv_win_weight_grad(gy, x, w)) if g1 is not None else float("nan")
    g2 = fe.conv3x3x3_small_weight_grad(gy, x, w)
    t_small = timed(lambda: fe.conv3x3x3_small_weight_grad(gy, x, w)) if g2 is not None else float("nan")
    ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False])[1]
    t_mi = timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]), reps=5)
    print("wgrad %s %d->%d: window kernel %.0f us   conv3x3x3_small_wgrad %.0f us   MIOpen %.0f us   max |diff| %.3g of %.3g" % (
        (B, Y, X, Z), C, Co, t_win, t_small, t_mi, float((g1 - ref).abs().max()) if g1 is not None else float("nan"), float(ref.abs().max())), flush=True)
