"""times mdt_conv1x1_forward on the C2 / C3 bottleneck shapes of the benchmark patch against MIOpen + mdt_bias_act_forward (usage: python tools/conv1x1_fwd_probe.py)"""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import _lib, miopen_env
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe

miopen_env.setup()
dev = torch.device("cuda:0")
mf = torch.channels_last_3d


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (cin, cout, shp, with_res, relu) in [(18, 72, (8, 64, 64, 32), True, True), (18, 72, (8, 64, 64, 32), False, False), (72, 18, (8, 64, 64, 32), False, True),
                                         (36, 144, (8, 32, 32, 16), True, True), (36, 144, (8, 32, 32, 16), False, False)]:
    x = torch.randn((shp[0], cin) + shp[1:], device=dev).contiguous(memory_format=mf)
    w = torch.randn(cout, cin, 1, 1, 1, device=dev) * 0.1
    b = torch.randn(cout, device=dev)
    res = torch.randn((shp[0], cout) + shp[1:], device=dev).contiguous(memory_format=mf) if with_res else None
    V = x.numel() // cin
    out = torch.empty((shp[0], cout) + shp[1:], device=dev).contiguous(memory_format=mf)
    L = _lib.lib()

    def own():
        rc = L.mdt_conv1x1_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr() if with_res else None, out.data_ptr(), V, cin, cout, 1 if relu else 0,
                                   _lib.raw_stream())
        assert rc == 0

    def two_pass():
        y = F.conv3d(x, w)
        L.mdt_bias_act_forward(y.data_ptr(), y.data_ptr(), b.data_ptr(), res.data_ptr() if with_res else None, y.numel(), cout, 1, 1 if relu else 0, _lib.raw_stream())
        return y

    t_own, t_two = timed(own), timed(two_pass)
    mb = (x.numel() + out.numel() * (2 if with_res else 1)) * 4 / 1e6
    ref = two_pass()
    print("%3d -> %3d %s res=%d relu=%d: own %.1f us (%.2f TB/s of %.0f MB)   MIOpen + epilogue %.1f us   max |diff| %.2e" % (
        cin, cout, shp, with_res, relu, t_own, mb / t_own, mb, t_two, float((out - ref).abs().max())))

# the RPN heads on the raw conv_shared output, P2 and P3 of the benchmark patch
L = _lib.lib()
for (B, vox) in [(8, 32 * 32 * 128), (8, 16 * 16 * 64)]:
    h = torch.randn(B * vox, 128, device=dev)
    bs = torch.randn(128, device=dev)
    w = torch.randn(24, 128, device=dev) * 0.1
    b = torch.randn(24, device=dev)
    total = vox * 3
    logits = torch.empty(B, total, 2, device=dev)
    deltas = torch.empty(B, total, 6, device=dev)

    def heads():
        rc = L.mdt_rpn_heads_forward(h.data_ptr(), bs.data_ptr(), w.data_ptr(), b.data_ptr(), logits.data_ptr(), deltas.data_ptr(), B, vox, 128, 6, 18, total, 0, _lib.raw_stream())
        assert rc == 0

    t = timed(heads)
    mb = (h.numel() + logits.numel() + deltas.numel()) * 4 / 1e6
    print("rpn heads %d x %d voxels: %.1f us (%.2f TB/s of %.0f MB)" % (B, vox, t, mb / t, mb))

# conv3 (18 -> 72) backward in one pass, C2 maps of the benchmark patch
V = 8 * 32 * 32 * 128
gy = torch.randn(V, 72, device=dev)
yy = torch.randn(V, 72, device=dev)
w = torch.randn(72, 18, device=dev) * 0.1
g = torch.empty_like(gy)
gx = torch.empty(V, 18, device=dev)
gb = torch.empty(72, device=dev)
wsb = L.mdt_conv1x1_backward_workspace_bytes(V, 72)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
for relu in (1, 0):
    def bwd():
        rc = L.mdt_conv1x1_backward(gy.data_ptr(), yy.data_ptr() if relu else None, w.data_ptr(), g.data_ptr() if relu else None, gx.data_ptr(), gb.data_ptr(), V, 18, 72,
                                    ws.data_ptr(), wsb, _lib.raw_stream())
        assert rc == 0
    t = timed(bwd)
    mb = (gy.numel() * (3 if relu else 1) + gx.numel()) * 4 / 1e6
    print("conv3 backward relu=%d: %.1f us (%.2f TB/s of %.0f MB)" % (relu, t, mb / t, mb))
