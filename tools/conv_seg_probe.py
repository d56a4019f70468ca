"""times the three kernels of csrc/conv_seg.hip (the Retina U-Net's composed segmentation layer, 36 <-> 2 channels) and of csrc/conv_c0.hip at 8 x 128^3"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import _lib
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
dev = torch.device("cuda:0")
mf = torch.channels_last_3d


def timed(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


B, Y, X, Z = 8, 128, 128, 128
x = torch.randn(B, 36, Y, X, Z, device=dev).contiguous(memory_format=mf)
w = torch.randn(2, 36, 3, 3, 3, device=dev) * 0.1
b = torch.randn(2, device=dev)
gy = torch.randn(B, 2, Y, X, Z, device=dev).contiguous(memory_format=mf)
L = _lib.lib()
wt = w.permute(2, 3, 4, 1, 0).contiguous()
wd = w.flip(2, 3, 4).permute(2, 3, 4, 0, 1).contiguous()
y = torch.empty(B, 2, Y, X, Z, device=dev).contiguous(memory_format=mf)
gx = torch.empty_like(x)
gw = torch.empty(2, 36, 3, 3, 3, device=dev)
gb = torch.empty(2, device=dev)
wsb = L.mdt_conv_seg_wgrad_workspace_bytes(B, Y, X, Z)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
S = _lib.raw_stream()
print("conv_seg forward        %.2f ms" % timed(lambda: L.mdt_conv_seg_forward(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), B, Y, X, Z, 36, 2, S)))
print("conv_seg input gradient %.2f ms" % timed(lambda: L.mdt_conv_seg_input_grad(gy.data_ptr(), wd.data_ptr(), gx.data_ptr(), B, Y, X, Z, 36, 2, S)))
print("conv_seg weight gradient %.2f ms" % timed(lambda: L.mdt_conv_seg_weight_grad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, Y, X, Z, 36, 2, ws.data_ptr(), wsb, S)))
del x, gx
x1 = torch.randn(B, 1, Y, X, Z, device=dev)
w1 = torch.randn(27, 18, device=dev) * 0.1
b1 = torch.randn(18, device=dev)
y1 = torch.empty(B, 18, Y, X, Z, device=dev).contiguous(memory_format=mf)
g1 = torch.randn(B, 18, Y, X, Z, device=dev).contiguous(memory_format=mf)
gw1 = torch.empty(18, 27, device=dev)
gb1 = torch.empty(18, device=dev)
wsb1 = L.mdt_conv_c0_wgrad_workspace_bytes(B, Y, X, Z)
ws1 = torch.empty(wsb1, dtype=torch.uint8, device=dev)
print("conv_c0 forward  %.2f ms" % timed(lambda: L.mdt_conv_c0_forward(x1.data_ptr(), w1.data_ptr(), b1.data_ptr(), 1, y1.data_ptr(), B, Y, X, Z, 18, S)))
print("conv_c0 backward %.2f ms" % timed(lambda: L.mdt_conv_c0_backward(g1.data_ptr(), y1.data_ptr(), x1.data_ptr(), 1, gw1.data_ptr(), gb1.data_ptr(), B, Y, X, Z, 18, ws1.data_ptr(), wsb1, S)))
