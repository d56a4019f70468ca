"""bias_act backward: one-launch (ticket) form against the two-launch form, event-timed on the step's shapes.  Usage: bias_bwd_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
dev = torch.device("cuda:0")
for shape in [(8, 18, 64, 64, 128), (8, 72, 32, 32, 128), (8, 36, 32, 32, 128), (8, 36, 16, 16, 64), (8, 144, 16, 16, 64), (8, 288, 8, 8, 32), (8, 128, 32, 32, 128), (48, 36, 14, 14, 5)]:
    gy = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last_3d)
    y = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last_3d)
    res = []
    for flag in (True, False):
        fe.BIAS_GRAD_IN_LAUNCH = flag
        for _ in range(3):
            fe._bias_act_bwd(gy, y, True, torch.channels_last_3d)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fe._bias_act_bwd(gy, y, True, torch.channels_last_3d)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    mb = gy.numel() * 4 * 3 / 1e6
    print("%-24s one launch %7.1f us (%5.2f TB/s)   two launches %7.1f us" % (shape, res[0], mb / res[0], res[1]), flush=True)
