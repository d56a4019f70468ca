"""conv1x1_dgrad_add (csrc/epilogue.hip) against what it replaces (MIOpen input gradient as a forward convolution + the accumulation add) on
the C2 / C3 maps of the benchmarked configuration; event-timed.  One JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
import torch.nn.functional as F
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def tm(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2] * 1e3


for cout, cin, sp in ((18, 72, (8, 32, 32, 128)), (36, 144, (8, 16, 16, 64))):
    gy = torch.randn((sp[0], cout) + sp[1:], device=dev).contiguous(memory_format=torch.channels_last_3d)
    res = torch.randn((sp[0], cin) + sp[1:], device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((cout, cin, 1, 1, 1), device=dev)
    wf = fe.flip_transpose_filter(w, torch.channels_last_3d)
    t_fused = tm(lambda: fe.conv1x1_dgrad_add(gy, w, res))
    t_conv = tm(lambda: F.conv3d(gy, wf))
    gx = F.conv3d(gy, wf)
    t_add = tm(lambda: gx.add_(res))
    byts = 4.0 * (gy.numel() + 2 * res.numel())
    print(json.dumps({"layer": "%d<-%d on %s" % (cin, cout, "x".join(map(str, sp))), "fused_us": round(t_fused, 1), "miopen_dgrad_as_fwd_us": round(t_conv, 1),
                      "add_us": round(t_add, 1), "fused_GBps": round(byts / t_fused / 1e3, 1)}), flush=True)
