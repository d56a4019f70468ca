"""timings of the fused conv epilogue vs the torch ops it replaces, on the backbone's activation shapes"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for shape in ((8, 18, 64, 64, 128), (8, 72, 32, 32, 128), (8, 36, 32, 32, 128), (8, 144, 16, 16, 64), (8, 288, 8, 8, 32)):
    for cl in (True, False):
        mf = torch.channels_last_3d if cl else torch.contiguous_format
        x = torch.randn(shape, device=dev).contiguous(memory_format=mf)
        bias = torch.randn(shape[1], device=dev, requires_grad=True)
        res = torch.randn(shape, device=dev).contiguous(memory_format=mf)
        gy = torch.randn(shape, device=dev).contiguous(memory_format=mf)
        def fused():
            xx = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
            y = fe.bias_act(xx * 1.0, bias, res, True); y.backward(gy)
        def plain():
            xx = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
            y = F.relu(xx * 1.0 + bias.view(1, -1, 1, 1, 1) + res); y.backward(gy)
        mb = x.numel() * 4 / 1e6
        print(json.dumps({"shape": shape, "channels_last": cl, "MB": round(mb, 1), "fused_us": round(timeit(fused), 1), "torch_us": round(timeit(plain), 1)}), flush=True)
