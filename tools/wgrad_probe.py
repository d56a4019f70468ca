"""1x1x1 weight gradient: csrc/conv1x1_wgrad.hip vs MIOpen (aten.convolution_backward), us per call, B = 8 channels-last."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def timeit(fn, reps=20, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) * 1e3 / reps)
    return sorted(best)[len(best) // 2]


CASES = [(18, 72, (32, 32, 128)), (72, 18, (32, 32, 128)), (18, 18, (32, 32, 128)), (128, 18, (32, 32, 128)), (72, 36, (32, 32, 128)),
         (36, 144, (16, 16, 64)), (144, 36, (16, 16, 64)), (72, 36, (16, 16, 64)), (72, 288, (8, 8, 32)), (288, 72, (8, 8, 32)), (144, 576, (4, 4, 16)),
         (18, 72, (64, 64, 128))]
for cin, cout, sp in CASES:
    B = 8 if sp[0] <= 32 else 2
    x = torch.randn((B, cin) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
    gy = torch.randn((B, cout) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((cout, cin, 1, 1, 1), device=dev)
    t_new = timeit(lambda: fe.conv1x1_weight_grad(gy, x, w, force=True))
    t_ref = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [0, 0, 0], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]))
    byts = 4.0 * gy.numel() + 4.0 * x.numel()
    print(json.dumps({"cin": cin, "cout": cout, "spatial": sp, "batch": B, "mdt_us": round(t_new, 1), "miopen_us": round(t_ref, 1),
                      "alg_MB": round(byts / 1e6, 1), "mdt_frac_of_8TBps": round(byts / (t_new * 1e-6) / 8e12, 3)}), flush=True)
