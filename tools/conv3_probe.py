"""few-channel 3x3x3 convolution: csrc/conv3x3x3_small.hip vs MIOpen (F.conv3d), us per call, B = 8 channels-last"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
import torch.nn.functional as F
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def timeit(fn, reps=10, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    return sorted(ts)[len(ts) // 2]


for cin, cout, sp in [(18, 18, (32, 32, 128)), (18, 18, (64, 64, 128)), (6, 6, (32, 32, 128)), (16, 16, (32, 32, 128))]:
    B = 8 if sp[0] == 32 else 2
    x = torch.randn((B, cin) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev).contiguous(memory_format=torch.channels_last_3d)
    t_new = timeit(lambda: fe.conv3x3x3_small(x, w))
    t_ref = timeit(lambda: F.conv3d(x, w, None, 1, 1))
    gy = torch.randn((B, cout) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
    t_wn = timeit(lambda: fe.conv3x3x3_small_weight_grad(gy, x, w))
    t_wr = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]))
    flop = 2.0 * 27 * cin * cout * B * sp[0] * sp[1] * sp[2]
    print(json.dumps({"cin": cin, "cout": cout, "spatial": sp, "batch": B, "mdt_us": round(t_new, 1), "miopen_us": round(t_ref, 1),
                      "wgrad_mdt_us": round(t_wn, 1), "wgrad_miopen_us": round(t_wr, 1), "mdt_TFLOPs": round(flop / t_new / 1e6, 1), "miopen_TFLOPs": round(flop / t_ref / 1e6, 1)}), flush=True)
