"""Does a forward-type 3x3x3 convolution with an awkward filter count (36) get faster as TWO convolutions (32 + 4 filters)?  MIOpen / CK
tile the filter count in 32 / 64: the RPN conv_shared input gradient on P2 (128 -> 36 as a forward convolution, 261 GFLOP) runs at 69 TF/s
against 124 TF/s for its 36 -> 128 forward.  Times every piece with the exhaustive find (channels_last_3d, fp32).  One JSON line per case."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
os.environ.setdefault("MDT_MIOPEN_SKIP_NAIVE", "1")
miopen_env.setup()
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def t_conv(x, w, n=10):
    for _ in range(3):
        F.conv3d(x, w, None, 1, 1)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); F.conv3d(x, w, None, 1, 1); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2] * 1e3


for tag, cin, sp in (("rpn_dgrad_P2", 128, (32, 32, 128)), ("p2_conv2_P2", 36, (32, 32, 128)), ("c3_conv2", 36, (16, 16, 64)), ("rpn_dgrad_P3", 128, (16, 16, 64))):
    x = torch.randn((8, cin) + sp, device=dev).contiguous(memory_format=torch.channels_last_3d)
    rec = {"case": tag, "cin": cin, "spatial": sp}
    for cout in (36, 32, 4, 8, 16, 20, 40):
        w = torch.randn((cout, cin, 3, 3, 3), device=dev).contiguous(memory_format=torch.channels_last_3d)
        rec["cout_%d_us" % cout] = round(t_conv(x, w), 1)
    rec["split_32_4_us"] = round(rec["cout_32_us"] + rec["cout_4_us"], 1)
    rec["split_20_16_us"] = round(rec["cout_20_us"] + rec["cout_16_us"], 1)
    print(json.dumps(rec), flush=True)
