"""Per-layer timing of the backbone's heaviest convolutions: MIOpen forward / backward-data / backward-weights as torch
calls them, and backward-data computed as a FORWARD convolution of the output gradient with the flipped, transposed
filter (same arithmetic, different MIOpen solver family).  usage: python tools/conv_bwd_probe.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CL = torch.channels_last_3d
B = 8
SHAPES = [  # cin, cout, ks, spatial
    (36, 128, 3, (32, 32, 128)), (18, 18, 3, (32, 32, 128)), (36, 36, 3, (32, 32, 128)), (36, 36, 3, (16, 16, 64)),
    (72, 72, 3, (8, 8, 32)), (36, 128, 3, (16, 16, 64)), (144, 144, 3, (4, 4, 16)), (18, 72, 1, (32, 32, 128)),
    (72, 18, 1, (32, 32, 128)), (128, 18, 1, (32, 32, 128)), (36, 144, 1, (16, 16, 64)), (72, 288, 1, (8, 8, 32)),
    (36, 36, 3, (14, 14, 5), 48),
]


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


print("cin cout ks spatial | fwd  bwd_data  bwd_weight | bwd_data_as_fwd  (us)   max|diff|")
for spec in SHAPES:
    cin, cout, ks, sp = spec[:4]
    B = spec[4] if len(spec) > 4 else 8
    pad = ks // 2
    x = torch.randn((B, cin) + sp, device=dev).contiguous(memory_format=CL)
    w = (torch.randn((cout, cin, ks, ks, ks), device=dev) * 0.05).contiguous(memory_format=CL)
    gy = torch.randn((B, cout) + sp, device=dev).contiguous(memory_format=CL)
    args = ([1, 1, 1], [pad] * 3, [1, 1, 1], False, [0, 0, 0], 1)
    t_f = timeit(lambda: F.conv3d(x, w, padding=pad))
    t_bd = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *args, [True, False, False]))
    t_bw = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *args, [False, True, False]))

    def as_fwd():
        wt = w.flip(2, 3, 4).transpose(0, 1).contiguous(memory_format=CL)
        return F.conv3d(gy, wt, padding=pad)
    t_alt = timeit(as_fwd)
    ref = torch.ops.aten.convolution_backward(gy, x, w, None, *args, [True, False, False])[0]
    err = (as_fwd() - ref).abs().max().item()
    extra = ""
    if ks == 1:      # 1x1 weight gradient as one skinny GEMM over the channels-last activations (views, no copies)
        g2, x2 = gy.permute(0, 2, 3, 4, 1).reshape(-1, cout), x.permute(0, 2, 3, 4, 1).reshape(-1, cin)
        t_mm = timeit(lambda: g2.t() @ x2)
        ref_w = torch.ops.aten.convolution_backward(gy, x, w, None, *args, [False, True, False])[1]
        extra = "  wrw as mm %7.0f us (rel err %.1e)" % (t_mm, ((g2.t() @ x2).view_as(ref_w) - ref_w).abs().max().item() / ref_w.abs().max().item())
    print("%3d %3d %d %-14s | %7.0f %7.0f %7.0f | %7.0f   %.2e%s" % (cin, cout, ks, "x".join(map(str, sp)), t_f, t_bd, t_bw, t_alt, err, extra), flush=True)
