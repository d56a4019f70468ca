"""Which MIOpen problem to pose for the layers that lead the step after round 3's own kernels (profiles/r03_op_profile.txt), and
the stem forward kernel (csrc/conv_stem_fwd.hip) against the space-to-depth MIOpen path.  us per call, B = 8, channels-last, fp32.
One JSON line per measurement."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
import torch.nn.functional as F
from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
CL = torch.channels_last_3d


def timeit(fn, reps=8, rounds=5):
    for _ in range(2):
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


def emit(**kw):
    print(json.dumps(kw), flush=True)


def rnd(*shape):
    return torch.randn(shape, device=dev).contiguous(memory_format=CL)


which = set(sys.argv[1:]) or {"stem", "rpn", "p2", "c3", "h2d"}

if "stem" in which:
    x = torch.randn((8, 1, 128, 128, 128), device=dev)
    w = torch.randn((18, 1, 7, 7, 7), device=dev) * 0.05
    b = torch.randn(18, device=dev)
    fe.STEM_FWD = True
    t_k = timeit(lambda: fe.stem_forward(x, w))
    t_kb = timeit(lambda: fe.stem_forward(x, w, b, True))
    xp = F.pad(x.reshape(8, 128, 128, 128), (3, 3, 3, 3, 3, 3)).contiguous()
    t_pad = timeit(lambda: F.pad(x.reshape(8, 128, 128, 128), (3, 3, 3, 3, 3, 3)).contiguous())
    fe.STEM_FWD = False
    t_s2d = timeit(lambda: fe._ConvStem221.apply(x, w))
    fe.STEM_FWD = True
    got = fe.stem_forward(x, w)[0]
    want = F.conv3d(x, w, None, (2, 2, 1), 3)
    emit(case="stem forward 8x1x128^3 -> 18 (pad copy included)", mdt_us=round(t_k, 1), mdt_bias_relu_us=round(t_kb, 1), pad_copy_us=round(t_pad, 1),
         miopen_s2d_us=round(t_s2d, 1), max_abs_err=float((got - want).abs().max()), ref_abs_max=float(want.abs().max()),
         mdt_TFLOPs_useful=round(2.0 * 343 * 18 * 8 * 64 * 64 * 128 / (t_k - t_pad) / 1e6, 1))

if "rpn" in which:
    # RPN conv_shared 36 -> 128 (3x3x3) on P2: its input gradient is posed as a FORWARD convolution 128 -> 36 today (3.9 ms)
    gy = rnd(8, 128, 32, 32, 128)
    x = rnd(8, 36, 32, 32, 128)
    w = rnd(128, 36, 3, 3, 3)
    wt = fe.flip_transpose_filter(w, CL)                              # [36, 128, 3, 3, 3]
    t_fwd = timeit(lambda: F.conv3d(gy, wt, None, 1, 1))
    emit(case="rpn 36->128 input gradient as forward conv 128->36 (today)", us=round(t_fwd, 1))
    t_nat = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [True, False, False]))
    emit(case="rpn 36->128 input gradient, MIOpen backward-data", us=round(t_nat, 1))
    for pad_to in (40, 48, 64):
        wp = torch.zeros((pad_to, 128, 3, 3, 3), device=dev).contiguous(memory_format=CL)
        wp[:36] = wt
        t_p = timeit(lambda: F.conv3d(gy, wp, None, 1, 1))
        t_ps = timeit(lambda: F.conv3d(gy, wp, None, 1, 1)[:, :36].contiguous(memory_format=CL))
        emit(case="rpn input gradient as forward conv 128->%d (zero filters), then slice copy" % pad_to, conv_us=round(t_p, 1), conv_plus_slice_us=round(t_ps, 1))
    t_f = timeit(lambda: F.conv3d(x, w, None, 1, 1))
    emit(case="rpn 36->128 forward (today)", us=round(t_f, 1))
    x40 = rnd(8, 40, 32, 32, 128)
    w40 = rnd(128, 40, 3, 3, 3)
    emit(case="rpn 40->128 forward (input channels padded)", us=round(timeit(lambda: F.conv3d(x40, w40, None, 1, 1)), 1))
    t_w = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]))
    t_w40 = timeit(lambda: torch.ops.aten.convolution_backward(gy, x40, w40, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]))
    emit(case="rpn weight gradient 36->128 / 40->128", us=round(t_w, 1), padded_us=round(t_w40, 1))

if "p2" in which:
    for c in (36, 40, 48):
        x = rnd(8, c, 32, 32, 128)
        w = rnd(c, c, 3, 3, 3)
        gy = rnd(8, c, 32, 32, 128)
        t_f = timeit(lambda: F.conv3d(x, w, None, 1, 1))
        t_w = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]))
        emit(case="P2 post conv %d->%d 3x3x3 on 8x32x32x128" % (c, c), fwd_us=round(t_f, 1), wgrad_us=round(t_w, 1))

if "c3" in which:
    for c in (36, 40):
        x = rnd(8, c, 16, 16, 64)
        w = rnd(c, c, 3, 3, 3)
        gy = rnd(8, c, 16, 16, 64)
        t_f = timeit(lambda: F.conv3d(x, w, None, 1, 1))
        t_w = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]))
        emit(case="C3 conv2 %d->%d 3x3x3 on 8x16x16x64" % (c, c), fwd_us=round(t_f, 1), wgrad_us=round(t_w, 1))
    for c in (72, 80):
        x = rnd(8, c, 8, 8, 32)
        w = rnd(c, c, 3, 3, 3)
        emit(case="C4 conv2 %d->%d 3x3x3 on 8x8x8x32" % (c, c), fwd_us=round(timeit(lambda: F.conv3d(x, w, None, 1, 1)), 1))

if "h2d" in which:
    import numpy as np
    import time
    a = np.random.rand(8, 1, 128, 128, 128).astype(np.float32)
    for name, fn in (("pageable torch.from_numpy(a).to(dev)", lambda: torch.from_numpy(a).to(dev)),
                     ("torch.as_tensor(a, device=dev)", lambda: torch.as_tensor(a, device=dev))):
        fn(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
        emit(case="H2D 67 MB " + name, ms=round(dt * 1e3, 2), GBps=round(a.nbytes / dt / 1e9, 2))
