"""Which convolution PROBLEM to pose for the Retina U-Net's C1 layer (backbone.py:84: 18 -> 18, 7x7x7, stride (2, 2, 1), pad 3, on the
full-resolution C0 output)?  40 % of the config-2 step is this one layer on MIOpen / CK (profiles/r04/r04_retina_unet_step_steady_state_kernels.csv:
39.7 ms forward, 58.1 ms input gradient, 46.0 ms weight gradient at 8 x 128^3).

    python tools/c1_probe.py [--batch 8] [--patch 128] [--iters 3]

Times, with MIOpen's find mode on and channels-last storage:
  direct      F.conv3d / aten.convolution_backward on the original problem (what the model ran until round 5)
  s2d         the 2 x 2 (y, x) phases of the padded input as 4x the channels: 72 -> 18, (4, 4, 7), stride 1 -- forward;
              input gradient = a FORWARD convolution of the padded output gradient with the flipped filter (18 -> 72) + depth-to-space;
              weight gradient: this repo's fp32-MFMA kernel (csrc/conv_s221.hip); with --s2d-wgrad also MIOpen on the space-to-depth problem
and checks each s2d result against the direct one.  One JSON line per measurement."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medicaldetectiontoolkit_amd  # noqa: E402,F401
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from medicaldetectiontoolkit_amd.utils import fused_epilogue as fe  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--cin", type=int, default=18)
    ap.add_argument("--s2d-wgrad", action="store_true")
    ap.add_argument("--direct", action="store_true", help="also time MIOpen on the direct problem and compare the results (finds: minutes on a fresh box)")
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    mf = torch.channels_last_3d
    B, C, O, k, P = args.batch, args.cin, 18, 7, args.patch
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((B, C, P, P, P), device=dev, generator=g).contiguous(memory_format=mf)
    w = (torch.randn((O, C, k, k, k), device=dev, generator=g) * 0.02).contiguous(memory_format=mf)
    gy = torch.randn((B, O, P // 2, P // 2, P), device=dev, generator=g).contiguous(memory_format=mf)

    def say(name, ms, **kw):
        print(json.dumps(dict(case=name, ms=round(ms, 3), **kw)), flush=True)

    # this repo's path first (the direct problem's timings need MIOpen finds that take minutes on a box without them in the find-db)
    ms, xs = timed(lambda: fe.s2d_input(x, k), args.iters)
    say("s2d_input_copy", ms, shape=list(xs.shape))
    ms, ws = timed(lambda: fe.s2d_filter(w), args.iters)
    say("s2d_filter", ms, shape=list(ws.shape))
    ms, y1 = timed(lambda: F.conv3d(xs, ws, None, 1, 0), args.iters)
    say("s2d_fwd_conv", ms)
    ms, gxs = timed(lambda: fe.s2d_input_grad_conv(gy, ws), args.iters)
    say("s2d_dgrad_conv_as_fwd", ms, shape=list(gxs.shape))
    ms, gx1 = timed(lambda: fe.s2d_input_grad_fold(gxs, x.shape, k), args.iters)
    say("s2d_dgrad_depth_to_space", ms)
    del gxs
    ms, gw1 = timed(lambda: fe.s221_weight_grad(gy, x, w), args.iters)
    say("own_wgrad_mfma", ms, supported=gw1 is not None)
    if args.direct:
        ms, y0 = timed(lambda: F.conv3d(x, w, None, (2, 2, 1), 3), args.iters)
        say("direct_fwd", ms, s2d_max_abs_err=float((y1 - y0).abs().max()), ref_max=float(y0.abs().max()))
        ms, (gx0, _, _) = timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2, 1], [3, 3, 3], [1, 1, 1], False, [0, 0, 0], 1, [True, False, False]), args.iters)
        say("direct_dgrad", ms, s2d_max_abs_err=float((gx1 - gx0).abs().max()), ref_max=float(gx0.abs().max()))
        ms, (_, gw0, _) = timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2, 1], [3, 3, 3], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]), args.iters)
        say("direct_wgrad", ms, own_max_abs_err=float((gw1 - gw0).abs().max()) if gw1 is not None else None, ref_max=float(gw0.abs().max()))
    if args.s2d_wgrad:        # MIOpen's find for this problem took > 6 minutes on the round-5 box: opt-in
        ms, (_, gws, _) = timed(lambda: torch.ops.aten.convolution_backward(gy, xs, ws, None, [1, 1, 1], [0, 0, 0], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]), args.iters)
        say("s2d_wgrad_conv", ms)


if __name__ == "__main__":
    main()
