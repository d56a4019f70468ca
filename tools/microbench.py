"""Per-kernel timing on the GPU box (SURVEY.md 8(d) micro-benchmarks).  Prints one JSON line per case.
Usage: python tools/microbench.py [--iters 50]"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import _lib  # noqa: E402
from medicaldetectiontoolkit_amd.cuda_functions import _nms_impl, _roi_align_impl  # noqa: E402
from medicaldetectiontoolkit_amd.utils.synthetic_data import nms_boxes, random_boxes_3d, trainlike_rois_3d  # noqa: E402

HBM_PEAK = 8.0e12


def timeit(fn, iters, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)  # us
    return t[len(t) // 2], t[0], sum(t) / len(t)


def report(name, us, bytes_=None, **kw):
    rec = {"case": name, "median_us": round(us[0], 2), "min_us": round(us[1], 2), "mean_us": round(us[2], 2)}
    if bytes_:
        rec["alg_MB"] = round(bytes_ / 1e6, 2)
        rec["GBps"] = round(bytes_ / us[0] / 1e3, 1)
        rec["frac_of_8TBps"] = round(bytes_ / (us[0] * 1e-6) / HBM_PEAK, 4)
    rec.update(kw)
    print(json.dumps(rec), flush=True)


def ref_lib(name):
    p = os.path.join(ROOT, "oracle", "_ref", name)
    return ctypes.CDLL(p) if os.path.exists(p) else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    vp = ctypes.c_void_p
    levels = {"P2": (32, 32, 128), "P3": (16, 16, 64), "P4": (8, 8, 32), "P5": (4, 4, 16)}
    B, C = 8, 36
    refra = ref_lib("libref_gpu_roialign3d.so")

    for lvl, (Y, X, Z) in levels.items():
        for N, crop in ((48, (14, 14, 5)), (48, (7, 7, 3)), (600, (7, 7, 3)), (240, (14, 14, 5)), (4096, (7, 7, 3))):
            if lvl != "P2" and N > 48:
                continue
            shape = (B, C, Y, X, Z)
            boxes = torch.from_numpy(random_boxes_3d(rng, N)).to(dev)
            box_ind = torch.from_numpy(rng.integers(0, B, size=N).astype(np.int32)).to(dev)
            image = torch.randn(shape, device=dev)
            P = crop[0] * crop[1] * crop[2]
            g = torch.randn((N, C) + crop, device=dev)
            V = Y * X * Z
            bwd_bytes = 4 * N * C * P + 4 * B * C * V + 28 * N
            fwd_bytes = 4 * N * C * P + 28 * N  # + touched input voxels (reported separately)
            tag = "%s_N%d_%s" % (lvl, N, "x".join(map(str, crop)))
            report("roialign3d_bwd_fast_" + tag,
                   timeit(lambda: _roi_align_impl.crop_backward(g, boxes, box_ind, shape), args.iters), bwd_bytes)
            report("roialign3d_bwd_twophase_r1_" + tag,
                   timeit(lambda: _roi_align_impl.crop_backward(g, boxes, box_ind, shape, mode="twophase"), args.iters), bwd_bytes)
            if N == 48:
                # train-realistic placement: 6 sampled RoIs per batch element (train_rois_per_image), clustered around
                # one object, sized as the level rule sends them to this level (box side ~ anchor scale of the level)
                tb, ti = trainlike_rois_3d(rng, B, 6, {"P2": 8.0, "P3": 16.0, "P4": 32.0, "P5": 64.0}[lvl])
                tboxes, tind = torch.from_numpy(tb).to(dev), torch.from_numpy(ti).to(dev)
                report("roialign3d_bwd_fast_trainlike_" + tag,
                       timeit(lambda: _roi_align_impl.crop_backward(g, tboxes, tind, shape), args.iters), bwd_bytes)
                report("roialign3d_bwd_twophase_r1_trainlike_" + tag,
                       timeit(lambda: _roi_align_impl.crop_backward(g, tboxes, tind, shape, mode="twophase"), args.iters), bwd_bytes)
            report("roialign3d_bwd_ordered_" + tag,
                   timeit(lambda: _roi_align_impl.crop_backward(g, boxes, box_ind, shape, mode="ordered"), args.iters), bwd_bytes)
            if N <= 48:
                report("roialign3d_bwd_atomic_" + tag,
                       timeit(lambda: _roi_align_impl.crop_backward(g, boxes, box_ind, shape, mode="atomic"), args.iters), bwd_bytes)
            os.environ["MDT_FWD_KERNEL"] = "staged"
            report("roialign3d_fwd_staged_" + tag,
                   timeit(lambda: _roi_align_impl.crop_forward(image, boxes, box_ind, crop), args.iters), fwd_bytes)
            os.environ["MDT_FWD_KERNEL"] = "direct"
            report("roialign3d_fwd_direct_" + tag,
                   timeit(lambda: _roi_align_impl.crop_forward(image, boxes, box_ind, crop), args.iters), fwd_bytes)
            os.environ.pop("MDT_FWD_KERNEL")
            if refra is not None and N <= 600:
                out = torch.empty(shape, device=dev)

                def ref_bwd():
                    # the reference zero-fills twice (crop_and_resize.py:40, crop_and_resize_gpu.c:61)
                    out.zero_()
                    out.zero_()
                    refra.CropAndResizeBackpropImageLaucher(
                        vp(g.data_ptr()), vp(boxes.data_ptr()), vp(box_ind.data_ptr()), N, B, Y, X, Z,
                        crop[0], crop[1], crop[2], C, vp(out.data_ptr()), vp(torch.cuda.current_stream().cuda_stream))
                report("REF_cuda_kernel_roialign3d_bwd_" + tag, timeit(ref_bwd, args.iters), bwd_bytes)
                crops = torch.empty((N, C) + crop, device=dev)

                def ref_fwd():
                    crops.zero_()
                    refra.CropAndResizeLaucher(
                        vp(image.data_ptr()), vp(boxes.data_ptr()), vp(box_ind.data_ptr()), N, B, Y, X, Z,
                        crop[0], crop[1], crop[2], C, ctypes.c_float(0), vp(crops.data_ptr()),
                        vp(torch.cuda.current_stream().cuda_stream))
                report("REF_cuda_kernel_roialign3d_fwd_" + tag, timeit(ref_fwd, args.iters), fwd_bytes)

    # all pyramid levels in one launch vs one launch per level (train-like: 48 sampled RoIs routed by the level rule)
    for crop in ((7, 7, 3), (14, 14, 5)):
        shapes = [(B, C) + levels[k] for k in ("P2", "P3", "P4", "P5")]
        per = []
        for li, side in enumerate((8.0, 16.0, 32.0, 64.0)):
            tb, ti = trainlike_rois_3d(rng, B, 6, side)
            keep = rng.permutation(len(tb))[:{0: 24, 1: 12, 2: 8, 3: 4}[li]]
            per.append((tb[keep], ti[keep], np.full(len(keep), li, dtype=np.int32)))
        pb = torch.from_numpy(np.concatenate([x[0] for x in per])).to(dev)
        pi = torch.from_numpy(np.concatenate([x[1] for x in per])).to(dev)
        pl = torch.from_numpy(np.concatenate([x[2] for x in per])).to(dev)
        maps = [torch.randn(sh, device=dev) for sh in shapes]
        g = torch.randn((48, C) + crop, device=dev)
        P = crop[0] * crop[1] * crop[2]
        nbytes = 4 * 48 * C * P + sum(4 * int(np.prod(sh)) for sh in shapes) + 36 * 48
        inds = [torch.where(pl == li, pi, torch.full_like(pi, -1)) for li in range(4)]
        tag = "N48_" + "x".join(map(str, crop))
        report("pyramid_roialign3d_bwd_one_launch_" + tag,
               timeit(lambda: _roi_align_impl.pyramid_backward(g, pb, pi, pl, shapes), args.iters), nbytes)
        report("pyramid_roialign3d_bwd_four_launches_" + tag,
               timeit(lambda: [_roi_align_impl.crop_backward(g, pb, inds[li], shapes[li]) for li in range(4)], args.iters), nbytes)
        report("pyramid_roialign3d_fwd_one_launch_" + tag,
               timeit(lambda: _roi_align_impl.pyramid_forward(maps, pb, pi, pl, crop), args.iters), 4 * 48 * C * P)
        report("pyramid_roialign3d_fwd_four_launches_" + tag,
               timeit(lambda: [_roi_align_impl.crop_forward(maps[li], pb, inds[li], crop) for li in range(4)], args.iters),
               4 * 48 * C * P * 4)

    # fill-only ceiling for the P2 gradient map (what a perfect bwd could reach)
    out = torch.empty((B, C, 32, 32, 128), device=dev)
    report("torch_zero_fill_151MB", timeit(lambda: out.zero_(), args.iters), out.numel() * 4)

    # NMS
    for n, thresh in ((300, 1e-5), (6000, 0.7), (50000, 1e-5)):
        dets = nms_boxes(rng, n)
        ds = torch.from_numpy(dets[np.argsort(-dets[:, -1].astype(np.float64), kind="stable")]).to(dev)
        cb = (n + 63) // 64
        nb = 28 * n + 8 * n * cb + 8 * n
        report("nms3d_full_N%d_t%g" % (n, thresh), timeit(lambda: _nms_impl.nms_sorted(ds, thresh, 3), args.iters), nb)
        report("nms3d_keep75_N%d_t%g" % (n, thresh),
               timeit(lambda: _nms_impl.nms_sorted(ds, thresh, 3, max_keep=75), args.iters), nb)


if __name__ == "__main__":
    main()
