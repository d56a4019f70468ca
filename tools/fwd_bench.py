"""RoIAlign-3D forward at the reference's inference / training call sizes, event-timed through the Python boundary (round 5: the
channel-quad kernel of csrc/roi_align_fwd.hip; the round-4 figures of the wave-staged and direct kernels are in profiles/r04/), fp32 and bf16
maps.  One JSON line per case.  Algorithmic bytes as SURVEY.md 8(d) defines them for the forward: 4 N C P written + every TOUCHED input
voxel once (4 C x the voxel box the RoI's samples reach, floor .. ceil per axis) + 28 N; `out_only_frac` is the output bytes alone against
8 TB/s (what VERDICT r4 quoted).  usage: python tools/fwd_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl
from medicaldetectiontoolkit_amd.utils.synthetic_data import random_boxes_3d, trainlike_rois_3d

dev = torch.device("cuda:0")
kern = "cq"
rng = np.random.default_rng(0)
torch.manual_seed(0)
B, C = 8, 36
P2 = torch.randn((B, C, 32, 32, 128), device=dev)
P3 = torch.randn((B, C, 16, 16, 64), device=dev)


def time_op(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return float(np.mean(t)), float(t[len(t) // 2])


def touched_voxels(boxes, shape, crop):
    """per RoI: product over the axes of (max ceil - min floor + 1) of its sample coordinates (crop_and_resize_kernel.cu:51-75)"""
    tot = 0
    for bx in np.asarray(boxes, dtype=np.float64):
        n = 1
        for a, (lo, hi) in enumerate(((0, 2), (1, 3), (4, 5))):
            L, Pn = shape[a], crop[a]
            if Pn > 1:
                sc = (bx[hi] - bx[lo]) * L / Pn
                t = bx[lo] * L + np.arange(Pn) * sc + sc / 2 - 0.5
            else:
                t = np.array([0.5 * (bx[lo] + bx[hi]) * L])
            t = np.clip(t, 0, L - 1)
            n *= int(np.ceil(t).max() - np.floor(t).min() + 1)
        tot += n
    return tot


cases = [("N240_14x14x5_P2_survey_boxes", P2, random_boxes_3d(rng, 240), rng.integers(0, B, 240), (14, 14, 5)),
         ("N600_7x7x3_P2_survey_boxes", P2, random_boxes_3d(rng, 600), rng.integers(0, B, 600), (7, 7, 3)),
         ("N4096_7x7x3_P2_survey_boxes", P2, random_boxes_3d(rng, 4096), rng.integers(0, B, 4096), (7, 7, 3)),
         ("N48_14x14x5_P2_trainlike", P2, *trainlike_rois_3d(rng, B, 6, 8.0, 128.0), (14, 14, 5)),
         ("N600_7x7x3_P3_survey_boxes", P3, random_boxes_3d(rng, 600), rng.integers(0, B, 600), (7, 7, 3))]
for tag, img, boxes, ind, crop in cases:
    bx, bi = torch.from_numpy(np.ascontiguousarray(boxes)).to(dev), torch.from_numpy(np.asarray(ind, dtype=np.int32)).to(dev)
    for dt in ("f32", "bf16"):
        im = img if dt == "f32" else img.bfloat16()
        out = _roi_align_impl.crop_forward(im, bx, bi, crop)
        mean, med = time_op(lambda: _roi_align_impl.crop_forward(im, bx, bi, crop))
        esz = 4 if dt == "f32" else 2
        tv = touched_voxels(boxes, tuple(img.shape[2:]), crop)
        byts = 4.0 * out.numel() + esz * C * tv + 28 * len(boxes)
        # the same RoIs on the SAME map in channels-last storage (what the conv path produces): mdt_pyramid_roi_align_forward_cl, one level
        im_cl = im.contiguous(memory_format=torch.channels_last_3d)
        lvl0 = torch.zeros(len(boxes), dtype=torch.int32, device=dev)
        out_cl = _roi_align_impl.pyramid_forward([im_cl], bx, bi, lvl0, crop, channels_last=True)
        mean_cl, med_cl = time_op(lambda: _roi_align_impl.pyramid_forward([im_cl], bx, bi, lvl0, crop, channels_last=True))
        print(json.dumps({"case": tag, "map": dt, "kernel": "channels_last", "avg_us": round(mean_cl, 2), "median_us": round(med_cl, 2),
                          "alg_MB": round(byts / 1e6, 2), "frac_of_8TBps": round(byts / (mean_cl * 1e-6) / 8e12, 4),
                          "out_only_frac": round(4.0 * out.numel() / (mean_cl * 1e-6) / 8e12, 4), "bit_equal_to_row_major_kernel": bool(torch.equal(out, out_cl))}), flush=True)
        print(json.dumps({"case": tag, "map": dt, "kernel": kern, "avg_us": round(mean, 2), "median_us": round(med, 2), "out_MB": round(4e-6 * out.numel(), 2),
                          "touched_input_MB": round(esz * C * tv / 1e6, 2), "alg_MB": round(byts / 1e6, 2),
                          "GBps": round(byts / mean / 1e3, 1), "frac_of_8TBps": round(byts / (mean * 1e-6) / 8e12, 4),
                          "out_only_frac": round(4.0 * out.numel() / (mean * 1e-6) / 8e12, 4), "checksum": float(out.double().sum())}), flush=True)
