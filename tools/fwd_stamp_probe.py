"""Where the time of the channel-quad RoIAlign forward goes (csrc/roi_align_fwd.hip): per-workgroup wall-clock stamps (100 MHz) at the
workgroup's start, after its box / extents, when its first stage has landed in LDS, at its end.  usage: python tools/fwd_stamp_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from medicaldetectiontoolkit_amd import _lib
from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl
from medicaldetectiontoolkit_amd.utils.synthetic_data import random_boxes_3d

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
torch.manual_seed(0)
B, C = 8, 36
P2 = torch.randn((B, C, 32, 32, 128), device=dev)
L = _lib.use_tuning_build()      # libmdt_hip_tuning.so: the product sources + the stamp / role hooks (include/mdt_hip_ab.h)
for tag, n, crop in (("N240_14x14x5", 240, (14, 14, 5)), ("N600_7x7x3", 600, (7, 7, 3)), ("N48_14x14x5", 48, (14, 14, 5))):
    bx = torch.from_numpy(random_boxes_3d(rng, n)).to(dev)
    bi = torch.from_numpy(rng.integers(0, B, n).astype(np.int32)).to(dev)
    for _ in range(5):
        _roi_align_impl.crop_forward(P2, bx, bi, crop)
    torch.cuda.synchronize()
    buf = torch.zeros(4 * n * 9 + 64, dtype=torch.int64, device=dev)
    L.mdt_debug_fwd_stamps(buf.data_ptr())
    _roi_align_impl.crop_forward(P2, bx, bi, crop)
    torch.cuda.synchronize()
    L.mdt_debug_fwd_stamps(None)
    st = buf.cpu().numpy()[:4 * n * 9].reshape(-1, 4)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0
    dur = us[:, 3] - us[:, 0]
    sg = st[:, 2] > 0            # workgroups that took the staged path (the direct ones record only start and end)
    us_s = us[sg] if sg.any() else us[:1]
    rec = {"case": tag, "workgroups": int(len(st)), "kernel_span_us": round(float(us[:, 3].max()), 2),
           "wg_start_us_p50_p90_max": [round(float(np.percentile(us[:, 0], q)), 2) for q in (50, 90, 100)],
           "wg_duration_us_p10_p50_p90_max": [round(float(np.percentile(dur, q)), 2) for q in (10, 50, 90, 100)],
           "box_extents_us_p50": round(float(np.median(us_s[:, 1] - us_s[:, 0])), 2),
           "first_stage_landed_us_p50_p90": [round(float(np.percentile(us_s[:, 2] - us_s[:, 1], q)), 2) for q in (50, 90)],
           "stages_us_p50_p90": [round(float(np.percentile(us_s[:, 3] - us_s[:, 2], q)), 2) for q in (50, 90)],
           "staged_workgroups": int(sg.sum()), "direct_wg_duration_us_p50": round(float(np.median(dur[~sg])), 2) if (~sg).any() else None}
    print(json.dumps(rec), flush=True)
