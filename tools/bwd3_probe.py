"""Where the gather-form RoIAlign backward (csrc/roi_align_bwd_v3.hip) spends its time: wall-clock stamps of one scatter
workgroup (mdt_debug_bwd3) and event-timed launches with the roles switched off one at a time."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from medicaldetectiontoolkit_amd import _lib
from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl
from medicaldetectiontoolkit_amd.utils.synthetic_data import random_boxes_3d, trainlike_rois_3d

dev = torch.device("cuda:0")
L = _lib.use_tuning_build()      # libmdt_hip_tuning.so: the product sources + the stamp / role hooks (include/mdt_hip_ab.h)
B, C, N = 8, 36, 48
LEVELS = {"P2": (32, 32, 128), "P3": (16, 16, 64), "P5": (4, 4, 16)}
names = ["list", "plan", "wzero", "wtables", "dma_wait", "pass_y", "pass_x", "final", "zero_fill"]


def timeit(fn, iters=40):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return round(t[len(t) // 2], 2)


for lname, sp in LEVELS.items():
    shape = (B, C) + sp
    for rois in ("trainlike", "random"):
        rng = np.random.default_rng(0)
        if rois == "trainlike":
            bx, bi = trainlike_rois_3d(rng, B, 6, 8.0)
        else:
            bx, bi = random_boxes_3d(rng, N), rng.integers(0, B, size=N).astype(np.int32)
        bx, bi = torch.from_numpy(bx).to(dev), torch.from_numpy(bi).to(dev)
        g = torch.randn((N, C, 14, 14, 5), device=dev)
        fn = lambda: _roi_align_impl.crop_backward(g, bx, bi, shape)
        rec = {"level": lname, "rois": rois}
        for label, dbg in ():
            L.mdt_debug_bwd3(None, dbg, 0)
            rec["us_" + label] = timeit(fn)
        for wg in (0, 150):
            ts = torch.zeros(32, dtype=torch.int64, device=dev)
            L.mdt_debug_bwd3(ctypes.c_void_p(ts.data_ptr()), 0, wg)
            for _ in range(3):
                ts.zero_(); fn(); torch.cuda.synchronize()
            t = ts.cpu().numpy()
            rec["wg%d_stage_us" % wg] = {names[i]: round(float(t[i + 1] - t[i]) * 0.01, 2) for i in range(9)}
            rec["wg%d_total_us" % wg] = round(float(t[9] - t[0]) * 0.01, 2)
        L.mdt_debug_bwd3(None, 0, 0)
        print(json.dumps(rec), flush=True)
