import os, sys, json, ctypes
os.environ["MDT_BWD_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from medicaldetectiontoolkit_amd import _lib
from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl
from medicaldetectiontoolkit_amd.utils.synthetic_data import random_boxes_3d
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B, C = 8, 36
shape = (B, C, 32, 32, 128)
N = 48
boxes = torch.from_numpy(random_boxes_3d(rng, N)).to(dev)
ind_rand = torch.from_numpy(rng.integers(0, B, size=N).astype(np.int32)).to(dev)
ctr = rng.uniform(0.2, 0.8, size=(B, 3)); tb = []
for b in range(B):
    for k in range(6):
        c = ctr[b] + rng.normal(0, 0.02, size=3); s = rng.uniform(6, 11, size=3) / 128.0
        tb.append([c[0]-s[0]/2, c[1]-s[1]/2, c[0]+s[0]/2, c[1]+s[1]/2, c[2]-s[2]/2, c[2]+s[2]/2])
boxes_train = torch.tensor(tb, dtype=torch.float32, device=dev)
ind_train = torch.arange(N, dtype=torch.int32, device=dev) // 6
g = torch.randn((N, C, 14, 14, 5), device=dev)
L = _lib.lib()
ts = torch.zeros(32, dtype=torch.int64, device=dev)
L._handle if False else None
_lib.ab_lib().mdt_debug_bwd_timestamps(ctypes.c_void_p(ts.data_ptr()))
names = ["start", "bitmap", "prefix", "tab(a)", "compact(b)", "offsets", "passes(c)", "combine(d)"]
for nt in ("512",):
    os.environ["MDT_BWD_THREADS"] = nt
    for zero_off in (0,):
        os.environ["MDT_BWD_DBG"] = str(zero_off)
        ind_bal = torch.arange(N, dtype=torch.int32, device=dev) % B
        for name, (bx, ind) in {"random": (boxes, ind_rand), "balanced": (boxes, ind_bal), "train": (boxes_train, ind_train)}.items():
            for wg in (0, 150):
                os.environ["MDT_BWD_DBG_WG"] = str(wg)
                for _ in range(3):
                    ts.zero_()
                    _roi_align_impl.crop_backward(g, bx, ind, shape, mode="territory")
                    torch.cuda.synchronize()
                t = ts.cpu().numpy()
                d = {names[i]: round(float(t[i] - t[i - 1]) * 0.01, 2) for i in range(1, 8)}
                print(json.dumps({"nt": nt, "case": name, "zero_off": zero_off, "wg": wg, "stage_us": d, "total_us": round(float(t[7] - t[0]) * 0.01, 2), "d_first_iter": {"rank": round(float(t[8]-t[6])*0.01,2), "decode+bbq": round(float(t[9]-t[8])*0.01,2), "jloop+store": round(float(t[10]-t[9])*0.01,2)}, "MHz": round(float(t[23] - t[16]) / (float(t[7] - t[0]) * 0.01), 1)}), flush=True)
