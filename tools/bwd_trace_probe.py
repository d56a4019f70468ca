import os, sys, json, ctypes
os.environ["MDT_BWD_TUNE"] = "1"
sys.path.insert(0, "/root/repo")
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
_lib.lib()
ts = torch.zeros(64 + 4 * 8192, dtype=torch.int64, device=dev)
_lib.ab_lib().mdt_debug_bwd_timestamps(ctypes.c_void_p(ts.data_ptr()))
os.environ["MDT_BWD_DBG"] = "2"
nscat = 288
for nt, parts in (("512", "0"),):
    os.environ["MDT_BWD_THREADS"] = nt; os.environ["MDT_BWD_ZERO_WGS"] = parts
    ind_bal = torch.arange(N, dtype=torch.int32, device=dev) % B
    for name, (bx, ind) in {"train": (boxes_train, ind_train), "balanced": (boxes, ind_bal)}.items():
        for _ in range(3):
            ts.zero_()
            _roi_align_impl.crop_backward(g, bx, ind, shape, mode="territory")
            torch.cuda.synchronize()
        t = ts.cpu().numpy()[64:].reshape(-1, 4)
        nwg = int((t[:, 0] != 0).sum())
        t = t[:nwg]
        t0 = t[:, 0].min()
        st = (t[:, 0] - t0) * 0.01; en = (t[:, 1] - t0) * 0.01
        cu = (t[:, 2] >> 8) & 0xF; se = (t[:, 2] >> 13) & 0x7; sh = (t[:, 2] >> 12) & 1; xcc = t[:, 3] & 0xF
        key = xcc * 10000 + se * 100 + sh * 50 + cu
        sk = key[:nscat]
        uniq, counts = np.unique(sk, return_counts=True)
        rec = {"nt": nt, "parts": parts, "case": name, "kernel_span_us": round(float(en.max()), 2),
               "scatter_start_max": round(float(st[:nscat].max()), 2), "scatter_end_p50": round(float(np.median(en[:nscat])), 2), "scatter_end_max": round(float(en[:nscat].max()), 2),
               "scatter_dur_p50": round(float(np.median(en[:nscat] - st[:nscat])), 2), "scatter_dur_max": round(float((en[:nscat] - st[:nscat]).max()), 2),
               "zero_start_p50": round(float(np.median(st[nscat:])), 2), "zero_start_max": round(float(st[nscat:].max()), 2),
               "zero_dur_p50": round(float(np.median(en[nscat:] - st[nscat:])), 2), "zero_dur_max": round(float((en[nscat:] - st[nscat:]).max()), 2),
               "zero_end_p50": round(float(np.median(en[nscat:])), 2), "zero_end_max": round(float(en[nscat:].max()), 2),
               "distinct_cus_scatter": int(len(uniq)), "max_scatter_per_cu": int(counts.max()), "distinct_cus_all": int(len(np.unique(key)))}
        print(json.dumps(rec), flush=True)
        hist, edges = np.histogram(st[nscat:], bins=[0, 1, 2, 5, 10, 15, 20, 25, 30, 40, 60, 100])
        print(json.dumps({"zero_start_hist": hist.tolist(), "edges": edges.tolist()}), flush=True)
