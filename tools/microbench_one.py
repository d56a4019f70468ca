"""Event-timed RoIAlign-3D backward (default path) on P2 for MDT_N RoIs; MDT_INVALID=1 routes all rows off-level."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl
from medicaldetectiontoolkit_amd.utils.synthetic_data import random_boxes_3d
N = int(os.environ.get("MDT_N", 48)); dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
shape = (8, 36, 32, 32, 128)
boxes = torch.from_numpy(random_boxes_3d(rng, N)).to(dev)
ind = torch.from_numpy(rng.integers(0, 8, size=N).astype(np.int32)).to(dev)
if os.environ.get("MDT_INVALID"):
    ind = torch.full_like(ind, -1)
g = torch.randn((N, 36, 14, 14, 5), device=dev)
fn = lambda: _roi_align_impl.crop_backward(g, boxes, ind, shape)
for _ in range(10): fn()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
for a, b in ev:
    a.record(); fn(); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
print(json.dumps({"N": N, "invalid": bool(os.environ.get("MDT_INVALID")), "median_us": round(t[20], 2), "min_us": round(t[0], 2)}))
