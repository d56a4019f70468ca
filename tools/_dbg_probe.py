import os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from medicaldetectiontoolkit_amd.cuda_functions import _roi_align_impl
from tools.bwd_tune import timeit
from tests.helpers import random_boxes_3d
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
os.environ["MDT_BWD_SSPLIT"] = "1"
for zero_off in (1, 0):
    for stop in (1, 2, 3, 0):
        os.environ["MDT_BWD_DBG"] = str(zero_off | (stop << 4))
        rec = {"zero_role_off": zero_off, "stop_after": {1: "bitmap", 2: "tables", 3: "columns", 0: "all"}[stop]}
        rec["rand_us"] = timeit(lambda: _roi_align_impl.crop_backward(g, boxes, ind_rand, shape), 40)[0]
        rec["train_us"] = timeit(lambda: _roi_align_impl.crop_backward(g, boxes_train, ind_train, shape), 40)[0]
        print(json.dumps(rec), flush=True)
