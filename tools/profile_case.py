"""One case in a loop, for rocprofv3 --kernel-trace --stats.  Usage: profile_case.py <case> [iters]
cases: bwd_fast | bwd_twophase | bwd_ordered | fwd | fwd_cl | nms6000 | nms6000_keep75 | pmc_bwd
       | pyramid_bwd (all four levels in one launch, 48 RoIs routed 24/12/8/4)
env: MDT_N, MDT_CROP, MDT_ROIS=random|trainlike, MDT_INVALID=1, MDT_LEVEL=P2..P5,
     MDT_ROTATE=k (round 4: bwd_fast / pmc_bwd write k rotating output buffers -- 4 x 151 MB = 604 MB > the 256 MiB Infinity Cache -- instead of
     a fresh allocation that the caching allocator hands back as the SAME block every launch, i.e. cache-warm)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("MDT_BWD3_DBG"):       # role switches of the gather-form backward: the tuning build (include/mdt_hip_ab.h mdt_debug_bwd3),
    from medicaldetectiontoolkit_amd import _lib      # chosen before anything loads the product library
    _lib.use_tuning_build().mdt_debug_bwd3(None, int(os.environ["MDT_BWD3_DBG"]), 0)
from medicaldetectiontoolkit_amd.cuda_functions import _nms_impl, _roi_align_impl  # noqa: E402
from medicaldetectiontoolkit_amd.utils.synthetic_data import nms_boxes, random_boxes_3d, trainlike_rois_3d  # noqa: E402

case = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N = int(os.environ.get("MDT_N", 48))
crop = tuple(int(v) for v in os.environ.get("MDT_CROP", "14,14,5").split(","))
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
LEVELS = {"P2": (32, 32, 128), "P3": (16, 16, 64), "P4": (8, 8, 32), "P5": (4, 4, 16)}
shape = (8, 36) + LEVELS[os.environ.get("MDT_LEVEL", "P2")]
boxes = torch.from_numpy(random_boxes_3d(rng, N)).to(dev)
box_ind = torch.from_numpy(rng.integers(0, 8, size=N).astype(np.int32)).to(dev)
if os.environ.get("MDT_ROIS", "random") == "trainlike":   # 6 RoIs per element, P2-sized (utils/synthetic_data.trainlike_rois_3d)
    tb, ti = trainlike_rois_3d(rng, 8, N // 8, 8.0)
    boxes, box_ind = torch.from_numpy(tb).to(dev), torch.from_numpy(ti).to(dev)
if os.environ.get("MDT_INVALID"):        # all rows routed to other pyramid levels (what the random-init bench sees on P2)
    box_ind = torch.full_like(box_ind, -1)
g = torch.randn((N, 36) + crop, device=dev)
image = torch.randn(shape, device=dev)
image_cl = image.contiguous(memory_format=torch.channels_last_3d)      # the same map in the conv path's layout (mdt_pyramid_roi_align_forward_cl)
lvl0 = torch.zeros(N, dtype=torch.int32, device=dev)
dets = nms_boxes(rng, 6000)
ds = torch.from_numpy(dets[np.argsort(-dets[:, -1].astype(np.float64), kind="stable")]).to(dev)
n_rot = int(os.environ.get("MDT_ROTATE", "0"))
rot = [torch.empty(shape, device=dev) for _ in range(n_rot)]
state = {"k": 0}


def bwd_fast():
    if n_rot:
        state["k"] += 1
        return _roi_align_impl.crop_backward(g, boxes, box_ind, shape, out=rot[state["k"] % n_rot])
    return _roi_align_impl.crop_backward(g, boxes, box_ind, shape)


fns = {
    "bwd_fast": bwd_fast,
    "bwd_twophase": lambda: _roi_align_impl.crop_backward(g, boxes, box_ind, shape, mode="twophase"),
    "bwd_ordered": lambda: _roi_align_impl.crop_backward(g, boxes, box_ind, shape, mode="ordered"),
    "fwd": lambda: _roi_align_impl.crop_forward(image, boxes, box_ind, crop),
    "fwd_cl": lambda: _roi_align_impl.pyramid_forward([image_cl], boxes, box_ind, lvl0, crop, channels_last=True),
    "nms6000": lambda: _nms_impl.nms_sorted(ds, 0.7, 3),
    "nms6000_keep75": lambda: _nms_impl.nms_sorted(ds, 0.7, 3, max_keep=75),
}
if case == "pyramid_bwd":
    shapes = [(8, 36) + LEVELS[k] for k in ("P2", "P3", "P4", "P5")]
    per = []
    for li, (side, n) in enumerate(((8.0, 24), (16.0, 12), (32.0, 8), (64.0, 4))):
        tb, ti = trainlike_rois_3d(rng, 8, 6, side)
        keep = rng.permutation(len(tb))[:n]
        per.append((tb[keep], ti[keep], np.full(n, li, dtype=np.int32)))
    order = rng.permutation(48)
    pb = torch.from_numpy(np.concatenate([q[0] for q in per])[order]).to(dev)
    pi = torch.from_numpy(np.concatenate([q[1] for q in per])[order]).to(dev)
    pl = torch.from_numpy(np.concatenate([q[2] for q in per])[order]).to(dev)
    gp = torch.randn((48, 36) + crop, device=dev)
    fns["pyramid_bwd"] = lambda: _roi_align_impl.pyramid_backward(gp, pb, pi, pl, shapes)
if case == "pmc_bwd":
    # calibration dispatches (known byte count: plain 151 MB fill) followed by the op under test
    out = torch.empty(shape, device=dev)
    for i in range(iters):
        (rot[i % n_rot] if n_rot else out).zero_()
    torch.cuda.synchronize()
    for _ in range(iters):
        fns["bwd_fast"]()
    torch.cuda.synchronize()
else:
    for _ in range(iters):
        fns[case]()
    torch.cuda.synchronize()
