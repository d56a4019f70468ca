"""Timings of the remaining hot-path ops on the GPU box (SURVEY 8(d) secondary metrics): 2D RoIAlign / NMS,
anchor generation, anchor matching, decode, WBC, 2D->3D merge.  One JSON line per case."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import predictor
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.cuda_functions import _nms_impl, _roi_align_impl
from medicaldetectiontoolkit_amd.utils import model_utils as mutils
from medicaldetectiontoolkit_amd.utils.synthetic_data import nms_boxes, random_boxes_2d

dev = torch.device("cuda:0"); rng = np.random.default_rng(0)


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return round(t[len(t) // 2], 2)


def rep(name, us, **kw):
    print(json.dumps(dict(case=name, median_us=us, **kw)), flush=True)


# 2D RoIAlign at the LIDC 2D shapes (C=192, P2 72x72 at 288^2, batch 20)
B, C, H, W = 20, 192, 72, 72
img = torch.randn(B, C, H, W, device=dev)
for N, crop in ((120, (7, 7)), (120, (14, 14)), (2500, (7, 7))):
    boxes = torch.from_numpy(random_boxes_2d(rng, N)).to(dev)
    ind = torch.from_numpy(rng.integers(0, B, size=N).astype(np.int32)).to(dev)
    g = torch.randn((N, C) + crop, device=dev)
    rep("roialign2d_fwd_N%d_%dx%d" % (N, crop[0], crop[1]), timeit(lambda: _roi_align_impl.crop_forward(img, boxes, ind, crop)))
    by = 4 * B * C * H * W + 4 * N * C * crop[0] * crop[1] + 20 * N
    us = timeit(lambda: _roi_align_impl.crop_backward(g, boxes, ind, (B, C, H, W)))
    rep("roialign2d_bwd_N%d_%dx%d" % (N, crop[0], crop[1]), us, alg_MB=round(by / 1e6, 2), GBps=round(by / us / 1e3, 1))
# 2D NMS
for n, thr in ((3000, 0.7), (10000, 1e-5)):
    d = nms_boxes(rng, n, dim=2, patch=288.0)
    ds = torch.from_numpy(d[np.argsort(-d[:, -1].astype(np.float64), kind="stable")]).to(dev)
    rep("nms2d_full_N%d_t%g" % (n, thr), timeit(lambda: _nms_impl.nms_sorted(ds, thr, 2)))
    rep("nms2d_keep500_N%d_t%g" % (n, thr), timeit(lambda: _nms_impl.nms_sorted(ds, thr, 2, max_keep=500)))
# anchors + matching
for model in ("mrcnn", "retina_unet"):
    cf = Configs(dim=3, model=model)
    rep("generate_pyramid_anchors_3d_%s" % model, timeit(lambda: mutils.generate_pyramid_anchors(None, cf, device=dev)))
    anchors = mutils.generate_pyramid_anchors(None, cf, device=dev)
    for G in (1, 3, 8):
        c = rng.uniform(30, 100, size=(G, 3)); s = rng.uniform(8, 30, size=(G, 3))
        gt = torch.from_numpy(np.stack([c[:, 0] - s[:, 0], c[:, 1] - s[:, 1], c[:, 0] + s[:, 0], c[:, 1] + s[:, 1], c[:, 2] - s[:, 2] / 2, c[:, 2] + s[:, 2] / 2], 1)).to(dev)
        cls = torch.from_numpy(rng.integers(1, 3, size=G).astype(np.int32)).to(dev)
        A = anchors.shape[0]
        rep("anchor_match_A%d_G%d" % (A, G), timeit(lambda: mutils.anchor_match_labels(anchors, gt, cls, 0.01, cf.anchor_matching_iou)),
            alg_MB=round((48 * A + 48 * G + 16 * A) / 1e6, 2))
# decode
cf = Configs(dim=3, model="mrcnn")
anchors32 = mutils.generate_pyramid_anchors(None, cf, device=dev, return_f32=True)[1]
deltas = torch.randn(anchors32.shape[0], 6, device=dev); order = torch.randperm(anchors32.shape[0], device=dev)[:6000]; sc = torch.rand(6000, device=dev)
rep("decode_clip_gather_6000", timeit(lambda: mutils.decode_clip_boxes(anchors32, deltas, cf.rpn_bbox_std_dev, cf.window, order=order, scores=sc)))
# WBC
for n, n_true in ((2000, 20), (45000, 20)):
    true = rng.uniform(40, 400, size=(n_true, 3)); which = rng.integers(0, n_true, size=n)
    c = true[which] + rng.normal(0, 2.0, size=(n, 3)); s = rng.uniform(6, 20, size=(n, 3))
    dets = np.concatenate([np.stack([c[:, 0] - s[:, 0], c[:, 1] - s[:, 1], c[:, 0] + s[:, 0], c[:, 1] + s[:, 1], c[:, 2] - s[:, 2], c[:, 2] + s[:, 2]], 1),
                           rng.permutation(np.linspace(0.02, 0.99, n))[:, None], rng.uniform(0.2, 1, (n, 1)), rng.integers(1, 5, (n, 1)).astype(float)], 1)
    o = np.argsort(-dets[:, 6], kind="stable")
    d = torch.from_numpy(dets[o]).to(dev); p = torch.from_numpy(rng.integers(0, 1500, size=n).astype(np.int32)[o]).to(dev)
    t0 = time.time(); ks, kc = predictor.weighted_box_clustering_device(d, p, 1e-5, 20.0, 1500); torch.cuda.synchronize()
    reps = []
    for _ in range(5):
        t0 = time.time(); ks, kc = predictor.weighted_box_clustering_device(d, p, 1e-5, 20.0, 1500); torch.cuda.synchronize(); reps.append((time.time() - t0) * 1e6)
    rep("wbc3d_n%d" % n, round(sorted(reps)[2], 1), clusters=int(ks.numel()))
# 2D -> 3D merge
n = 5000
ctr = rng.uniform(30, 220, size=(60, 2)); zc = rng.integers(5, 120, size=60); which = rng.integers(0, 60, size=n)
c = ctr[which] + rng.normal(0, 1.5, size=(n, 2)); sz = rng.uniform(8, 24, size=(n, 2)); sl = np.clip(zc[which] + np.round(rng.normal(0, 3, size=n)).astype(int), 0, 127)
dets = np.stack([c[:, 0] - sz[:, 0] / 2, c[:, 1] - sz[:, 1] / 2, c[:, 0] + sz[:, 0] / 2, c[:, 1] + sz[:, 1] / 2, rng.permutation(np.linspace(0.05, 0.99, n)), sl.astype(float)], 1)
reps = []
for _ in range(4):
    t0 = time.time(); k, kz = predictor.nms_2to3D(dets, 0.1, device=dev); reps.append((time.time() - t0) * 1e6)
rep("nms_2to3D_n%d (incl. H2D/D2H)" % n, round(sorted(reps)[1], 1), cubes=len(k))
