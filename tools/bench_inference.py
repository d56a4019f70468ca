"""BASELINE config 5: patch-tiled 3D inference of one 512x512x256 volume (75 patches of 128^3, min overlap 30; with
--test-aug 1 the 4 mirrored passes of predictor.py:279-368 = 300 forwards), Mask R-CNN per chunk of 8 patches with the volume
and the detections resident on the device (predictor.collect_raw_boxes), boxes moved to patient coordinates, weighted box
clustering on the device.  Prints one JSON line (patients/min, patches/s, WBC ms).
usage: bench_inference.py [--amp bf16|none] [--test-aug 0|1]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import numpy as np, torch
from medicaldetectiontoolkit_amd import predictor
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn

ap = argparse.ArgumentParser()
ap.add_argument("--amp", default="bf16")
ap.add_argument("--volume", default="512,512,256")
ap.add_argument("--repeats", type=int, default=2)
ap.add_argument("--test-aug", type=int, default=0)
a = ap.parse_args()
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
cf = Configs(dim=3, model="mrcnn", batch_size=8, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev).eval()
rng = np.random.default_rng(0)
shape = tuple(int(v) for v in a.volume.split(","))
vol = rng.standard_normal((1,) + shape, dtype=np.float32)
amp = torch.bfloat16 if a.amp == "bf16" else None
aug = bool(a.test_aug)
n_ens = 4 if aug else 1
res = predictor.predict_patient(net, vol, cf, n_ens=n_ens, amp_dtype=amp, test_aug=aug)      # warm-up (MIOpen find)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.repeats):
    res = predictor.predict_patient(net, vol, cf, n_ens=n_ens, amp_dtype=amp, test_aug=aug)
torch.cuda.synchronize()
dt = (time.time() - t0) / a.repeats
# WBC alone on a worst-case-sized table (75 patches x 30 dets x 4 TTA x 5 epochs = 45 000 rows)
n = 45000
true = rng.uniform(40, 400, size=(20, 3))
which = rng.integers(0, 20, size=n)
c = true[which] + rng.normal(0, 2.0, size=(n, 3))
s = rng.uniform(6, 20, size=(n, 3))
dets = np.concatenate([np.stack([c[:, 0] - s[:, 0], c[:, 1] - s[:, 1], c[:, 0] + s[:, 0], c[:, 1] + s[:, 1], c[:, 2] - s[:, 2], c[:, 2] + s[:, 2]], 1),
                       rng.permutation(np.linspace(0.02, 0.99, n))[:, None], rng.uniform(0.2, 1, (n, 1)), rng.integers(1, 5, (n, 1)).astype(float)], 1)
order = np.argsort(-dets[:, 6], kind="stable")
d = torch.from_numpy(dets[order]).to(dev); p = torch.from_numpy(rng.integers(0, 1500, size=n).astype(np.int32)[order]).to(dev)
predictor.weighted_box_clustering_device(d, p, 1e-5, 20.0, 1500); torch.cuda.synchronize()
t1 = time.time(); ks, kc = predictor.weighted_box_clustering_device(d, p, 1e-5, 20.0, 1500); torch.cuda.synchronize(); wbc_ms = (time.time() - t1) * 1e3
print(json.dumps({"metric": "patch-tiled 3D inference, %s volume" % "x".join(map(str, shape)), "patients_per_min": round(60.0 / dt, 2),
                  "patches_per_s": round(res["n_patches"] * res["n_passes"] / dt, 1), "s_per_patient": round(dt, 3), "n_patches": res["n_patches"],
                  "n_passes": res["n_passes"], "forwards": res["n_patches"] * res["n_passes"],
                  "raw_boxes": res["n_raw_boxes"], "boxes_after_wbc": len(res["boxes"][0]), "amp": a.amp,
                  "wbc_45000_rows_ms": round(wbc_ms, 2), "wbc_clusters": int(ks.numel())}))
