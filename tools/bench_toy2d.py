"""BASELINE config 1 on the GPU: the toy experiment's 2D Retina Net (experiments/toy_exp: 320x320 images, batch 20,
one circle / donut per image; BASELINE asks for 64x64 shapes as the plumbing case) trained for a few steps through the 2D
kernels (2D NMS, anchor matching, decode, SHEM losses).  Prints one JSON line per shape: images/s and the loss
trajectory.  The reference runs this config on the CPU; the product has no CPU path by design."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from medicaldetectiontoolkit_amd import training  # noqa: E402
from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402
from medicaldetectiontoolkit_amd.models import retina_unet  # noqa: E402


def toy_batch(size, batch, seed):
    """experiments/toy_exp/generate_toys.py:30-54: U(0,1) noise image + 0.2 inside a circle (class 1, radius 20 at 320 px)
    or a donut (class 2); one object per image; batch dict in the reference's format"""
    rng = np.random.default_rng(seed)
    H = W = size
    data = rng.uniform(0, 1, size=(batch, 1, H, W)).astype(np.float32)
    seg = np.zeros((batch, 1, H, W), dtype=np.uint8)
    bb, labels, masks = [], [], []
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    r = max(5, int(round(20 * size / 320.0)))
    for b in range(batch):
        cls = int(rng.integers(1, 3))
        cy, cx = rng.integers(r + 2, H - r - 2), rng.integers(r + 2, W - r - 2)
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        m = d2 <= r * r
        if cls == 2:
            m &= d2 >= (r // 2) ** 2
        data[b, 0][m] += 0.2
        seg[b, 0][m] = 1
        bb.append(np.array([[cy - r, cx - r, cy + r + 1, cx + r + 1]], dtype=np.float32))
        labels.append(np.array([cls], dtype=np.int64))
        masks.append(m[None, None].astype(np.uint8))
    return {"data": data, "seg": seg, "bb_target": bb, "roi_labels": labels, "roi_masks": masks, "pid": list(range(batch))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--sizes", type=str, default="64,320", help="image sizes to run (BASELINE config 1 names 64x64)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    for size, batch in [(int(v), 20) for v in args.sizes.split(",")]:
        cf = Configs(dim=2, model="retina_net", patch_size=[size, size], batch_size=batch, rpn_train_anchors_per_image=2,
                     train_rois_per_image=2, class_dict={1: "circle", 2: "donut"})
        torch.manual_seed(0)
        net = retina_unet.net(cf, device=dev)
        opt = training.build_optimizer(net, cf)
        batches = [toy_batch(size, batch, s) for s in range(8)]
        losses = []
        for i in range(5):
            training.train_step(net, opt, batches[i % 8], monitor=False)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(args.steps):
            r = training.train_step(net, opt, batches[i % 8], monitor=False)
            losses.append(r["torch_loss"].detach())
        torch.cuda.synchronize()
        dt = time.time() - t0
        losses = [float(v) for v in losses]
        res = net.test_forward(batches[0])
        n_det = sum(1 for b in res["boxes"] for d in b if d["box_type"] == "det")
        print(json.dumps({"config": "toy_exp 2D Retina Net, %dx%d, batch %d (BASELINE config 1 shapes, run on MI355X)" % (size, size, batch),
                          "metric": "2D images/sec (train), toy_exp Retina Net", "value": round(batch * args.steps / dt, 1), "unit": "images/s",
                          "images_per_s": round(batch * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 2), "steps": args.steps,
                          "loss_first5_mean": round(float(np.mean(losses[:5])), 4), "loss_last5_mean": round(float(np.mean(losses[-5:])), 4),
                          "test_forward_detections": n_det, "finite": bool(np.isfinite(losses).all())}), flush=True)


if __name__ == "__main__":
    main()
