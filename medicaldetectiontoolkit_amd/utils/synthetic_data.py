"""Synthetic batches in the reference's batch-dict format (Appendix B).

Volumes: N(0,1) noise plus 1-3 solid ellipsoids of intensity +1 per element (construction mirrors the
reference's toy generator, experiments/toy_exp/generate_toys.py:30-40).  'bb_target' / 'roi_masks' /
'roi_labels' follow what batchgenerators' ConvertSegToBoundingBoxCoordinates emits and the models consume
(models/mrcnn.py:864-868): bb_target[b] [n, 2*dim] pixel (y1,x1,y2,x2,(z1,z2)), roi_labels[b] [n] class ids >= 1,
roi_masks[b] [n, 1, Y, X, (Z)] uint8, seg [B, 1, ...] uint8.
"""
import numpy as np


def make_batch(patch_size, batch_size, seed=0, max_objects=3, radius=(4, 12), n_classes=2, with_empty=False):
    rng = np.random.default_rng(seed)
    dim = len(patch_size)
    shape = tuple(int(p) for p in patch_size)
    data = rng.standard_normal((batch_size, 1) + shape, dtype=np.float32)
    seg = np.zeros((batch_size, 1) + shape, dtype=np.uint8)
    grids = np.meshgrid(*[np.arange(s, dtype=np.float32) for s in shape], indexing="ij")
    bb_target, roi_labels, roi_masks, class_target = [], [], [], []
    for b in range(batch_size):
        n = int(rng.integers(1, max_objects + 1))
        if with_empty and b == batch_size - 1:
            n = 0
        boxes, labels, masks = [], [], []
        for _ in range(n):
            r = rng.uniform(radius[0], radius[1], size=dim)
            if dim == 3:
                r[2] = max(2.0, r[2] * 0.5)
            c = np.array([rng.uniform(r[i] + 1, shape[i] - r[i] - 1) for i in range(dim)])
            d2 = sum(((g - c[i]) / r[i]) ** 2 for i, g in enumerate(grids))
            m = d2 <= 1.0
            if not m.any():
                continue
            data[b, 0][m] += 1.0
            idx = np.nonzero(m)
            lo = [int(i.min()) for i in idx]
            hi = [int(i.max()) + 1 for i in idx]
            box = [lo[0], lo[1], hi[0], hi[1]] + ([lo[2], hi[2]] if dim == 3 else [])
            boxes.append(box)
            labels.append(int(rng.integers(1, n_classes + 1)))
            masks.append(m[None].astype(np.uint8))
            seg[b, 0][m] = 1
        bb_target.append(np.array(boxes, dtype=np.float32).reshape(-1, 2 * dim))
        roi_labels.append(np.array(labels, dtype=np.int64))
        roi_masks.append(np.array(masks, dtype=np.uint8).reshape((-1, 1) + shape))
        class_target.append([l - 1 for l in labels])
    return {"data": data, "seg": seg, "pid": ["synthetic_%d_%d" % (seed, b) for b in range(batch_size)],
            "class_target": class_target, "bb_target": bb_target, "roi_labels": roi_labels, "roi_masks": roi_masks}


def to_device(batch, device):
    """Upload the bulky entries of a batch dict once (image, stacked GT masks, seg); box lists stay host numpy.
    train_forward accepts the result unchanged ('data' may be a device tensor; 'roi_masks_device' = GT masks of
    all elements stacked [sum_G, 1, Y, X, (Z)] uint8)."""
    import torch
    out = dict(batch)
    out["data"] = torch.from_numpy(np.ascontiguousarray(batch["data"])).to(device)
    masks = [torch.from_numpy(np.ascontiguousarray(m)) for m in batch["roi_masks"] if len(m) > 0]
    out["roi_masks_device"] = torch.cat(masks, 0).to(device) if masks else None
    out["seg"] = torch.from_numpy(np.ascontiguousarray(batch["seg"])).to(device)
    return out
