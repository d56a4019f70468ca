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


def batch_with_gt_from_proposals(net, cf, batch, device, per_element=2):
    """A batch whose GT boxes are `per_element` large, mutually disjoint PROPOSALS of the net's own RPN on the batch's image (rounded
    outwards to integers) -- the construction of tests/golden/make_step_golden.py -- so that detection_target_layer finds positive RoIs
    and every train_rois_per_image slot of the RoI heads is valid (random-init weights on random GT leave most slots empty).  GT masks
    are the solid boxes.  Device-resident like to_device()'s result."""
    import torch
    dim = cf.dim
    img = batch["data"] if torch.is_tensor(batch["data"]) else torch.from_numpy(np.ascontiguousarray(batch["data"])).to(device)
    with torch.no_grad():
        net.forward(img.float(), is_training=True, with_masks=False)
    props = net.rpn_rois_batch_info.detach().cpu().numpy()
    scale = np.asarray(cf.scale, dtype=np.float64)
    B = int(img.shape[0])
    shape = tuple(int(v) for v in img.shape[2:])
    lo_cols, hi_cols = ([0, 1, 4], [2, 3, 5]) if dim == 3 else ([0, 1], [2, 3])
    bb, labels, masks = [], [], []
    for b in range(B):
        pb = props[props[:, -1] == b][:, :2 * dim] * scale
        pb[:, lo_cols] = np.floor(pb[:, lo_cols])
        pb[:, hi_cols] = np.ceil(pb[:, hi_cols])
        pb = np.clip(pb, 0, scale)
        ext = pb[:, hi_cols] - pb[:, lo_cols]
        ok = np.nonzero((ext >= np.array([4, 4, 2][:dim])).all(1))[0]
        order = ok[np.argsort(-ext[ok].prod(1), kind="stable")]
        chosen = []
        for i in order:
            if all(any(pb[i, h] <= pb[j, l] or pb[i, l] >= pb[j, h] for l, h in zip(lo_cols, hi_cols)) for j in chosen):
                chosen.append(i)
            if len(chosen) == per_element:
                break
        boxes = pb[chosen].astype(np.float32).reshape(-1, 2 * dim)
        ms = np.zeros((len(chosen), 1) + shape, dtype=np.uint8)
        for k, bx in enumerate(boxes.astype(np.int64)):
            sl = (slice(bx[0], bx[2]), slice(bx[1], bx[3])) + ((slice(bx[4], bx[5]),) if dim == 3 else ())
            ms[k, 0][sl] = 1
        bb.append(boxes)
        labels.append(np.array([1 + (k % 2) for k in range(len(chosen))], dtype=np.int64))
        masks.append(ms)
    out = {"data": img, "seg": None, "pid": ["gt_from_proposals_%d" % b for b in range(B)], "bb_target": bb, "roi_labels": labels,
           "roi_masks": masks, "class_target": [[int(v) - 1 for v in l] for l in labels]}
    stack = [torch.from_numpy(m) for m in masks if len(m) > 0]
    out["roi_masks_device"] = torch.cat(stack, 0).to(device) if stack else None
    return out


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


# ---------------------------------------------------------------------------------------------------------------------------------
# Synthetic box / RoI workloads of the measurement path (bench.py, tools/microbench*.py, tools/profile_case.py) and of the tests
# (tests/helpers.py re-exports them): SURVEY.md section 8(d).


def random_boxes_3d(rng, n, patch=128.0, xy=(8.0, 64.0), z=(2.0, 16.0), spill=False):
    """normalised (y1,x1,y2,x2,z1,z2); centre U(0,1), size log-uniform; clipped to [0,1] unless spill."""
    c = rng.uniform(0, 1, size=(n, 3))
    sxy = np.exp(rng.uniform(np.log(xy[0]), np.log(xy[1]), size=(n, 2))) / patch
    sz = np.exp(rng.uniform(np.log(z[0]), np.log(z[1]), size=(n, 1))) / patch
    half = np.concatenate([sxy, sz], 1) / 2
    lo, hi = c - half, c + half
    if not spill:
        lo, hi = np.clip(lo, 0, 1), np.clip(hi, 0, 1)
    b = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1)
    return b.astype(np.float32)


def random_boxes_2d(rng, n, patch=288.0, size=(8.0, 128.0), spill=False):
    c = rng.uniform(0, 1, size=(n, 2))
    s = np.exp(rng.uniform(np.log(size[0]), np.log(size[1]), size=(n, 2))) / patch
    lo, hi = c - s / 2, c + s / 2
    if not spill:
        lo, hi = np.clip(lo, 0, 1), np.clip(hi, 0, 1)
    return np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1]], 1).astype(np.float32)


def nms_boxes(rng, n, dim=3, patch=128.0, tie_free=True):
    """pixel-coordinate detections [n, 2*dim+1] clustered around a few centres, tie-free scores."""
    k = max(1, n // 40)
    centres = rng.uniform(0.1 * patch, 0.9 * patch, size=(k, dim))
    which = rng.integers(0, k, size=n)
    c = centres[which] + rng.normal(0, 3.0, size=(n, dim))
    s = np.exp(rng.uniform(np.log(4), np.log(32), size=(n, dim)))
    lo = np.clip(c - s / 2, 0, patch)
    hi = np.clip(c + s / 2, 0, patch)
    if dim == 3:
        b = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1)
    else:
        b = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1]], 1)
    scores = rng.permutation(np.linspace(0.0, 1.0, n)) if tie_free else np.round(rng.uniform(0, 1, n), 1)
    return np.concatenate([b, scores[:, None]], 1).astype(np.float32)


def trainlike_rois_3d(rng, batch, per_element=6, side=8.0, patch=128.0):
    """RoIs as a training step hands them to one pyramid level (SURVEY.md 8(d) "train-realistic", forced onto one
    level): `per_element` sampled RoIs per batch element (train_rois_per_image, lidc configs.py:258) scattered around
    one object per element, box sides 0.75..1.4 x `side` px -- the sizes the level rule of mrcnn.py:403 routes to
    the level whose anchor scale is `side` (8 px = P2).  Returns normalised boxes [batch*per_element, 6] f32 and
    box_ind [batch*per_element] i32."""
    ctr = rng.uniform(0.25, 0.75, size=(batch, 3))
    rows = []
    for b in range(batch):
        for _ in range(per_element):
            c = ctr[b] + rng.normal(0, 0.02, size=3)
            s = rng.uniform(0.75 * side, 1.4 * side, size=3) / patch
            rows.append([c[0] - s[0] / 2, c[1] - s[1] / 2, c[0] + s[0] / 2, c[1] + s[1] / 2, c[2] - s[2] / 2, c[2] + s[2] / 2])
    boxes = np.clip(np.asarray(rows), 0.0, 1.0).astype(np.float32)
    box_ind = (np.arange(batch * per_element) // per_element).astype(np.int32)
    return boxes, box_ind
