"""Writes tests/golden/step_reference.npz: loss terms and per-module gradient norms of ONE assembled training step of the
REFERENCE's own models, run UNMODIFIED on the CPU:
  /root/reference/models/mrcnn.py:801-1082       net.train_forward  (forward :987, loss_samples_forward :1052, five loss terms :946)
  /root/reference/models/retina_unet.py:338-513  net.train_forward  (K = 3 class logits, compute_class_loss :126, batch_dice seg loss)
with the four `cuda_functions.*` modules replaced by the CPU oracle (RoIAlign as an autograd Function over the oracle's
forward / backward, so gradients flow through it as through the reference extension), `Tensor.cuda()` the identity, and
torch-0.4 semantics the code relies on restored (integer `/` on index tensors).  The only values changed are DEFAULT
ARGUMENTS / config entries that make the step deterministic (tests/golden/step_inputs.py): the default shem_poolsize of
retina_unet.compute_class_loss (20 -> 1; the call site does not pass it, retina_unet.py:432).

Two passes for Mask R-CNN: (1) `forward` only, to read the proposals the name-seeded weights produce; two of them per
batch element become the GT boxes (rounded to integers) so that detection_target_layer finds positive RoIs; (2) the step.
Run once in the build container:  timeout 1800 python tests/golden/make_step_golden.py [small|large]
`large` (round 4) = patch 128 x 128 x 64, batch 1 -> step_reference_large.npz: the C2 map has 65 536 voxels, the size from which this
repo's fp32-MFMA convolution kernels are dispatched, so the assembled-step parity covers them (tests/test_step_parity_gpu.py asserts
the dispatch)."""
import importlib.util
import logging
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from tests.golden import step_inputs as si  # noqa: E402


def nms_gpu(dets, thresh):
    keep = oracle.gpu_nms(dets.detach().numpy().astype(np.float32), float(thresh), True)
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


class _OracleCrop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, boxes, box_ind, crop):
        img = image.detach().numpy().astype(np.float32)
        squeezed = 0
        while img.ndim > len(crop) + 2 and img.shape[-1] == 1:
            img = img[..., 0]
            squeezed += 1
        bx = np.ascontiguousarray(boxes.detach().numpy().astype(np.float32))
        bi = np.ascontiguousarray(box_ind.detach().numpy().astype(np.int32))
        ctx.meta = (bx, bi, img.shape, tuple(image.shape))
        return torch.from_numpy(oracle.crop_and_resize_forward(np.ascontiguousarray(img), bx, bi, crop))

    @staticmethod
    def backward(ctx, g):
        bx, bi, shp, orig = ctx.meta
        gi = oracle.crop_and_resize_backward(np.ascontiguousarray(g.detach().numpy().astype(np.float32)), bx, bi, shp)
        return torch.from_numpy(gi).reshape(orig), None, None, None


class CropAndResizeFunction(object):
    """crop_and_resize.CropAndResizeFunction (roi_align_3D/roi_align/crop_and_resize.py:10-51)"""

    def __init__(self, *args):
        self.crop = tuple(int(a) for a in args[:-1])

    def __call__(self, image, boxes, box_ind):
        return _OracleCrop.apply(image, boxes, box_ind, self.crop)


for name in ["cuda_functions", "cuda_functions.nms_2D", "cuda_functions.nms_2D.pth_nms", "cuda_functions.nms_3D",
             "cuda_functions.nms_3D.pth_nms", "cuda_functions.roi_align_2D", "cuda_functions.roi_align_2D.roi_align",
             "cuda_functions.roi_align_2D.roi_align.crop_and_resize", "cuda_functions.roi_align_3D",
             "cuda_functions.roi_align_3D.roi_align", "cuda_functions.roi_align_3D.roi_align.crop_and_resize"]:
    m = types.ModuleType(name)
    m.nms_gpu = nms_gpu
    m.CropAndResizeFunction = CropAndResizeFunction
    sys.modules[name] = m
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


mu = load("utils/model_utils.py", "ref_mu")
sys.modules["utils.model_utils"] = mu
mr = load("models/mrcnn.py", "ref_mrcnn")
ru = load("models/retina_unet.py", "ref_retina")


class torch04(object):
    """torch 0.4 behaviours the reference relies on: `long_tensor / int` floor-divides (retina_unet.py:212);
    torch.LongTensor(uint8 ndarray) converts (retina_unet.py:396)"""

    def __enter__(self):
        self._div = torch.Tensor.__truediv__

        def div(a, b):
            if not a.is_floating_point() and not (torch.is_tensor(b) and b.is_floating_point()) and not isinstance(b, float):
                return torch.div(a, b, rounding_mode="floor")
            return self._div(a, b)
        torch.Tensor.__truediv__ = div

    def __exit__(self, *exc):
        torch.Tensor.__truediv__ = self._div


class Recorder(object):
    """observes (does not change) the value a reference loss helper returns: the step only exposes the total and one term"""

    def __init__(self, mod, name):
        self.fn, self.vals = getattr(mod, name), []
        setattr(mod, name, self)

    def __call__(self, *a, **k):
        r = self.fn(*a, **k)
        v = r[0] if isinstance(r, tuple) else r
        self.vals.append(float(v.detach().double().sum()))
        return r


def grad_norms(net):
    mods = {}
    for name, p in net.named_parameters():
        g = p.grad
        mods.setdefault(si.module_of(name), []).append(0.0 if g is None else float((g.double() ** 2).sum()))
    return {k: float(np.sqrt(sum(v))) for k, v in mods.items()}


def retina_bench():
    """round 6: BASELINE config 2 AT ITS BENCHMARKED SIZE -- the reference's retina_unet.py `net.train_forward` + backward at
    128^3, batch 8, on the CPU -> step_reference_bench_retina.npz.  The batch is the one of the Mask R-CNN `bench` case (GT boxes read from
    step_reference_bench.npz, so the ~15 min Mask R-CNN step is not repeated).
    Run:  (ulimit -v 56000000; timeout 7200 python tests/golden/make_step_golden.py bench-retina)"""
    log = logging.getLogger("step_golden")
    log.addHandler(logging.NullHandler())
    case = "bench"
    nb = si.CASES[case][1]
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    g = np.load(os.path.join(HERE, "step_reference_bench.npz"), allow_pickle=False)
    out = {}
    gt_boxes = [g["gt_boxes_%d" % b] for b in range(nb)]
    gt_labels = [g["gt_labels_%d" % b] for b in range(nb)]
    for b in range(nb):
        out["gt_boxes_%d" % b], out["gt_labels_%d" % b] = gt_boxes[b], gt_labels[b]
    batch = si.make_batch(si.make_image(case=case), gt_boxes, gt_labels)
    cfr = si.make_cf("retina_unet", case)
    cfr.backbone_path = os.path.join(REF, "models/backbone.py")
    ru.compute_class_loss.__defaults__ = (1,)                   # shem_poolsize default 20 -> 1 (deterministic SHEM)
    netr = ru.net(cfr, log)
    si.fill_by_name(netr)
    recr = {"class": Recorder(ru, "compute_class_loss"), "bbox": Recorder(ru, "compute_bbox_loss")}
    dice = Recorder(ru.mutils, "batch_dice")
    np.random.seed(0)
    torch.manual_seed(0)
    with torch04():
        resr = netr.train_forward(batch)
    for k, r in recr.items():
        out["retina_term_" + k] = np.float64(sum(r.vals) / nb)
    out["retina_term_seg_dice"] = np.float64(1.0 - dice.vals[0])
    out["retina_term_seg_ce"] = np.float64(2.0 * (resr["torch_loss"].item() - out["retina_term_class"] - out["retina_term_bbox"]) - out["retina_term_seg_dice"])
    netr.zero_grad()
    resr["torch_loss"].backward()
    out["retina_logger_string"] = np.array(resr["logger_string"])
    out["retina_loss"] = np.float64(resr["torch_loss"].item())
    out["retina_class_loss"] = np.float64(resr["monitor_values"]["class_loss"])
    for k, v in grad_norms(netr).items():
        out["retina_gradnorm_" + k] = np.float64(v)
    out["retina_n_pos_neg_anchors"] = np.array([sum(1 for bl in resr["boxes"] for bx in bl if bx["box_type"] == "pos_anchor"),
                                                 sum(1 for bl in resr["boxes"] for bx in bl if bx["box_type"] == "neg_anchor")])
    print("retina[bench]:", resr["logger_string"], out["retina_n_pos_neg_anchors"])
    np.savez_compressed(os.path.join(HERE, "step_reference_bench_retina.npz"), **out)
    for k, v in out.items():
        print(k, np.asarray(v).shape, v if np.asarray(v).size < 8 else "")


def main(case="small"):
    if case == "bench-retina":
        return retina_bench()
    log = logging.getLogger("step_golden")
    log.addHandler(logging.NullHandler())
    out = {}
    img = si.make_image(case=case)
    nb = si.CASES[case][1]
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))

    # ------------------------------------------------------------------ Mask R-CNN
    cf = si.make_cf("mrcnn", case)
    cf.backbone_path = os.path.join(REF, "models/backbone.py")
    net = mr.net(cf, log)
    si.fill_by_name(net)
    # pass 1: proposals of the seeded weights -> GT boxes
    with torch.no_grad():
        net.forward(torch.from_numpy(img))
    props = net.rpn_rois_batch_info.numpy()                     # normalised (y1, x1, y2, x2, z1, z2, batch_ix)
    scale = np.asarray(cf.scale, dtype=np.float64)
    gt_boxes, gt_labels = [], []
    for b in range(nb):
        pb = props[props[:, -1] == b][:, :6] * scale
        lo_cols, hi_cols = [0, 1, 4], [2, 3, 5]
        pb[:, lo_cols] = np.floor(pb[:, lo_cols])               # outward rounding: the GT box contains its proposal
        pb[:, hi_cols] = np.ceil(pb[:, hi_cols])
        pb = np.clip(pb, 0, scale)
        ext = np.stack([pb[:, 2] - pb[:, 0], pb[:, 3] - pb[:, 1], pb[:, 5] - pb[:, 4]], 1)
        ok = np.nonzero((ext >= np.array([4, 4, 2])).all(1))[0]
        order = ok[np.argsort(-ext[ok].prod(1), kind="stable")]
        chosen = []
        for i in order:                                         # two large, mutually disjoint proposals
            if all(pb[i, 2] <= pb[j, 0] or pb[i, 0] >= pb[j, 2] or pb[i, 3] <= pb[j, 1] or pb[i, 1] >= pb[j, 3] or
                   pb[i, 5] <= pb[j, 4] or pb[i, 4] >= pb[j, 5] for j in chosen):
                chosen.append(i)
            if len(chosen) == 2:
                break
        gt_boxes.append(pb[chosen].astype(np.float32))
        gt_labels.append(np.array([1, 2][:len(chosen)] if b == 0 else [2, 1][:len(chosen)], dtype=np.int64))
    for b in range(nb):
        out["gt_boxes_%d" % b] = gt_boxes[b]
        out["gt_labels_%d" % b] = gt_labels[b]
    batch = si.make_batch(img, gt_boxes, gt_labels)
    # pass 2: the step
    rec = {k: Recorder(mr, "compute_" + k + "_loss") for k in ("rpn_class", "rpn_bbox", "mrcnn_class", "mrcnn_bbox", "mrcnn_mask")}
    np.random.seed(0)
    torch.manual_seed(0)
    with torch04():
        res = net.train_forward(batch)
    for k, r in rec.items():      # RPN terms: sum_b loss_b / B (mrcnn.py:911-912); head terms: one call each
        out["mrcnn_term_" + k] = np.float64(sum(r.vals) / (nb if k.startswith("rpn") else 1))
    net.zero_grad()
    res["torch_loss"].backward()
    ls = res["logger_string"]
    out["mrcnn_logger_string"] = np.array(ls)
    # the five terms are only exposed through the logger string at 2 decimals; recompute them exactly from the pieces the
    # step stores: run the same helpers on the stored tensors
    out["mrcnn_loss"] = np.float64(res["torch_loss"].item())
    out["mrcnn_class_loss"] = np.float64(res["monitor_values"]["class_loss"])
    for k, v in grad_norms(net).items():
        out["mrcnn_gradnorm_" + k] = np.float64(v)
    n_pos = sum(1 for bl in res["boxes"] for bx in bl if bx["box_type"] == "pos_class")
    n_neg = sum(1 for bl in res["boxes"] for bx in bl if bx["box_type"] == "neg_class")
    out["mrcnn_n_pos_neg_rois"] = np.array([n_pos, n_neg])
    out["mrcnn_n_pos_neg_anchors"] = np.array([sum(1 for bl in res["boxes"] for bx in bl if bx["box_type"] == "pos_anchor"),
                                                sum(1 for bl in res["boxes"] for bx in bl if bx["box_type"] == "neg_anchor")])
    print("mrcnn:", ls, out["mrcnn_n_pos_neg_rois"], out["mrcnn_n_pos_neg_anchors"])

    if case == "bench":       # the benchmarked configuration: Mask R-CNN only
        np.savez_compressed(os.path.join(HERE, "step_reference_%s.npz" % case), **out)
        return
    # ------------------------------------------------------------------ Retina U-Net (K = 3 class logits, dice + CE seg loss)
    cfr = si.make_cf("retina_unet", case)
    cfr.backbone_path = os.path.join(REF, "models/backbone.py")
    ru.compute_class_loss.__defaults__ = (1,)                   # shem_poolsize default 20 -> 1 (deterministic SHEM)
    netr = ru.net(cfr, log)
    si.fill_by_name(netr)
    recr = {"class": Recorder(ru, "compute_class_loss"), "bbox": Recorder(ru, "compute_bbox_loss")}
    dice = Recorder(ru.mutils, "batch_dice")
    np.random.seed(0)
    torch.manual_seed(0)
    with torch04():
        resr = netr.train_forward(batch)
    for k, r in recr.items():
        out["retina_term_" + k] = np.float64(sum(r.vals) / nb)
    out["retina_term_seg_dice"] = np.float64(1.0 - dice.vals[0])
    out["retina_term_seg_ce"] = np.float64(2.0 * (resr["torch_loss"].item() - out["retina_term_class"] - out["retina_term_bbox"]) - out["retina_term_seg_dice"])
    netr.zero_grad()
    resr["torch_loss"].backward()
    out["retina_logger_string"] = np.array(resr["logger_string"])
    out["retina_loss"] = np.float64(resr["torch_loss"].item())
    out["retina_class_loss"] = np.float64(resr["monitor_values"]["class_loss"])
    for k, v in grad_norms(netr).items():
        out["retina_gradnorm_" + k] = np.float64(v)
    out["retina_n_pos_neg_anchors"] = np.array([sum(1 for bl in resr["boxes"] for bx in bl if bx["box_type"] == "pos_anchor"),
                                                 sum(1 for bl in resr["boxes"] for bx in bl if bx["box_type"] == "neg_anchor")])
    print("retina:", resr["logger_string"], out["retina_n_pos_neg_anchors"])
    np.savez_compressed(os.path.join(HERE, "step_reference.npz" if case == "small" else "step_reference_%s.npz" % case), **out)
    for k, v in out.items():
        print(k, np.asarray(v).shape, v if np.asarray(v).size < 8 else "")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "small")
