"""Writes tests/golden/glue_reference.npz: outputs of the REFERENCE's own call-site glue
(/root/reference/models/mrcnn.py: proposal_layer :297, pyramid_roi_align :373, detection_target_layer :461,
refine_detections :620, the five loss helpers :176-286; models/retina_unet.py: refine_detections :194;
utils/model_utils.py: shem :674) on the seeded inputs of tests/golden/glue_inputs.py.

The reference functions are imported from /root/reference and run UNMODIFIED on the CPU:
  * its four `cuda_functions.*` modules are replaced by the CPU oracle (oracle/mdt_oracle.c restates the CUDA kernels
    and is itself pinned against the reference kernels compiled for gfx950, tests/test_hip_gpu.py);
  * `Tensor.cuda()` is made the identity;
  * torch-0.4 semantics the code relies on are restored for the duration of the calls: integer `/` on index tensors is
    floor division (retina_unet.py:212) -- nothing else had to be patched.
Run once in the build container:  timeout 900 python tests/golden/make_glue_golden.py
Only the outputs are stored; the GPU tests regenerate the inputs from the same seeds."""
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
from tests.golden import glue_inputs as gi  # noqa: E402


# ---------------------------------------------------------------- stand-ins for the reference's native extensions
def nms_gpu(dets, thresh):
    """pth_nms.nms_gpu (cuda_functions/nms_3D/pth_nms.py:5-17): indices into `dets`, best score first, GPU rule (>)"""
    keep = oracle.gpu_nms(dets.detach().numpy().astype(np.float32), float(thresh), True)
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


class CropAndResizeFunction(object):
    """crop_and_resize.CropAndResizeFunction (roi_align_3D/roi_align/crop_and_resize.py:10-34), forward only"""

    def __init__(self, *args):
        self.crop = tuple(int(a) for a in args[:-1])

    def __call__(self, image, boxes, box_ind):
        img = image.detach().numpy().astype(np.float32)
        while img.ndim > len(self.crop) + 2 and img.shape[-1] == 1:
            img = img[..., 0]
        out = oracle.crop_and_resize_forward(np.ascontiguousarray(img), np.ascontiguousarray(boxes.detach().numpy().astype(np.float32)),
                                             np.ascontiguousarray(box_ind.detach().numpy().astype(np.int32)), self.crop)
        return torch.from_numpy(out)


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


class torch04_integer_division(object):
    """`long_tensor / int` floor-divides in torch 0.4 (retina_unet.py:212 builds index tensors with it)"""

    def __enter__(self):
        self._orig = torch.Tensor.__truediv__

        def div(a, b):
            if not a.is_floating_point() and not (torch.is_tensor(b) and b.is_floating_point()) and not isinstance(b, float):
                return torch.div(a, b, rounding_mode="floor")
            return self._orig(a, b)
        torch.Tensor.__truediv__ = div

    def __exit__(self, *exc):
        torch.Tensor.__truediv__ = self._orig


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def main():
    log = logging.getLogger("glue_golden")
    log.addHandler(logging.NullHandler())
    out = {}
    cf = gi.make_cf("mrcnn")
    anchors = mu.generate_pyramid_anchors(log, cf)                      # float64 [A, 6]
    out["n_anchors"] = np.int64(anchors.shape[0])
    anchors_t = t(anchors).float()

    # ---- proposal_layer
    probs, deltas = gi.proposal_layer_inputs(anchors.shape[0])
    torch.manual_seed(0)
    nb, op = mr.proposal_layer(t(probs), t(deltas), cf.post_nms_rois_training, anchors_t, cf)
    out["proposal_normalized_boxes"] = nb.numpy()
    out["proposal_out_proposals"] = np.asarray(op)

    # ---- pyramid_roi_align
    fmaps, rois = gi.pyramid_inputs(cf)
    pooled = mr.pyramid_roi_align([t(f) for f in fmaps], t(rois), cf.pool_size, cf.pyramid_levels, cf.dim)
    out["pyramid_pooled"] = pooled.numpy()
    pooled_m = mr.pyramid_roi_align([t(f) for f in fmaps], t(rois), cf.mask_pool_size, cf.pyramid_levels, cf.dim)
    out["pyramid_pooled_mask"] = pooled_m.numpy()

    # ---- detection_target_layer
    bp, scores, gt_cls, gt_boxes, gt_masks = gi.target_layer_inputs(cf)
    torch.manual_seed(0)
    si, tc, td, tm = mr.detection_target_layer(t(bp), t(scores), gt_cls, gt_boxes, gt_masks, cf)
    order = np.argsort(si.numpy(), kind="stable")                        # positives are permuted by randperm: compare by index
    out["target_sample_indices"] = si.numpy()[order]
    out["target_class_ids"] = tc.numpy()[order]
    out["target_deltas"] = td.numpy()[order]
    out["target_masks"] = tm.numpy()[order].astype(np.uint8)
    # the head losses on those targets
    logits, pred_deltas, pred_masks = gi.head_loss_inputs(cf, si.shape[0])
    out["loss_mrcnn_class"] = mr.compute_mrcnn_class_loss(tc[order], t(logits)).numpy()
    out["loss_mrcnn_bbox"] = mr.compute_mrcnn_bbox_loss(td[order], t(pred_deltas), tc[order]).numpy()
    out["loss_mrcnn_mask"] = mr.compute_mrcnn_mask_loss(tm[order], t(pred_masks), tc[order]).numpy()

    # ---- RPN losses (one batch element; targets from the reference's own gt_anchor_matching; shem_poolsize = 1 and
    #      a large rpn_train_anchors_per_image make both the matching and the SHEM draw deterministic)
    gt_r, rlogits, rdeltas = gi.rpn_loss_inputs(anchors.shape[0])
    np.random.seed(0)
    match, rtargets = mu.gt_anchor_matching(cf, anchors, gt_r)
    out["rpn_match_counts"] = np.array([(match == 1).sum(), (match == -1).sum(), (match == 0).sum()])
    torch.manual_seed(0)
    rl, neg_ix = mr.compute_rpn_class_loss(t(match), t(rlogits), 1)
    out["loss_rpn_class"] = rl.numpy()
    out["loss_rpn_neg_ix"] = np.sort(np.nonzero(match == -1)[0][np.asarray(neg_ix)])
    out["loss_rpn_bbox"] = mr.compute_rpn_bbox_loss(t(rtargets).float(), t(rdeltas), t(match)).numpy()

    # ---- refine_detections (Mask R-CNN)
    rois_r, probs_r, deltas_r, bix = gi.refine_inputs(cf)
    det = mr.refine_detections(t(rois_r), t(probs_r), t(deltas_r), t(bix).float(), cf).numpy()
    out["refine_detections"] = det[np.lexsort(det.T[::-1])]

    # ---- refine_detections (Retina U-Net)
    cfr = gi.make_cf("retina_unet")
    anchors_r = mu.generate_pyramid_anchors(log, cfr)
    out["n_anchors_retina"] = np.int64(anchors_r.shape[0])
    probs_u, deltas_u, bix_u = gi.retina_refine_inputs(cfr, anchors_r.shape[0])
    with torch04_integer_division():
        det_u = ru.refine_detections(t(anchors_r).float(), t(probs_u), t(deltas_u), t(bix_u), cfr).numpy()
    out["retina_refine_detections"] = det_u[np.lexsort(det_u.T[::-1])]

    np.savez_compressed(os.path.join(HERE, "glue_reference.npz"), **out)
    for k, v in out.items():
        print(k, np.asarray(v).shape, np.asarray(v).dtype)


if __name__ == "__main__":
    main()
