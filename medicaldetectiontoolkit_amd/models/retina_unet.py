"""Retina U-Net / Retina Net (2D / 3D) on the gfx950 hot-path kernels.

Mirror of the reference's models/retina_unet.py (and retina_net.py, which differs only by the missing
segmentation head): same module names / state_dict keys (Fpn, Classifier, BBRegressor, final_conv), same
net(cf, logger) with train_forward / test_forward / forward.  Native call sites: per-(element, class) NMS in
refine_detections (retina_unet.py:248-250) and the per-element numpy gt_anchor_matching (:416).

refine_detections here runs ONE device NMS over the global top pre_nms_limit candidates: boxes of different
(batch element, class) groups are shifted apart along y by a multiple of 4096 px, so cross-group IoU is exactly 0
while in-group IoUs are unchanged bit for bit (rounded pixel coordinates and their +1 extents are exact in fp32) --
equivalent to the reference's loop of per-group NMS calls, without its nonzero()/unique1d() host syncs.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..cuda_functions import _nms_impl
from ..utils import fused_epilogue
from ..utils import model_utils as mutils
from . import backbone as backbone_module
from .mrcnn import GtOnDevice, compute_rpn_losses

GROUP_SHIFT = 4096.0


class Classifier(nn.Module):
    """retina_unet.py:40-79."""

    def __init__(self, cf, conv):
        super(Classifier, self).__init__()
        self.dim = conv.dim
        self.n_classes = cf.head_classes
        nf, s = cf.n_rpn_features, cf.rpn_anchor_stride
        self.conv_1 = conv(cf.end_filts, nf, ks=3, stride=s, pad=1, relu=cf.relu)
        self.conv_2 = conv(nf, nf, ks=3, stride=s, pad=1, relu=cf.relu)
        self.conv_3 = conv(nf, nf, ks=3, stride=s, pad=1, relu=cf.relu)
        self.conv_4 = conv(nf, nf, ks=3, stride=s, pad=1, relu=cf.relu)
        self.conv_final = conv(nf, cf.n_anchors_per_pos * cf.head_classes, ks=3, stride=s, pad=1, relu=None)

    def forward(self, x):
        x = self.conv_4(self.conv_3(self.conv_2(self.conv_1(x))))
        class_logits = self.conv_final(x)
        axes = (0, 2, 3, 1) if self.dim == 2 else (0, 2, 3, 4, 1)
        return [class_logits.permute(*axes).contiguous().view(x.size(0), -1, self.n_classes)]


class BBRegressor(nn.Module):
    """retina_unet.py:82-119."""

    def __init__(self, cf, conv):
        super(BBRegressor, self).__init__()
        self.dim = conv.dim
        nf, s = cf.n_rpn_features, cf.rpn_anchor_stride
        self.conv_1 = conv(cf.end_filts, nf, ks=3, stride=s, pad=1, relu=cf.relu)
        self.conv_2 = conv(nf, nf, ks=3, stride=s, pad=1, relu=cf.relu)
        self.conv_3 = conv(nf, nf, ks=3, stride=s, pad=1, relu=cf.relu)
        self.conv_4 = conv(nf, nf, ks=3, stride=s, pad=1, relu=cf.relu)
        self.conv_final = conv(nf, cf.n_anchors_per_pos * self.dim * 2, ks=3, stride=s, pad=1, relu=None)

    def forward(self, x):
        x = self.conv_4(self.conv_3(self.conv_2(self.conv_1(x))))
        bb_logits = self.conv_final(x)
        axes = (0, 2, 3, 1) if self.dim == 2 else (0, 2, 3, 4, 1)
        return [bb_logits.permute(*axes).contiguous().view(x.size(0), -1, self.dim * 2)]


def batch_dice(pred, y, false_positive_weight=1.0, smooth=1e-6):
    """utils/model_utils.py:845-866: soft dice over the batch pseudo-volume, foreground classes only."""
    axes = (0,) + tuple(range(2, pred.dim()))
    intersect = (pred * y).sum(axes)
    denom = (false_positive_weight * pred + y).sum(axes)
    return torch.mean(((2 * intersect + smooth) / (denom + smooth))[1:])


def refine_detections(anchors, probs, deltas, B, cf):
    """retina_unet.py:194-271.  anchors [A, 2*dim] fp32; probs [B*A, K]; deltas [B*A, 2*dim].
    Returns detections [B*M, 2*dim+3] (pixel box rounded, batch_ix, class_id, score) and a validity mask."""
    dev = probs.device
    dim = cf.dim
    A = anchors.shape[0]
    fg = probs.shape[1] - 1
    fg_probs = probs[:, 1:].contiguous()
    n_pre = min(cf.pre_nms_limit, fg_probs.numel())
    flat_probs, keep_ix = torch.topk(fg_probs.view(-1), n_pre, sorted=True)            # :208-210
    row = torch.div(keep_ix, fg, rounding_mode="floor")
    cls = keep_ix % fg + 1
    bix = torch.div(row, A, rounding_mode="floor")
    scale = mutils.const_tensor(cf.scale, torch.float32, dev)
    no_clip = [-3e38, -3e38, 3e38, 3e38] + ([-3e38, 3e38] if dim == 3 else [])
    dec = mutils.decode_clip_boxes((anchors[row % A] / scale).contiguous(), deltas[row].contiguous(),
                                   np.asarray(cf.rpn_bbox_std_dev, dtype=np.float32), no_clip) * scale      # :226-228
    rois = torch.round(mutils.clip_boxes(dec, [float(v) for v in cf.window]))
    # one NMS for all (element, class) groups: shift groups apart along y
    group = (bix * fg + (cls - 1)).float()
    shifted = rois.clone()
    shifted[:, 0] += group * GROUP_SHIFT
    shifted[:, 2] += group * GROUP_SHIFT
    dets = torch.cat([shifted, flat_probs.unsqueeze(1)], 1).contiguous()               # already sorted by score
    keep, num = _nms_impl.nms_sorted(dets, float(cf.detection_nms_threshold), dim)
    kept = torch.zeros(n_pre + 1, dtype=torch.bool, device=dev)
    kept.scatter_(0, torch.where(keep >= 0, keep, torch.full_like(keep, n_pre)), True)      # (indexed assignment would sync)
    kept = kept[:n_pre]
    # top model_max_instances_per_batch_element per batch element (:261-263)
    M = cf.model_max_instances_per_batch_element
    per_b = torch.where(kept[None, :] & (bix[None, :] == torch.arange(B, device=dev)[:, None]), flat_probs[None, :],
                        torch.full((1, 1), -1.0, device=dev))
    top_s, top_i = torch.topk(per_b, min(M, n_pre), dim=1)
    valid = top_s >= 0
    out = torch.cat([rois[top_i], torch.arange(B, device=dev, dtype=torch.float32)[:, None, None].expand(-1, top_i.shape[1], 1),
                     cls[top_i].float().unsqueeze(-1), top_s.unsqueeze(-1)], 2)
    out = out * valid.unsqueeze(-1).to(out.dtype)
    return out.view(-1, 2 * dim + 3), valid.view(-1)


def get_results(cf, img_shape, detections, det_valid, seg_logits, box_results_list=None, seg_preds=None):
    """retina_unet.py:275-335.  detections / det_valid: device tensors or host arrays that were already read back; seg_preds: the label map already
    computed (deferred read-out) instead of seg_logits."""
    det_a = detections if isinstance(detections, np.ndarray) else detections.detach().cpu().numpy()
    val_a = det_valid if isinstance(det_valid, np.ndarray) else det_valid.detach().cpu().numpy()
    det = det_a[val_a.astype(bool)]
    dim = cf.dim
    if box_results_list is None:
        box_results_list = [[] for _ in range(img_shape[0])]
    batch_ixs = det[:, dim * 2] if det.shape[0] else np.zeros(0)
    for ix in range(img_shape[0]):
        d = det[batch_ixs == ix]
        if d.shape[0] == 0:
            continue
        boxes = d[:, :2 * dim].astype(np.int32)
        class_ids = d[:, 2 * dim + 1].astype(np.int32)
        scores = d[:, 2 * dim + 2]
        ext = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
        if dim == 3:
            ext = ext * (boxes[:, 5] - boxes[:, 4])
        for ix2 in np.nonzero(ext > 0)[0]:
            if scores[ix2] >= cf.model_min_confidence:
                box_results_list[ix].append({"box_coords": boxes[ix2], "box_score": scores[ix2], "box_type": "det",
                                             "box_pred_class_id": class_ids[ix2]})
    results_dict = {"boxes": box_results_list}
    if seg_preds is not None:
        results_dict["seg_preds"] = seg_preds
    elif seg_logits is None:
        results_dict["seg_preds"] = np.zeros(img_shape)[:, 0][:, np.newaxis]
    else:
        results_dict["seg_preds"] = F.softmax(seg_logits, 1).argmax(1).cpu().numpy()[:, np.newaxis].astype("uint8")
    return results_dict


class net(nn.Module):
    def __init__(self, cf, logger=None, device=None):
        super(net, self).__init__()
        self.cf = cf
        self.logger = logger
        self.device_ = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.build()

    def build(self):
        h, w = self.cf.patch_size[:2]
        if h / 2 ** 5 != int(h / 2 ** 5) or w / 2 ** 5 != int(w / 2 ** 5):
            raise Exception("Image size must be dividable by 2 at least 5 times to avoid fractions when downscaling and upscaling.")
        conv = mutils.NDConvGenerator(self.cf.dim)
        self.anchors_f64, self.anchors = mutils.generate_pyramid_anchors(self.logger, self.cf, device=self.device_, return_f32=True)
        self.Fpn = backbone_module.FPN(self.cf, conv, operate_stride1=self.cf.operate_stride1)
        self.Classifier = Classifier(self.cf, conv)
        self.BBRegressor = BBRegressor(self.cf, conv)
        if self.cf.model == "retina_unet":
            self.final_conv = conv(self.cf.end_filts, self.cf.num_seg_classes, ks=1, pad=0, norm=None, relu=None)
        self.to(self.device_)
        self.memory_format = None
        if getattr(self.cf, "channels_last", False):
            self.memory_format = torch.channels_last_3d if self.cf.dim == 3 else torch.channels_last
            self.to(memory_format=self.memory_format)

    @property
    def np_anchors(self):
        return self.anchors_f64.cpu().numpy()

    def forward(self, img):
        """retina_unet.py:477-513."""
        if self.memory_format is not None:
            img = img.contiguous(memory_format=self.memory_format)
        # retina_unet: fpn_outs[0] (P0_conv2's output) is read by final_conv only, and neither layer has an activation: the two are composed into one
        # 36 -> 2 3x3x3 layer where csrc/conv_seg.hip serves the shape (the 36 -> 36 layer on the full-resolution map is 58 ms of the step at 8 x 128^3)
        compose = self.cf.model == "retina_unet" and self.cf.operate_stride1 and fused_epilogue.SEG_HEAD_COMPOSED \
            and getattr(self.Fpn, "P0_conv2", None) is not None and self._seg_compose_ok(img)
        self.Fpn.defer_p0_conv2 = bool(compose)
        try:
            fpn_outs = self.Fpn(img)
        finally:
            self.Fpn.defer_p0_conv2 = False
        off = 1 if self.cf.operate_stride1 else 0
        if compose and fused_epilogue.seg_head_composed_applies(self.Fpn.P0_conv2, self.final_conv, fpn_outs[0]):
            seg_logits = fused_epilogue.seg_head_composed(self.Fpn.P0_conv2, self.final_conv, fpn_outs[0])
        elif compose:
            seg_logits = self.final_conv(self.Fpn.P0_conv2(fpn_outs[0]))
        else:
            seg_logits = self.final_conv(fpn_outs[0]) if self.cf.model == "retina_unet" else None
        selected = [fpn_outs[i + off] for i in self.cf.pyramid_levels]
        class_logits = torch.cat([self.Classifier(p)[0] for p in selected], dim=1)
        bb_outputs = torch.cat([self.BBRegressor(p)[0] for p in selected], dim=1)
        B = class_logits.shape[0]
        flat_class_softmax = F.softmax(class_logits.detach().view(-1, class_logits.shape[-1]), 1)
        flat_bb_outputs = bb_outputs.detach().view(-1, bb_outputs.shape[-1])
        detections, det_valid = refine_detections(self.anchors, flat_class_softmax, flat_bb_outputs, B, self.cf)
        return detections, det_valid, class_logits, bb_outputs, seg_logits

    def _seg_compose_ok(self, img):
        """cheap pre-check (before the FPN runs) that the composed segmentation layer can apply: fp32 GPU input, supported full-resolution shape"""
        from .. import _lib
        if not (img.is_cuda and img.dtype == torch.float32 and img.dim() == 5 and not torch.is_autocast_enabled()):
            return False
        c2 = self.Fpn.P0_conv2
        return isinstance(c2, fused_epilogue.ConvBias) and isinstance(self.final_conv, fused_epilogue.ConvBias) and bool(
            _lib.lib().mdt_conv_seg_supported(int(c2.in_channels), int(self.final_conv.out_channels), int(img.shape[2]), int(img.shape[3]), int(img.shape[4])))

    # ------------------------------------------------------------------ which parameters have a gradient only under a condition of the step
    def grad_condition_spec(self):
        """[(condition, [parameters])] for training.FlatAdam.attach_conditions: the reference's compute_class_loss returns constants for a step
        without sampled anchors (retina_unet.py:139-165) and compute_bbox_loss without a positive anchor (:180-186), so autograd hands the
        Classifier / BBRegressor heads no gradient there and torch.optim.Adam (exec.py:74) leaves them alone.  The Retina U-Net's FPN and
        final_conv always learn from the segmentation loss; the Retina Net's FPN only from the anchors."""
        spec = [("anchor_samples", list(self.Classifier.parameters())), ("positive_anchors", list(self.BBRegressor.parameters()))]
        if self.cf.model != "retina_unet":
            spec.append(("anchor_samples_fpn", list(self.Fpn.parameters())))
        return spec

    def set_grad_cond_buffer(self, t):
        """float32 device tensor (or None) that train_forward fills with the counts of grad_condition_spec()"""
        self._grad_cond = t

    def train_forward(self, batch, monitor=True, **kwargs):
        """retina_unet.py:381-457."""
        cf, dev = self.cf, self.device_
        ev = batch.get("ready_event")            # training.DevicePrefetcher uploaded the image on its side stream
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            if torch.is_tensor(batch.get("data")):
                batch["data"].record_stream(cur)
        img = mutils.upload(batch["data"], dev).float()
        gt_class_ids, gt_boxes = batch["roi_labels"], batch["bb_target"]
        B = img.shape[0]
        gt_dev = GtOnDevice(gt_boxes, gt_class_ids, cf.dim, dev)      # one pinned async upload, before the backbone launch
        detections, det_valid, class_logits, pred_deltas, seg_logits = self.forward(img)
        # anchor matching of the whole batch: ONE launch pair, GT counts read on the device (the reference: one numpy
        # gt_anchor_matching per element on the host, retina_unet.py:408-420)
        neg_thr = 0.1 if cf.dim == 2 else 0.01
        rpn_match, rpn_argmax = mutils.anchor_match_labels_batched(self.anchors_f64, gt_dev.px, gt_dev.n_gt, gt_dev.cls_i32, neg_thr,
                                                                   float(cf.anchor_matching_iou))
        batch_class_loss, batch_bbox_loss, samples = compute_rpn_losses(
            rpn_match, rpn_argmax, class_logits, pred_deltas, self.anchors_f64, gt_boxes, cf,
            shem_poolsize=getattr(cf, "retina_shem_poolsize", 20),      # the reference calls compute_class_loss with its default 20 (retina_unet.py:432)
            gt_dev=gt_dev)
        loss = batch_class_loss + batch_bbox_loss
        gc = getattr(self, "_grad_cond", None)
        if gc is not None:          # who had something to learn from in this step (FlatAdam skips the others like torch.optim.Adam does)
            n_pos, n_neg = samples[1].sum(), samples[3].sum()
            vals = [n_pos + n_neg, n_pos] + ([n_pos + n_neg] if cf.model != "retina_unet" else [])
            gc.copy_(torch.stack(vals))
        seg_dice = seg_ce = None
        if seg_logits is not None:
            var_seg = mutils.upload(batch["seg"], dev, channel="seg").long()
            ohe = F.one_hot(var_seg[:, 0], cf.num_seg_classes).movedim(-1, 1).float()
            seg_dice = 1 - batch_dice(F.softmax(seg_logits, dim=1), ohe)
            seg_ce = F.cross_entropy(seg_logits, var_seg[:, 0])
            loss = loss + (seg_dice + seg_ce) / 2
        results_dict = {"torch_loss": loss,
                        "loss_terms": {"class": batch_class_loss.detach(), "bbox": batch_bbox_loss.detach(),
                                       "seg_dice": None if seg_dice is None else seg_dice.detach(),
                                       "seg_ce": None if seg_ce is None else seg_ce.detach()}}
        if monitor == "deferred":
            # the read-out one step late (mrcnn.net.train_forward has the same mode): the sampled anchors, detections and loss values packed into ONE
            # buffer, the label map as uint8 computed on the device (soft-max + arg-max as get_results does) with its mean -- both travel with
            # asynchronous copies into pinned memory; the entries returned now are those of the PREVIOUS call.  No host sync in the step.
            vals = [loss, batch_class_loss, batch_bbox_loss] + ([seg_dice, seg_ce] if seg_logits is not None else [])
            seg_u8 = None
            if seg_logits is not None:
                seg_u8 = F.softmax(seg_logits.detach(), 1).argmax(1).to(torch.uint8)
                vals = vals + [seg_u8.float().mean()]
            items = [("pidx", samples[0]), ("pvalid", samples[1]), ("nidx", samples[2]), ("nvalid", samples[3]), ("detections", detections),
                     ("det_valid", det_valid), ("vals", torch.stack([x.detach().float() for x in vals]))]
            if getattr(self, "_deferred", None) is None:
                self._deferred = mutils.DeferredReadout()
            prev = self._deferred.push(mutils.pack_for_readout(items), (batch, tuple(img.shape), seg_logits is not None), extra=seg_u8)
            if prev is not None:
                results_dict.update(self._resolve_deferred(prev))
            return results_dict
        if monitor:
            box_results_list = [[] for _ in range(B)]
            for b in range(B):
                for ix in range(len(gt_boxes[b])):
                    box_results_list[b].append({"box_coords": batch["bb_target"][b][ix], "box_label": batch["roi_labels"][b][ix], "box_type": "gt"})
            pidx, pvalid, nidx, nvalid = [t.cpu().numpy() for t in samples]
            if getattr(self, "_anchors_host", None) is None:
                self._anchors_host = self.anchors.cpu().numpy()      # constant table: read back once, not every step
            anchors_np = self._anchors_host
            for b in range(B):
                for a in anchors_np[pidx[b][pvalid[b]]]:
                    box_results_list[b].append({"box_coords": a, "box_type": "pos_anchor"})
                for a in anchors_np[nidx[b][nvalid[b]]]:
                    box_results_list[b].append({"box_coords": a, "box_type": "neg_anchor"})
            results_dict.update(get_results(cf, img.shape, detections, det_valid, seg_logits, box_results_list))
            vals = [loss, batch_class_loss, batch_bbox_loss] + ([seg_dice, seg_ce] if seg_logits is not None else [])
            v = torch.stack([x.detach() for x in vals]).cpu().numpy()
            results_dict["monitor_values"] = {"loss": float(v[0]), "class_loss": float(v[1])}
            results_dict["logger_string"] = "loss: {0:.2f}, class: {1:.2f}, bbox: {2:.2f}".format(v[0], v[1], v[2]) + (
                ", seg dice: {0:.3f}, seg ce: {1:.3f}, mean pix. pr.: {2:.5f}".format(v[3], v[4], float(np.mean(results_dict["seg_preds"])))
                if seg_logits is not None else "")
        return results_dict

    def _resolve_deferred(self, entry):
        packed, ctx = mutils.DeferredReadout.resolve(entry)
        if isinstance(ctx, tuple) and len(ctx) == 2 and torch.is_tensor(ctx[1]):
            (batch, img_shape, has_seg), seg_host = ctx
        else:
            (batch, img_shape, has_seg), seg_host = ctx, None
        r = mutils.unpack_readout(packed)
        B = img_shape[0]
        box_results_list = [[] for _ in range(B)]
        for b in range(B):
            for ix in range(len(batch["bb_target"][b])):
                box_results_list[b].append({"box_coords": batch["bb_target"][b][ix], "box_label": batch["roi_labels"][b][ix], "box_type": "gt"})
        if getattr(self, "_anchors_host", None) is None:
            self._anchors_host = self.anchors.cpu().numpy()
        anchors_np = self._anchors_host
        pidx, pvalid, nidx, nvalid = r["pidx"], r["pvalid"].astype(bool), r["nidx"], r["nvalid"].astype(bool)
        for b in range(B):
            for a in anchors_np[pidx[b][pvalid[b]]]:
                box_results_list[b].append({"box_coords": a, "box_type": "pos_anchor"})
            for a in anchors_np[nidx[b][nvalid[b]]]:
                box_results_list[b].append({"box_coords": a, "box_type": "neg_anchor"})
        seg_preds = seg_host.numpy()[:, np.newaxis].copy() if seg_host is not None else None      # (the pinned buffer is re-used two steps later)
        res = get_results(self.cf, img_shape, r["detections"], r["det_valid"], None, box_results_list, seg_preds=seg_preds)
        v = r["vals"]
        res["monitor_values"] = {"loss": float(v[0]), "class_loss": float(v[1])}
        res["logger_string"] = "loss: {0:.2f}, class: {1:.2f}, bbox: {2:.2f}".format(v[0], v[1], v[2]) + (
            ", seg dice: {0:.3f}, seg ce: {1:.3f}, mean pix. pr.: {2:.5f}".format(v[3], v[4], float(v[5])) if has_seg else "")
        res["monitor_of_previous_step"] = True
        return res

    def flush_deferred_monitor(self):
        """read-out entries of the LAST train_forward(monitor="deferred") call, or None"""
        d = getattr(self, "_deferred", None)
        entry = d.flush() if d is not None else None
        return self._resolve_deferred(entry) if entry is not None else None

    def test_forward(self, batch, **kwargs):
        """retina_unet.py:459-475."""
        img = batch["data"]
        img = torch.from_numpy(np.ascontiguousarray(img)).to(self.device_).float() if not torch.is_tensor(img) else img.to(self.device_).float()
        with torch.no_grad():
            detections, det_valid, _, _, seg_logits = self.forward(img)
        return get_results(self.cf, img.shape, detections, det_valid, seg_logits)

    def test_forward_detections(self, img):
        """test_forward without leaving the device (predictor.collect_raw_boxes): (rows [B * M, 2 * dim + 3] = integer box,
        batch_ix, class id, score; keep = real detection, positive extent, score >= model_min_confidence, :291-300)"""
        with torch.no_grad():
            img = img.to(self.device_).float()
            det, det_valid, _, _, _ = self.forward(img)
            dim = self.cf.dim
            boxes = det[:, :2 * dim].to(torch.int32).float()
            ext = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
            if dim == 3:
                ext = ext * (boxes[:, 5] - boxes[:, 4])
            keep = det_valid & (ext > 0) & (det[:, 2 * dim + 2] >= self.cf.model_min_confidence)
            return torch.cat([boxes, det[:, 2 * dim:]], 1), keep
