"""Call-site glue parity: the re-designed, sync-free glue of
medicaldetectiontoolkit_amd/models/{mrcnn,retina_unet}.py against the outputs of the REFERENCE's own functions
(tests/golden/glue_reference.npz, produced by tests/golden/make_glue_golden.py from /root/reference) on the same
seeded inputs (tests/golden/glue_inputs.py).  Bars: NMS-kept sets / sampled sets / class ids / mask targets / integer
pixel boxes identical; pooled features bit-exact; decoded coordinates and losses to fp32 rounding (device expf)."""
import os

import numpy as np
import pytest
import torch

from tests.golden import glue_inputs as gi

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_reference.npz"))


def _t(a, dev, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return x if dtype is None else x.to(dtype)


def _rows_sorted(a):
    return a[np.lexsort(a.T[::-1])]


@pytest.fixture(scope="module")
def ctx(cuda):
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    cf = gi.make_cf("mrcnn")
    anchors_f64, anchors = mutils.generate_pyramid_anchors(None, cf, device=cuda, return_f32=True)
    assert anchors.shape[0] == int(G["n_anchors"])
    return cf, anchors_f64, anchors


def test_proposal_layer_vs_reference(ctx, cuda):
    """mrcnn.py:297-369: same proposals in the same (score) order; coordinates to device-expf rounding"""
    from medicaldetectiontoolkit_amd.models import mrcnn
    cf, _, anchors = ctx
    probs, deltas = gi.proposal_layer_inputs(anchors.shape[0])
    nb, op = mrcnn.proposal_layer(_t(probs, cuda), _t(deltas, cuda), cf.post_nms_rois_training, anchors, cf)
    op, nb = op.cpu().numpy(), nb.cpu().numpy()
    want_op, want_nb = G["proposal_out_proposals"], G["proposal_normalized_boxes"]
    assert op.shape == want_op.shape
    assert np.array_equal(op[:, :, 6], want_op[:, :, 6])                       # RPN scores of the kept anchors, in order
    assert np.abs(op[:, :, :6] - want_op[:, :, :6]).max() <= 2e-4             # pixels (64-px patch)
    assert np.abs(nb - want_nb).max() <= 1e-5


@pytest.mark.parametrize("which", ["pool", "mask_pool"])
def test_pyramid_roi_align_vs_reference(ctx, cuda, which):
    """mrcnn.py:373-457: level rule + per-level pooling + order restoration; bit-exact (the forward kernel equals the
    oracle bit for bit and the other levels contribute exact zeros)"""
    from medicaldetectiontoolkit_amd.models import mrcnn
    cf, _, _ = ctx
    fmaps, rois = gi.pyramid_inputs(cf)
    pool = cf.pool_size if which == "pool" else cf.mask_pool_size
    got = mrcnn.pyramid_roi_align([_t(f, cuda) for f in fmaps], _t(rois, cuda), pool, cf.pyramid_levels, cf.dim)
    want = G["pyramid_pooled" if which == "pool" else "pyramid_pooled_mask"]
    assert np.array_equal(got.cpu().numpy(), want), np.abs(got.cpu().numpy() - want).max()


def _targets(ctx, cuda):
    from medicaldetectiontoolkit_amd.models import mrcnn
    cf, _, _ = ctx
    bp, scores, gt_cls, gt_boxes, gt_masks = gi.target_layer_inputs(cf)
    masks = torch.cat([_t(m, cuda).unsqueeze(1) for m in gt_masks], 0)          # [sum_G, 1, Y, X, Z]
    gen = torch.Generator(device=cuda).manual_seed(0)
    si, valid, is_pos, tc, td, tm = mrcnn.detection_target_layer(_t(bp, cuda), _t(scores, cuda), gt_cls, gt_boxes, masks, cf,
                                                                 gi.B, generator=gen)
    v = valid.cpu().numpy()
    si, tc, td, tm, is_pos = [x.cpu().numpy()[v] for x in (si, tc, td, tm, is_pos)]
    order = np.argsort(si, kind="stable")
    return si[order], tc[order], td[order], tm[order], is_pos[order]


def test_detection_target_layer_vs_reference(ctx, cuda):
    """mrcnn.py:461-613: sampled positives / SHEM negatives (as sets), class / delta / mask targets per sample"""
    si, tc, td, tm, is_pos = _targets(ctx, cuda)
    assert np.array_equal(si, G["target_sample_indices"])
    assert np.array_equal(tc.astype(np.int64), G["target_class_ids"].astype(np.int64))
    assert np.array_equal(is_pos, G["target_class_ids"] > 0)
    assert np.abs(td - G["target_deltas"]).max() <= 1e-5
    assert np.array_equal(tm.astype(np.uint8), G["target_masks"])


def test_head_losses_vs_reference(ctx, cuda):
    """compute_mrcnn_{class,bbox,mask}_loss (mrcnn.py:243-286) on the targets above"""
    from medicaldetectiontoolkit_amd.models import mrcnn
    cf, _, _ = ctx
    si, tc, td, tm, is_pos = _targets(ctx, cuda)
    logits, pred_deltas, pred_masks = gi.head_loss_inputs(cf, si.shape[0])
    tc_t, pos_t = _t(tc, cuda), _t(is_pos, cuda)
    valid = torch.ones(si.shape[0], dtype=torch.bool, device=cuda)
    lc = mrcnn.compute_mrcnn_class_loss(tc_t, _t(logits, cuda), valid).item()
    lb = mrcnn.compute_mrcnn_bbox_loss(_t(td, cuda), _t(pred_deltas, cuda), tc_t, pos_t).item()
    lm = mrcnn.compute_mrcnn_mask_loss(_t(tm, cuda).float(), _t(pred_masks, cuda), tc_t, pos_t).item()
    assert abs(lc - float(G["loss_mrcnn_class"])) <= 1e-5 * max(1.0, abs(lc))
    assert abs(lb - float(G["loss_mrcnn_bbox"])) <= 1e-5 * max(1.0, abs(lb))
    assert abs(lm - float(G["loss_mrcnn_mask"])) <= 1e-5 * max(1.0, abs(lm))


def test_rpn_losses_vs_reference(ctx, cuda):
    """gt_anchor_matching (model_utils.py:505-619) -> compute_rpn_class_loss / compute_rpn_bbox_loss (mrcnn.py:176-240):
    matching on the device, SHEM negatives, delta targets and both loss values"""
    from medicaldetectiontoolkit_amd.models import mrcnn
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    cf, anchors_f64, _ = ctx
    gt, logits, pred_deltas = gi.rpn_loss_inputs(anchors_f64.shape[0])
    m, am, _, _ = mutils.anchor_match_labels(anchors_f64, _t(gt, cuda), None, 0.01, float(cf.anchor_matching_iou))
    mm = m.cpu().numpy()
    assert [int((mm == 1).sum()), int((mm == -1).sum()), int((mm == 0).sum())] == G["rpn_match_counts"].tolist()
    gen = torch.Generator(device=cuda).manual_seed(0)
    cl, bl, (pidx, pvalid, nidx, nvalid) = mrcnn.compute_rpn_losses(
        m[None], am[None], _t(logits, cuda)[None], _t(pred_deltas, cuda)[None], anchors_f64, [gt], cf, generator=gen, shem_poolsize=1)
    neg = np.sort(nidx.cpu().numpy()[nvalid.cpu().numpy()])
    assert np.array_equal(neg, G["loss_rpn_neg_ix"])
    assert abs(cl.item() - float(G["loss_rpn_class"])) <= 1e-5 * max(1.0, abs(cl.item()))
    assert abs(bl.item() - float(G["loss_rpn_bbox"])) <= 1e-5 * max(1.0, abs(bl.item()))


def test_refine_detections_vs_reference(ctx, cuda):
    """mrcnn.py:620-714: per-(element, class) NMS at 1e-5, top-k per element; same detections (integer pixel boxes,
    element, class, score)"""
    from medicaldetectiontoolkit_amd.models import mrcnn
    cf, _, _ = ctx
    rois, probs, deltas, bix = gi.refine_inputs(cf)
    det, valid = mrcnn.refine_detections(_t(rois, cuda), _t(probs, cuda), _t(deltas, cuda), _t(bix, cuda), cf, gi.B)
    got = _rows_sorted(det.cpu().numpy()[valid.cpu().numpy()])
    want = G["refine_detections"]
    assert got.shape == want.shape
    assert np.array_equal(got[:, :8], want[:, :8])
    assert np.abs(got[:, 8] - want[:, 8]).max() <= 1e-6


def test_retina_refine_detections_vs_reference(cuda):
    """retina_unet.py:194-271"""
    from medicaldetectiontoolkit_amd.models import retina_unet
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    cf = gi.make_cf("retina_unet")
    _, anchors = mutils.generate_pyramid_anchors(None, cf, device=cuda, return_f32=True)
    assert anchors.shape[0] == int(G["n_anchors_retina"])
    probs, deltas, _ = gi.retina_refine_inputs(cf, anchors.shape[0])
    det, valid = retina_unet.refine_detections(anchors, _t(probs, cuda), _t(deltas, cuda), gi.B, cf)
    got = _rows_sorted(det.cpu().numpy()[valid.cpu().numpy()])
    want = G["retina_refine_detections"]
    assert got.shape == want.shape
    assert np.array_equal(got[:, :8], want[:, :8])
    assert np.abs(got[:, 8] - want[:, 8]).max() <= 1e-6
