"""GPU: the model call-site owners run end to end on the HIP kernels (training step and test_forward),
losses are finite, every parameter the architecture uses receives a gradient, results follow the
reference's results_dict format."""
import os

import numpy as np
import pytest
import torch

from medicaldetectiontoolkit_amd import training
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn, retina_unet
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch

pytestmark = pytest.mark.gpu


def _check_results(res, B, dim):
    assert set(res) >= {"boxes", "seg_preds"}
    assert len(res["boxes"]) == B
    for boxes in res["boxes"]:
        for b in boxes:
            assert "box_coords" in b and "box_type" in b
            if b["box_type"] == "det":
                assert len(b["box_coords"]) == 2 * dim and "box_score" in b and "box_pred_class_id" in b


@pytest.mark.parametrize("dim,patch", [(3, [64, 64, 32]), (2, [64, 64])])
def test_mrcnn_train_and_test_forward(dim, patch, cuda):
    B = 2
    cf = Configs(dim=dim, model="mrcnn", patch_size=patch, batch_size=B)
    torch.manual_seed(0)
    net = mrcnn.net(cf, device=cuda)
    opt = training.build_optimizer(net, cf)
    batches = [make_batch(patch, B, seed=s, with_empty=(s == 1)) for s in range(3)]
    losses = []
    for it in range(3):
        r = training.train_step(net, opt, batches[it], monitor=True)
        assert torch.isfinite(r["torch_loss"]).all()
        losses.append(float(r["torch_loss"]))
        _check_results(r, B, dim)
        assert "monitor_values" in r and "logger_string" in r
    unused = {n for n, p in net.named_parameters() if p.grad is None}
    assert unused <= {"fpn.P1_conv1.weight", "fpn.P1_conv1.bias", "fpn.P1_conv2.weight", "fpn.P1_conv2.bias"}, unused
    res = net.test_forward(batches[0], return_masks=True)
    _check_results(res, B, dim)
    assert res["seg_preds"].shape == (B, 1) + tuple(patch)


def test_mrcnn_batch_norm_heads_see_only_real_samples(cuda):
    """cf.norm = 'batch_norm': the RoI heads must not normalise over the padding slots of the fixed-size glue -- their
    running statistics after a training step equal those of a forward over the valid rows alone"""
    patch, B = [64, 64, 32], 2
    cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B, norm="batch_norm")
    torch.manual_seed(0)
    net = mrcnn.net(cf, device=cuda)
    assert net.classifier.compact_rows and net.mask.compact_rows
    opt = training.build_optimizer(net, cf)
    r = training.train_step(net, opt, make_batch(patch, B, seed=0), monitor=False)
    assert torch.isfinite(r["torch_loss"]).all()
    # direct check on the mask head: padding rows (batch_ix = -1) leave outputs zero and do not touch the statistics
    fm = [torch.randn(B, cf.end_filts, *[int(v) for v in s], device=cuda) for s in cf.backbone_shapes]
    rois = torch.tensor([[0.1, 0.1, 0.4, 0.4, 0.1, 0.6, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0],
                         [0.3, 0.2, 0.8, 0.7, 0.2, 0.9, 1.0], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0]], device=cuda)
    net.train()
    bn = [m for m in net.mask.modules() if isinstance(m, torch.nn.BatchNorm3d)][0]
    before = bn.running_mean.clone()
    out_all = net.mask(fm, rois)
    after_all = bn.running_mean.clone()
    bn.running_mean.copy_(before)
    out_valid = net.mask(fm, rois[[0, 2]])
    assert torch.allclose(after_all, bn.running_mean, atol=1e-6)
    assert torch.allclose(out_all[[0, 2]], out_valid, atol=1e-5) and float(out_all[[1, 3]].abs().max()) == 0.0


@pytest.mark.parametrize("model,dim,patch", [("retina_unet", 3, [64, 64, 32]), ("retina_net", 2, [64, 64])])
def test_retina_train_and_test_forward(model, dim, patch, cuda):
    B = 2
    cf = Configs(dim=dim, model=model, patch_size=patch, batch_size=B)
    torch.manual_seed(0)
    net = retina_unet.net(cf, device=cuda)
    opt = training.build_optimizer(net, cf)
    batches = [make_batch(patch, B, seed=s, with_empty=(s == 1)) for s in range(2)]
    for it in range(2):
        r = training.train_step(net, opt, batches[it], monitor=True)
        assert torch.isfinite(r["torch_loss"]).all()
        _check_results(r, B, dim)
    res = net.test_forward(batches[0])
    _check_results(res, B, dim)
    assert res["seg_preds"].shape == (B, 1) + tuple(patch)


def test_refine_detections_group_shift_equals_per_group_nms(cuda):
    """One NMS over y-shifted groups == the reference's loop of per-(element, class) NMS calls."""
    from medicaldetectiontoolkit_amd.cuda_functions.nms_3D.pth_nms import nms_gpu
    rng = np.random.default_rng(0)
    n, groups = 1200, 6
    c = rng.uniform(10, 110, size=(n, 3))
    s = rng.uniform(4, 30, size=(n, 3))
    boxes = np.round(np.stack([c[:, 0] - s[:, 0], c[:, 1] - s[:, 1], c[:, 0] + s[:, 0], c[:, 1] + s[:, 1], c[:, 2] - s[:, 2], c[:, 2] + s[:, 2]], 1))
    scores = rng.permutation(np.linspace(0.1, 1, n))
    g = rng.integers(0, groups, size=n)
    dets = torch.from_numpy(np.concatenate([boxes, scores[:, None]], 1).astype(np.float32)).to(cuda)
    shifted = dets.clone()
    shifted[:, 0] += torch.from_numpy(g).to(cuda).float() * retina_unet.GROUP_SHIFT
    shifted[:, 2] += torch.from_numpy(g).to(cuda).float() * retina_unet.GROUP_SHIFT
    one = set(nms_gpu(shifted, 1e-5).cpu().numpy().tolist())
    per = set()
    for k in range(groups):
        idx = np.nonzero(g == k)[0]
        per |= set(idx[nms_gpu(dets[torch.from_numpy(idx).to(cuda)], 1e-5).cpu().numpy()].tolist())
    assert one == per


def test_patch_tiled_prediction_with_wbc(cuda):
    """BASELINE config 5 in miniature: volume -> overlapping patches -> test_forward -> patient coordinates ->
    weighted box clustering (device); also under bf16 autocast."""
    from medicaldetectiontoolkit_amd import predictor
    cf = Configs(dim=3, model="mrcnn", patch_size=[64, 64, 32], batch_size=4)
    torch.manual_seed(0)
    net = mrcnn.net(cf, device=cuda)
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((1, 100, 90, 48)).astype(np.float32)
    res = predictor.predict_patient(net, vol, cf, n_ens=1)
    from medicaldetectiontoolkit_amd.utils.dataloader_utils import get_patch_crop_coords
    n_expected = get_patch_crop_coords(vol[0], cf.patch_size).shape[0]
    assert res["n_patches"] == n_expected and n_expected > 4
    assert len(res["boxes"]) == 1 and len(res["boxes"][0]) <= res["n_raw_boxes"]
    for b in res["boxes"][0]:
        c = b["box_coords"]
        assert len(c) == 6 and 0 < b["box_score"] <= 1.0 + 1e-9 and b["box_pred_class_id"] in (1, 2)
        assert c[0] >= -1 and c[2] <= 101 and c[1] >= -1 and c[3] <= 91 and c[4] >= -1 and c[5] <= 49
    res16 = predictor.predict_patient(net, vol, cf, n_ens=1, amp_dtype=torch.bfloat16)
    assert res16["n_patches"] == n_expected
    # test-time augmentation: 4 mirrored passes, boxes mirrored back into the original frame
    res_tta = predictor.predict_patient(net, vol, cf, test_aug=True)
    assert res_tta["n_passes"] == 4 and res_tta["n_raw_boxes"] >= res["n_raw_boxes"]
    for b in res_tta["boxes"][0]:
        c = b["box_coords"]
        assert c[0] >= -1 and c[2] <= 101 and c[1] >= -1 and c[3] <= 91 and c[2] >= c[0] and c[3] >= c[1]
    f = predictor.box_patch_center_factor([0, 0, 64, 64, 0, 32], [64, 64, 32])
    assert abs(f - 1.0) < 1e-12                     # box centred in the patch -> factor 1


def _iou3d(a, b):
    lo = np.maximum(a[:, None, [0, 1, 4]], b[None, :, [0, 1, 4]])
    hi = np.minimum(a[:, None, [2, 3, 5]], b[None, :, [2, 3, 5]])
    inter = np.clip(hi - lo, 0, None).prod(-1)
    va = (a[:, [2, 3, 5]] - a[:, [0, 1, 4]]).prod(-1)
    vb = (b[:, [2, 3, 5]] - b[:, [0, 1, 4]]).prod(-1)
    return inter / np.maximum(va[:, None] + vb[None, :] - inter, 1e-12)


def test_bf16_patch_tiled_inference_tracks_fp32(cuda):
    """BASELINE config 5 asks for bf16 (the reference's predictor.py:279-455 is all-fp32): pin what autocast(bf16) does to the
    DETECTIONS of the patch-tiled pipeline.  Name-seeded weights (tests/golden/step_inputs.fill_by_name: logits and deltas O(1), so
    scores spread over (0, 1)), a noise volume with bright ellipsoids, collect_raw_boxes in fp32 and under autocast:
      * same patch count and pass count; detection counts within 5 %;
      * >= 80 % of the fp32 raw detections with score > 0.1 have a bf16 detection of the SAME class in the SAME patch at IoU >= 0.5
        (>= 70 % at IoU >= 0.9), and vice versa (bf16 does not invent detections);
      * the scores of the matched pairs differ by <= 0.02 for 95 % of them and by <= 0.05 for all.
    Measured (round 4, MI355X): 524 vs 526 detections, 87 % / 86 % / 81 % matched at IoU 0.5 / 0.7 / 0.9, score gap p95 0.006, max
    0.028.  The unmatched eighth is the UNTRAINED net: its class scores sit near ties (0.33-0.45 over 3 classes), so a 1e-2 relative
    feature perturbation (bf16: 8 mantissa bits through ~50 conv layers) flips arg-max classes and NMS winners -- whole detections
    change hands -- while a matched detection barely moves.  The bars are stated in DESIGN.md 'numerics'."""
    from medicaldetectiontoolkit_amd import predictor
    from tests.golden import step_inputs as si
    cf = Configs(dim=3, model="mrcnn", patch_size=[64, 64, 32], batch_size=4)
    net = mrcnn.net(cf, device=cuda)
    si.fill_by_name(net)
    net.eval()
    rng = np.random.default_rng(5)
    vol = rng.standard_normal((1, 100, 90, 48)).astype(np.float32)
    yy, xx, zz = np.meshgrid(np.arange(100), np.arange(90), np.arange(48), indexing="ij")
    for c, r in (((30, 30, 16), (9, 8, 4)), ((70, 55, 30), (7, 10, 5)), ((50, 75, 12), (6, 6, 3))):
        vol[0][((yy - c[0]) / r[0]) ** 2 + ((xx - c[1]) / r[1]) ** 2 + ((zz - c[2]) / r[2]) ** 2 <= 1.0] += 2.0
    raw32, info32 = predictor.collect_raw_boxes(net, vol, cf)
    raw16, info16 = predictor.collect_raw_boxes(net, vol, cf, amp_dtype=torch.bfloat16)
    assert info16["n_patches"] == info32["n_patches"] and info16["n_passes"] == info32["n_passes"]

    def table(raw):
        keep = [b for b in raw if b["box_score"] > 0.1]
        return (np.array([b["box_coords"] for b in keep]).reshape(-1, 6), np.array([b["box_score"] for b in keep]),
                np.array([b["box_pred_class_id"] for b in keep]), np.array([b["patch_id"] for b in keep]))

    def matched(a, b, thr):
        """fraction of a's rows with a same-class, same-patch partner in b at IoU >= thr, and the score gaps of the matches"""
        ca, sa, ka, pa = a
        cb, sb, kb, pb = b
        iou = _iou3d(ca, cb)
        iou[(ka[:, None] != kb[None, :]) | (pa[:, None] != pb[None, :])] = 0.0
        j = iou.argmax(1)
        ok = iou[np.arange(len(ca)), j] >= thr
        return float(ok.mean()), np.abs(sa[ok] - sb[j[ok]])

    t32, t16 = table(raw32), table(raw16)
    assert len(t32[0]) >= 20, "the seeded net must produce detections for this test to mean anything (%d)" % len(t32[0])
    stats = {"n32": len(t32[0]), "n16": len(t16[0])}
    for thr in (0.5, 0.7, 0.9):
        f_fwd, gaps = matched(t32, t16, thr)
        f_bwd, _ = matched(t16, t32, thr)
        stats[thr] = (round(f_fwd, 3), round(f_bwd, 3), round(float(np.quantile(gaps, 0.95)), 4) if len(gaps) else None, round(float(gaps.max()), 4) if len(gaps) else None)
    print("bf16 vs fp32 detections:", stats)
    # bars (DESIGN.md 'numerics'; measured on the name-seeded UNTRAINED net, whose integer-rounded boxes of a few voxels flip by one
    # voxel under a 1e-2 relative feature perturbation -- IoU 0.9 is then lost although the detection is the same object):
    assert abs(stats["n32"] - stats["n16"]) <= 0.05 * stats["n32"], stats
    assert stats[0.5][0] >= 0.80 and stats[0.5][1] >= 0.80, stats
    assert stats[0.9][0] >= 0.70 and stats[0.9][1] >= 0.70, stats
    assert stats[0.5][2] <= 0.02 and stats[0.5][3] <= 0.05, stats       # matched pairs: |score(bf16) - score(fp32)| p95 <= 0.02, max <= 0.05


def test_predict_test_set_ensembling_and_raw_pickle(cuda, tmp_path):
    """Temporal ensembling over two saved checkpoints + the reference's raw-prediction pickle format."""
    import pickle
    from medicaldetectiontoolkit_amd import predictor, training
    from medicaldetectiontoolkit_amd.utils import exp_utils
    cf = Configs(dim=3, model="mrcnn", patch_size=[64, 64, 32], batch_size=4)
    torch.manual_seed(0)
    net = mrcnn.net(cf, device=cuda)
    opt = training.build_optimizer(net, cf)
    ck = []
    for ep in (1, 2):
        training.train_step(net, opt, make_batch(cf.patch_size, 2, seed=ep))
        exp_utils.save_best_checkpoint(str(tmp_path), net, ep)
        ck.append(str(tmp_path / ("%d_best_checkpoint" % ep)))
    rng = np.random.default_rng(1)
    patients = [(rng.standard_normal((1, 80, 70, 40)).astype(np.float32), "p%d" % i) for i in range(2)]
    res = predictor.predict_test_set(net, patients, cf, checkpoint_paths=ck, out_dir=str(tmp_path), test_aug=False)
    assert [pid for _, pid in res] == ["p0", "p1"] and all(len(b) == 1 for b, _ in res)
    raw = pickle.load(open(str(tmp_path / "raw_pred_boxes_list.pickle"), "rb"))
    assert len(raw) == 2 and raw[0][1] == "p0"
    members = {b["patch_id"].split("_")[0] for b in raw[0][0][0]}
    assert members <= {"0", "1"}
    for b in raw[0][0][0]:
        assert {"box_coords", "box_score", "box_pred_class_id", "patch_id", "box_patch_center_factor", "box_n_overlaps"} <= set(b)


def test_classifier_head_as_linear_equals_conv_path(cuda):
    """Classifier head: conv1 (kernel = pooled extent) and conv2 (1x1x1) as matrix products == the convolution path, forward
    and parameter gradients (models/mrcnn.py HEAD_AS_LINEAR)"""
    from medicaldetectiontoolkit_amd.configs import Configs
    from medicaldetectiontoolkit_amd.models import mrcnn
    from medicaldetectiontoolkit_amd.utils.model_utils import NDConvGenerator
    cf = Configs(dim=3, model="mrcnn", patch_size=[64, 64, 32], batch_size=2)
    torch.manual_seed(3)
    head = mrcnn.Classifier(cf, NDConvGenerator(3)).to(cuda)
    maps = [torch.randn((2, cf.end_filts) + tuple(int(v) for v in s), device=cuda) for s in cf.backbone_shapes]
    g = torch.Generator(device=cuda).manual_seed(4)
    lo = torch.rand((20, 3), device=cuda, generator=g) * 0.6
    ext = 0.1 + 0.3 * torch.rand((20, 3), device=cuda, generator=g)
    rois = torch.stack([lo[:, 0], lo[:, 1], lo[:, 0] + ext[:, 0], lo[:, 1] + ext[:, 1], lo[:, 2], lo[:, 2] + ext[:, 2],
                        torch.randint(0, 2, (20,), device=cuda, generator=g).float()], 1)
    outs, grads = [], []
    for flag in (True, False):
        mrcnn.HEAD_AS_LINEAR = flag
        head.zero_grad()
        logits, boxes = head(maps, rois)
        (logits.square().sum() + boxes.square().sum()).backward()
        outs.append((logits.detach().clone(), boxes.detach().clone()))
        grads.append([p.grad.detach().clone() for p in head.parameters()])
    mrcnn.HEAD_AS_LINEAR = True
    for a, b in zip(outs[0], outs[1]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), float((a - b).abs().max())
    for a, b in zip(grads[0], grads[1]):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-3, atol=1e-4 * float(b.abs().max()) + 1e-6), float((a - b).abs().max())


@pytest.mark.parametrize("channels_last", [False, True])
def test_rpn_merged_heads_equal_separate_layers(channels_last, cuda):
    """RPN.forward with conv_class / conv_bbox as ONE 1x1x1 convolution over the concatenated filters (models/mrcnn.py MERGE_RPN_HEADS)
    == the two layers of mrcnn.py:70-77: outputs, input gradient and all six parameter gradients (fp32 summation order only)"""
    from medicaldetectiontoolkit_amd.utils.model_utils import NDConvGenerator
    cf = Configs(dim=3, model="mrcnn", patch_size=[64, 64, 32], batch_size=2)
    torch.manual_seed(11)
    rpn = mrcnn.RPN(cf, NDConvGenerator(3)).to(cuda)
    x0 = torch.randn((2, cf.end_filts, 16, 16, 32), device=cuda)
    if channels_last:
        rpn = rpn.to(memory_format=torch.channels_last_3d)
        x0 = x0.contiguous(memory_format=torch.channels_last_3d)
    res = []
    for flag in (True, False):
        mrcnn.MERGE_RPN_HEADS = flag
        rpn.zero_grad()
        x = x0.clone().requires_grad_(True)
        logits, probs, bbox = rpn(x)
        (logits.square().sum() + (probs * probs).sum() + bbox.square().sum()).backward()
        res.append(([logits.detach().clone(), probs.detach().clone(), bbox.detach().clone(), x.grad.clone()],
                    {n: p.grad.clone() for n, p in rpn.named_parameters()}))
    mrcnn.MERGE_RPN_HEADS = True
    for a, b in zip(res[0][0], res[1][0]):
        assert a.shape == b.shape and torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max())), float((a - b).abs().max())
    assert set(res[0][1]) == set(res[1][1]) and len(res[0][1]) == 6
    for n in res[0][1]:
        a, b = res[0][1][n], res[1][1][n]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()) + 1e-7), (n, float((a - b).abs().max()))


@pytest.mark.parametrize("dim,patch", [(3, [64, 64, 32]), (2, [64, 64])])
def test_sparse_rpn_loss_step_equals_dense_graph_step(dim, patch, cuda):
    """train_forward_device with the RPN losses differentiated through the sampled anchors only (models/mrcnn.SPARSE_RPN_LOSS,
    rpn_at_anchors; the dense RPN forward carries no graph) == the same step through the dense RPN graph, as the reference builds it
    (mrcnn.py:870-946): all five loss terms, the sampled anchors, and the gradient of EVERY parameter (fp32 summation order only)"""
    from medicaldetectiontoolkit_amd.utils.synthetic_data import batch_with_gt_from_proposals
    B = 2
    cf = Configs(dim=dim, model="mrcnn", patch_size=patch, batch_size=B, channels_last=True)
    torch.manual_seed(3)
    net = mrcnn.net(cf, device=cuda)
    batch = batch_with_gt_from_proposals(net, cf, make_batch(patch, B, seed=5), cuda)
    res = []
    try:
        for flag in (True, False):
            mrcnn.SPARSE_RPN_LOSS = flag
            net.zero_grad(set_to_none=True)
            torch.manual_seed(17)                                                  # same random sub-sampling keys in both runs
            prep = net.prepare_batch(batch)
            out = net.train_forward_device(prep["img"], prep["gt"], prep["masks"])
            out["loss"].backward()
            res.append(({k: float(v) for k, v in out["terms"].items()}, [t.clone() for t in out["mon"]["rpn_samples"]],
                        {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
    finally:
        mrcnn.SPARSE_RPN_LOSS = True
    (ta, sa, ga), (tb, sb, gb) = res
    assert sum(int(v.sum()) for v in (sa[1], sa[3])) > 0, "no anchor was sampled: the comparison would be empty"
    for k in tb:
        assert abs(ta[k] - tb[k]) <= 1e-5 * max(1.0, abs(tb[k])), (k, ta[k], tb[k])
    for a, b in zip(sa, sb):
        assert torch.equal(a, b)
    assert set(ga) == set(gb) and any(n.startswith("rpn.") for n in ga)
    for n in gb:
        a, b = ga[n], gb[n]
        assert torch.allclose(a, b, rtol=1e-3, atol=2e-5 * float(b.abs().max()) + 1e-8), (n, float((a - b).abs().max()), float(b.abs().max()))


def test_train_py_exec_loop_graphed_then_resumed_eager(cuda, tmp_path):
    """train.py = the stand-in for exec.py's training loop: batches through training.DevicePrefetcher, the monitoring read-out and log line
    every batch, reference-format checkpoint per epoch.  Epoch 1 with the graphed step (--graph 1), then --resume for epoch 2 with the
    eager step: both print one 'tr. batch' line per batch with the five loss terms, the second run resumes at epoch 2."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["MDT_MIOPEN_SKIP_NAIVE"] = "1"
    base = [sys.executable, os.path.join(root, "train.py"), "--patch", "64,64,32", "--batch", "2", "--batches", "3", "--exp-dir", str(tmp_path)]
    r1 = subprocess.run(base + ["--epochs", "1", "--graph", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r1.returncode == 0, r1.stderr[-1500:]
    lines = [l for l in r1.stdout.splitlines() if l.startswith("tr. batch")]
    assert len(lines) == 3 and all("rpn_class" in l and "mrcnn_mask" in l for l in lines), r1.stdout[-800:]
    assert os.path.exists(os.path.join(str(tmp_path), "fold_0", "last_checkpoint", "params.pth"))
    r2 = subprocess.run(base + ["--epochs", "2", "--resume"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r2.returncode == 0, r2.stderr[-1500:]
    assert "resumed to checkpoint at epoch 2" in r2.stdout
    lines = [l for l in r2.stdout.splitlines() if l.startswith("tr. batch")]
    assert len(lines) == 3 and all("(ep. 2)" in l for l in lines), r2.stdout[-800:]


def test_deferred_monitor_readout_equals_synchronous_one_step_late(cuda):
    """train_forward(monitor="deferred") hands out, at call i + 1, the read-out entries of call i (exec.py:76-79 consumes them one batch
    later; no host sync in the step): the loss in `monitor_values` / `logger_string` is bit-for-bit the device loss call i returned, the GT
    boxes are call i's batch, the sampled-anchor boxes equal the synchronous form's on the same seed; flush_deferred_monitor() delivers the
    last one"""
    patch, B = [64, 64, 32], 2
    cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B)
    torch.manual_seed(0)
    net = mrcnn.net(cf, device=cuda)
    batches = [make_batch(patch, B, seed=s) for s in (3, 4, 5)]
    sync = []
    for i, b in enumerate(batches):
        torch.manual_seed(50 + i)
        sync.append(net.train_forward(b, monitor=True))
    got, losses = [], []
    for i, b in enumerate(batches):
        torch.manual_seed(50 + i)
        r = net.train_forward(b, monitor="deferred")
        losses.append(float(r["torch_loss"]))
        assert ("logger_string" in r) == (i > 0)
        if i > 0:
            assert r["monitor_of_previous_step"]
            got.append(r)
    got.append(net.flush_deferred_monitor())
    assert net.flush_deferred_monitor() is None
    assert len(got) == 3
    for i, (s, g) in enumerate(zip(sync, got)):
        assert g["monitor_values"]["loss"] == losses[i]                       # the SAME pass: exact
        assert g["logger_string"].startswith("loss: {0:.2f}".format(losses[i]))
        assert abs(s["monitor_values"]["loss"] - g["monitor_values"]["loss"]) <= 1e-4 * abs(losses[i])     # two passes: MIOpen run-to-run noise
        assert len(s["boxes"]) == len(g["boxes"]) == B
        for b_ix, (bs, bg) in enumerate(zip(s["boxes"], g["boxes"])):
            gt = [x for x in bg if x["box_type"] == "gt"]
            assert len(gt) == len(batches[i]["bb_target"][b_ix])
            for x, want in zip(gt, batches[i]["bb_target"][b_ix]):
                assert np.array_equal(np.asarray(x["box_coords"]), np.asarray(want))
            for t in ("pos_anchor", "neg_anchor", "prop"):
                assert sum(1 for x in bs if x["box_type"] == t) == sum(1 for x in bg if x["box_type"] == t), t


@pytest.mark.parametrize("model,dim,patch", [("retina_unet", 3, [64, 64, 32]), ("retina_net", 2, [64, 64])])
def test_retina_deferred_monitor_readout_equals_synchronous_one_step_late(model, dim, patch, cuda):
    """the Retina nets' train_forward(monitor="deferred") (round 6: what bench.py's exec-form step of config 2 runs): at call i + 1 the entries of
    call i -- loss values of the SAME pass exactly, the GT / sampled-anchor / detection boxes and, for the U-Net, the uint8 label map equal to the
    synchronous form's on the same seed (label voxels may flip only where MIOpen's run-to-run noise meets a tie)"""
    B = 2
    cf = Configs(dim=dim, model=model, patch_size=patch, batch_size=B)
    torch.manual_seed(0)
    net = retina_unet.net(cf, device=cuda)
    batches = [make_batch(patch, B, seed=s) for s in (3, 4, 5)]
    sync = []
    for i, b in enumerate(batches):
        torch.manual_seed(50 + i)
        sync.append(net.train_forward(b, monitor=True))
    got, losses = [], []
    for i, b in enumerate(batches):
        torch.manual_seed(50 + i)
        r = net.train_forward(b, monitor="deferred")
        losses.append(float(r["torch_loss"]))
        assert ("logger_string" in r) == (i > 0)
        if i > 0:
            assert r["monitor_of_previous_step"]
            got.append(r)
    got.append(net.flush_deferred_monitor())
    assert net.flush_deferred_monitor() is None and len(got) == 3
    for i, (s, g) in enumerate(zip(sync, got)):
        assert g["monitor_values"]["loss"] == losses[i]
        assert g["logger_string"].startswith("loss: {0:.2f}".format(losses[i]))
        assert abs(s["monitor_values"]["loss"] - g["monitor_values"]["loss"]) <= 1e-4 * abs(losses[i])
        for b_ix, (bs, bg) in enumerate(zip(s["boxes"], g["boxes"])):
            for t in ("gt", "pos_anchor", "neg_anchor", "det"):
                assert sum(1 for x in bs if x["box_type"] == t) == sum(1 for x in bg if x["box_type"] == t), (i, b_ix, t)
        assert s["seg_preds"].shape == g["seg_preds"].shape
        if model == "retina_unet":
            assert g["seg_preds"].dtype == np.uint8 and float((s["seg_preds"] != g["seg_preds"]).mean()) <= 1e-4
            assert "mean pix. pr." in g["logger_string"]
