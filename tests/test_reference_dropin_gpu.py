"""The reference's OWN model files, UNMODIFIED, on the MI355X with this repo's ops behind their import statements
("behind the existing models/{mrcnn,retina_unet}.py call sites"; VERDICT r4 "What's missing" 1).

oracle/_ref/ref_models.tar.gz holds /root/reference/{models/{mrcnn,retina_unet,backbone}.py, utils/{model_utils,exp_utils}.py,
plotting.py}, packed where they lie by `make -C oracle _ref_py` (run by __graft_entry__.build() in the build container; git-ignored like
the compiled reference objects, shipped to the GPU box with the snapshot; unpacked into a temporary directory by oracle/ref_models.py --
no reference source file lives in this repository's tree).  Here
    medicaldetectiontoolkit_amd.install_dropin()
registers this repo's `cuda_functions` package under the name the reference imports (models/mrcnn.py:24-27,
models/retina_unet.py:26-27), the reference `net` classes are built with `.cuda()` REAL (every tensor of the step lives on the GPU,
every nms_gpu / CropAndResizeFunction call lands in libmdt_hip.so -- counted), and one `train_forward` + `backward()` is compared with
tests/golden/step_reference.npz: the same reference code run on the CPU with the oracle behind the same imports
(tests/golden/make_step_golden.py).  Bars as everywhere: loss terms 1e-4 relative, module gradient norms 1e-3 relative, sampled-set
sizes equal.

Nothing of the reference is edited.  The only things in force are torch-0.4.1 behaviours its code relies on, applied from OUTSIDE as in
tests/golden/make_step_golden.py:30-80: integer `/` on index tensors floor-divides (retina_unet.py:212); `Tensor.__array__` of a CUDA
tensor copies to the host as torch 0.4.1 did (`np.argwhere(cuda_tensor == -1)`, mrcnn.py:915 -- needed only here, where tensors really
are CUDA tensors); and the default `shem_poolsize` of retina_unet.compute_class_loss is 1 instead of 20 for a deterministic sample (the
golden was made that way).
A missing archive is a FAILURE, not a skip."""
import importlib.util
import logging
import os
import sys

import numpy as np
import pytest
import torch

from tests.golden import step_inputs as si

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PY = None          # set by the `ref` fixture: the temporary directory the archive was unpacked into
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "step_reference.npz"), allow_pickle=False)
_FILES = ("models/mrcnn.py", "models/retina_unet.py", "models/backbone.py", "utils/model_utils.py", "utils/exp_utils.py", "plotting.py")


class torch04(object):
    """torch 0.4.1 (requirements.txt:25) behaviours the reference's host code relies on, restored from outside:
    * `long_tensor / int` floor-divides (retina_unet.py:212);
    * `Tensor.__array__` copies a CUDA tensor to the host (torch 0.4.1: `return self.cpu().numpy()`; torch 2 raises): the models hand
      CUDA tensors to numpy -- `np.argwhere(rpn_match == -1)` (mrcnn.py:897,915; retina_unet.py:421,438), `np.unique(batch_ixs)`
      (retina_unet.py:205).  Never reached by the CPU harness of make_step_golden.py, where every tensor is a host tensor."""

    def __enter__(self):
        self._div = torch.Tensor.__truediv__
        self._array = torch.Tensor.__array__
        prev = self._div
        prev_array = self._array

        def array(t, dtype=None):
            return prev_array(t.cpu() if t.is_cuda else t, dtype)
        torch.Tensor.__array__ = array

        def div(a, b):
            if not a.is_floating_point() and not (torch.is_tensor(b) and b.is_floating_point()) and not isinstance(b, float):
                return torch.div(a, b, rounding_mode="floor")
            return prev(a, b)
        torch.Tensor.__truediv__ = div

    def __exit__(self, *exc):
        torch.Tensor.__truediv__ = self._div
        torch.Tensor.__array__ = self._array


class Recorder(object):
    """observes (does not change) what a reference loss helper returns"""

    def __init__(self, mod, name):
        self.mod, self.name, self.fn, self.vals = mod, name, getattr(mod, name), []
        setattr(mod, name, self)

    def __call__(self, *a, **k):
        r = self.fn(*a, **k)
        v = r[0] if isinstance(r, tuple) else r
        self.vals.append(float(v.detach().double().sum()))
        return r

    def undo(self):
        setattr(self.mod, self.name, self.fn)


@pytest.fixture(scope="module")
def ref(cuda):
    """the reference modules, imported from the unpacked oracle/_ref/ref_models.tar.gz with this repo's cuda_functions behind their imports"""
    global REF_PY
    from oracle import ref_models
    try:
        REF_PY = ref_models.unpack(prefer_archive=True)       # what ships to the GPU box, also when the checkout itself is present
    except FileNotFoundError as e:
        pytest.fail(str(e))
    missing = [f for f in _FILES if not os.path.exists(os.path.join(REF_PY, f))]
    assert not missing, missing
    import medicaldetectiontoolkit_amd as m
    from medicaldetectiontoolkit_amd import _lib, miopen_env
    miopen_env.setup()
    _lib.lib()
    m.install_dropin()
    before = set(sys.modules)
    sys.path.insert(0, REF_PY)                 # `import utils.model_utils`, `import utils.exp_utils`, `import plotting` -> the reference's

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_PY, path))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    try:
        mr = load("models/mrcnn.py", "ref_dropin_mrcnn")
        ru = load("models/retina_unet.py", "ref_dropin_retina_unet")
        # what the reference imported IS this repo's op, not a stand-in
        import medicaldetectiontoolkit_amd.cuda_functions.nms_3D.pth_nms as my_nms3
        import medicaldetectiontoolkit_amd.cuda_functions.roi_align_3D.roi_align.crop_and_resize as my_ra3
        assert mr.nms_3D is my_nms3.nms_gpu and mr.ra3D is my_ra3.CropAndResizeFunction and ru.nms_3D is my_nms3.nms_gpu
        assert os.path.realpath(mr.mutils.__file__).startswith(os.path.realpath(REF_PY))
        yield {"mrcnn": mr, "retina_unet": ru, "lib": _lib}
    finally:
        sys.path.remove(REF_PY)
        for k in set(sys.modules) - before:    # `utils`, `plotting`, ... of the reference must not leak into other test modules
            if k.split(".")[0] in ("utils", "plotting", "ref_dropin_mrcnn", "ref_dropin_retina_unet"):
                sys.modules.pop(k, None)


def _batch():
    gt_boxes = [GOLD["gt_boxes_%d" % b] for b in range(si.CASES["small"][1])]
    gt_labels = [GOLD["gt_labels_%d" % b] for b in range(si.CASES["small"][1])]
    return si.make_batch(si.make_image(), gt_boxes, gt_labels)


def _grad_norms(net):
    mods = {}
    for name, p in net.named_parameters():
        mods.setdefault(si.module_of(name), []).append(0.0 if p.grad is None else float((p.grad.double() ** 2).sum()))
    return {k: float(np.sqrt(sum(v))) for k, v in mods.items()}


def _close(got, want, rel, what):
    assert abs(got - want) <= rel * abs(want) + 1e-6, "%s: got %.8g, reference-on-CPU %.8g (rel %.2e)" % (
        what, got, want, abs(got - want) / max(abs(want), 1e-30))


def _log():
    log = logging.getLogger("ref_dropin")
    log.addHandler(logging.NullHandler())
    return log


def test_reference_mrcnn_file_trains_on_the_hip_ops(ref, cuda):
    """/root/reference/models/mrcnn.py `net` (:801-1082), `.cuda()` real: proposal_layer -> nms_3D (:345), pyramid_roi_align -> ra3D (:441),
    detection_target_layer -> ra3D on the GT masks (:558), backward through CropAndResizeFunction"""
    mr, _lib = ref["mrcnn"], ref["lib"]
    nb = si.CASES["small"][1]
    cf = si.make_cf("mrcnn")
    cf.backbone_path = os.path.join(REF_PY, "models/backbone.py")
    net = mr.net(cf, _log()).cuda()
    si.fill_by_name(net)
    rec = {k: Recorder(mr, "compute_" + k + "_loss") for k in ("rpn_class", "rpn_bbox", "mrcnn_class", "mrcnn_bbox", "mrcnn_mask")}
    np.random.seed(0)
    torch.manual_seed(0)
    _lib.count_calls(True)
    try:
        with torch04():
            res = net.train_forward(_batch())
        net.zero_grad()
        res["torch_loss"].backward()
        torch.cuda.synchronize()
        calls = dict(_lib.CALLS)
    finally:
        _lib.count_calls(False)
        for r in rec.values():
            r.undo()
    assert res["torch_loss"].is_cuda and next(net.parameters()).is_cuda
    # the reference's own call sites reached libmdt_hip.so: one NMS per batch element (mrcnn.py:317-350), RoIAlign forward on >= 1 level for
    # the classifier (7,7,3) and the mask head (14,14,5) + one per element with positives on the GT masks (:558), backward for both heads
    assert calls.get("mdt_nms_3d", 0) >= nb, calls
    assert calls.get("mdt_crop_and_resize_3d_forward", 0) >= 2 + 1, calls
    assert calls.get("mdt_crop_and_resize_3d_backward", 0) >= 2, calls
    for k, r in rec.items():
        got = sum(r.vals) / (nb if k.startswith("rpn") else 1)
        _close(got, float(GOLD["mrcnn_term_" + k]), 1e-4, "reference mrcnn.py on HIP ops: " + k)
    _close(float(res["torch_loss"].item()), float(GOLD["mrcnn_loss"]), 1e-4, "reference mrcnn.py on HIP ops: total loss")
    boxes = [bx for bl in res["boxes"] for bx in bl]
    assert [sum(1 for bx in boxes if bx["box_type"] == t) for t in ("pos_class", "neg_class")] == GOLD["mrcnn_n_pos_neg_rois"].tolist()
    assert [sum(1 for bx in boxes if bx["box_type"] == t) for t in ("pos_anchor", "neg_anchor")] == GOLD["mrcnn_n_pos_neg_anchors"].tolist()
    for k, v in _grad_norms(net).items():
        _close(v, float(GOLD["mrcnn_gradnorm_" + k]), 1e-3, "reference mrcnn.py on HIP ops: grad norm of " + k)


def test_reference_retina_unet_file_trains_on_the_hip_ops(ref, cuda):
    """/root/reference/models/retina_unet.py `net` (:338-513), `.cuda()` real: refine_detections -> nms_3D (:248-250) per batch element"""
    ru, _lib = ref["retina_unet"], ref["lib"]
    nb = si.CASES["small"][1]
    cf = si.make_cf("retina_unet")
    cf.backbone_path = os.path.join(REF_PY, "models/backbone.py")
    old_defaults = ru.compute_class_loss.__defaults__
    ru.compute_class_loss.__defaults__ = (1,)          # shem_poolsize default 20 -> 1, as in make_step_golden.py
    net = ru.net(cf, _log()).cuda()
    si.fill_by_name(net)
    rec = {"class": Recorder(ru, "compute_class_loss"), "bbox": Recorder(ru, "compute_bbox_loss")}
    dice = Recorder(ru.mutils, "batch_dice")
    np.random.seed(0)
    torch.manual_seed(0)
    _lib.count_calls(True)
    try:
        with torch04():
            res = net.train_forward(_batch())
        net.zero_grad()
        res["torch_loss"].backward()
        torch.cuda.synchronize()
        calls = dict(_lib.CALLS)
    finally:
        _lib.count_calls(False)
        ru.compute_class_loss.__defaults__ = old_defaults
        for r in list(rec.values()) + [dice]:
            r.undo()
    assert res["torch_loss"].is_cuda
    assert calls.get("mdt_nms_3d", 0) >= 1, calls
    terms = {k: sum(r.vals) / nb for k, r in rec.items()}
    terms["seg_dice"] = 1.0 - dice.vals[0]
    terms["seg_ce"] = 2.0 * (float(res["torch_loss"].item()) - terms["class"] - terms["bbox"]) - terms["seg_dice"]
    for k in ("class", "bbox", "seg_dice"):
        _close(terms[k], float(GOLD["retina_term_" + k]), 1e-4, "reference retina_unet.py on HIP ops: " + k)
    _close(terms["seg_ce"], float(GOLD["retina_term_seg_ce"]), 5e-4, "reference retina_unet.py on HIP ops: seg_ce (derived by subtraction)")
    _close(float(res["torch_loss"].item()), float(GOLD["retina_loss"]), 1e-4, "reference retina_unet.py on HIP ops: total loss")
    boxes = [bx for bl in res["boxes"] for bx in bl]
    assert [sum(1 for bx in boxes if bx["box_type"] == t) for t in ("pos_anchor", "neg_anchor")] == GOLD["retina_n_pos_neg_anchors"].tolist()
    for k, v in _grad_norms(net).items():
        _close(v, float(GOLD["retina_gradnorm_" + k]), 1e-3, "reference retina_unet.py on HIP ops: grad norm of " + k)


def test_reference_adam_leaves_the_same_parameters_alone_as_flat_adam_in_a_step_without_positives(ref, cuda):
    """exec.py:39,72-74 on the reference's own mrcnn.py vs this repo's step + training.FlatAdam, three steps: full batch, a batch WITHOUT GT
    objects, full batch.  In the empty step the reference's mask / bbox / rpn-bbox losses are constants (mrcnn.py:233-234, 266-268, 287-288):
    torch.optim.Adam leaves those heads bit-for-bit alone and does not count a step for them.  FlatAdam must skip EXACTLY the same
    parameters (by state-dict name) -- it decides on the device from the counts the step wrote (net.grad_condition_spec) -- and end with
    the same per-parameter step counters."""
    from medicaldetectiontoolkit_amd import training
    from medicaldetectiontoolkit_amd.models import mrcnn as my_mrcnn
    mr = ref["mrcnn"]
    nb = si.CASES["small"][1]
    full = _batch()
    empty = si.make_batch(si.make_image(seed=32), [np.zeros((0, 6), np.float32)] * nb, [np.zeros((0,), np.int64)] * nb)
    cf = si.make_cf("mrcnn")
    cf.backbone_path = os.path.join(REF_PY, "models/backbone.py")
    rnet = mr.net(cf, _log()).cuda()
    si.fill_by_name(rnet)
    ropt = torch.optim.Adam(rnet.parameters(), lr=cf.learning_rate[0], weight_decay=cf.weight_decay)      # exec.py:39
    mnet = my_mrcnn.net(si.make_cf("mrcnn"), device=cuda)
    si.fill_by_name(mnet)
    mopt = training.build_optimizer(mnet, mnet.cf, flat=True)
    untouched = {"ref": [], "mine": []}
    for it, batch in enumerate((full, empty, full)):
        np.random.seed(it)
        torch.manual_seed(it)
        before = {n: p.detach().clone() for n, p in rnet.named_parameters()}
        with torch04():
            res = rnet.train_forward(batch)
        ropt.zero_grad()                       # exec.py:72 (torch >= 2: set_to_none)
        res["torch_loss"].backward()           # :73
        ropt.step()                            # :74
        untouched["ref"].append({n for n, p in rnet.named_parameters() if torch.equal(p, before[n])})
        before = {n: p.detach().clone() for n, p in mnet.named_parameters()}
        training.train_step(mnet, mopt, batch, monitor=False)
        untouched["mine"].append({n for n, p in mnet.named_parameters() if torch.equal(p, before[n])})
    for it in range(3):
        assert untouched["ref"][it] == untouched["mine"][it], (it, sorted(untouched["ref"][it] ^ untouched["mine"][it]))
    heads = {n for n, _ in rnet.named_parameters() if n.startswith("mask.") or n.startswith("classifier.linear_bbox") or n.startswith("rpn.conv_bbox")}
    # step 1 (positives exist) moves the heads, the empty step 2 leaves them alone (step 3 may or may not find positives again: the weights
    # have moved -- whatever it does, reference and FlatAdam did the same, asserted above)
    assert heads and heads <= untouched["ref"][1] and not (heads & untouched["ref"][0])
    rs, ms = ropt.state_dict()["state"], mopt.state_dict()["state"]
    names = [n for n, _ in rnet.named_parameters()]
    for k, n in enumerate(names):
        assert float(ms[k]["step"]) == (float(rs[k]["step"]) if k in rs else 0.0), n
    assert max(float(rs[k]["step"]) for k, n in enumerate(names) if n in heads) < 3.0 == max(float(v["step"]) for v in rs.values())


# ------------------------------------------------------------------------------------------------------------------------------
# Assembled INFERENCE forward (BASELINE config 5's own path; VERDICT r5 "What's missing" 1): the reference's `net.test_forward` --
# forward -> refine_detections -> mask head -> get_results incl. the mask unmolding into `seg_preds` (mrcnn.py:969-1050, 620-714, 717-799;
# retina_unet.py:459-520, 187-272, 275-338) -- UNMODIFIED on the HIP ops, against this repo's net with the same name-seeded weights.
# Bars: equal detection counts per element, integer box coordinates equal (the cast of mrcnn.py:752 happens after
# torch.round: a difference would be >= 1 px, none is allowed), class ids equal, scores <= 1e-5, seg_preds voxel-for-voxel -- a voxel may
# differ only where the reference's own un-rounded value is a tie (|value - 0.5| <= 1e-4 for the pasted masks; top-2 soft-max margin
# <= 1e-4 for the Retina U-Net arg-max), which the test counts and bounds.
# ------------------------------------------------------------------------------------------------------------------------------
INFER_CASES = {"small": ([64, 64, 32], 2), "bench_patch": ([128, 128, 128], 1)}


def _infer_cf(model, case, backbone_path=None):
    from medicaldetectiontoolkit_amd.configs import Configs
    patch, nb = INFER_CASES[case]
    cf = Configs(dim=3, model=model, patch_size=list(patch), batch_size=nb)
    if backbone_path is not None:
        cf.backbone_path = backbone_path
    return cf


def _infer_image(case, seed=57):
    patch, nb = INFER_CASES[case]
    return np.random.default_rng(seed).standard_normal([nb, 1] + list(patch)).astype(np.float32)


def _det_rows(res):
    """per element: detections as sorted rows (class id, coords..., score) -- the reference emits them in index order of its repeated
    class-major arrays (mrcnn.py:704-711 unique1d), this repo in score order; consumers (predictor.py:458-510, WBC) do not depend on it"""
    out = []
    for bl in res["boxes"]:
        rows = [[int(b["box_pred_class_id"])] + [int(v) for v in b["box_coords"]] + [float(b["box_score"])] for b in bl if b["box_type"] == "det"]
        out.append(sorted(rows, key=lambda r: (r[0], tuple(r[1:-1]), -r[-1])))
    return out


def _compare_detections(rres, mres, what, min_total):
    rr, mm = _det_rows(rres), _det_rows(mres)
    assert [len(r) for r in rr] == [len(m) for m in mm], "%s: detection counts per element differ: reference %s, this repo %s" % (
        what, [len(r) for r in rr], [len(m) for m in mm])
    assert sum(len(r) for r in rr) >= min_total, "%s: the case must produce detections to compare (got %d)" % (what, sum(len(r) for r in rr))
    worst = 0.0
    for e, (re_, me_) in enumerate(zip(rr, mm)):
        for r, m in zip(re_, me_):
            assert r[:-1] == m[:-1], "%s: element %d: class id / integer box differ: reference %s, this repo %s" % (what, e, r, m)
            worst = max(worst, abs(r[-1] - m[-1]))
    assert worst <= 1e-5, "%s: scores differ by %.3g" % (what, worst)
    return sum(len(r) for r in rr), worst


@pytest.mark.parametrize("case", ["small", "bench_patch"])
def test_reference_mrcnn_test_forward_equals_this_repos_test_forward(ref, cuda, case):
    """reference mrcnn.py `net.test_forward(batch, return_masks=True)` (:969-985) vs models/mrcnn.py `net.test_forward`, same weights.
    The reference calls `self.forward(img)` with the DEFAULT is_training=True (:982): its inference uses post_nms_rois_training proposals
    (75 in 3D), never post_nms_rois_inference -- this repo's test_forward does the same."""
    from medicaldetectiontoolkit_amd.models import mrcnn as my_mrcnn
    mr, _lib = ref["mrcnn"], ref["lib"]
    cf = _infer_cf("mrcnn", case, os.path.join(REF_PY, "models/backbone.py"))
    rnet = mr.net(cf, _log()).cuda().eval()
    si.fill_by_name(rnet)
    mnet = my_mrcnn.net(_infer_cf("mrcnn", case), device=cuda).eval()
    si.fill_by_name(mnet)
    batch = {"data": _infer_image(case)}
    unmolded = []
    orig_unmold = mr.mutils.unmold_mask_3D

    def spy(mask, bbox, image_shape):
        full = orig_unmold(mask, bbox, image_shape)
        unmolded.append(full)
        return full
    mr.mutils.unmold_mask_3D = spy
    _lib.count_calls(True)
    try:
        with torch04(), torch.no_grad():
            rres = rnet.test_forward(batch, return_masks=True)
        ref_calls = dict(_lib.CALLS)
    finally:
        _lib.count_calls(False)
        mr.mutils.unmold_mask_3D = orig_unmold
    _lib.count_calls(True)
    try:
        mres = mnet.test_forward(batch, return_masks=True)
        my_calls = dict(_lib.CALLS)
    finally:
        _lib.count_calls(False)
    nb = INFER_CASES[case][1]
    assert ref_calls.get("mdt_nms_3d", 0) >= nb and ref_calls.get("mdt_crop_and_resize_3d_forward", 0) >= 2, ref_calls
    assert sum(v for k, v in my_calls.items() if "roi_align" in k or "crop_and_resize" in k) >= 2 and sum(
        v for k, v in my_calls.items() if "nms" in k) >= 2, my_calls
    n, worst = _compare_detections(rres, mres, "mrcnn test_forward [%s]" % case, min_total=nb)
    # seg_preds: max over the pasted, zoomed masks, rounded (mrcnn.py:768-797)
    rs, ms = rres["seg_preds"], mres["seg_preds"]
    assert rs.shape == ms.shape == (nb, 1) + tuple(INFER_CASES[case][0]) and rs.dtype == ms.dtype == np.uint8
    assert rs.any(), "the case must paste masks"
    counts = [sum(1 for b in bl if b["box_type"] == "det") for bl in rres["boxes"]]
    assert sum(counts) == len(unmolded)
    diff = rs != ms
    off = 0
    for e, c in enumerate(counts):
        if c:
            pre = np.max(np.array(unmolded[off:off + c]), 0)
            off += c
            bad = diff[e, 0] & (np.abs(pre - 0.5) > 1e-4)
            assert not bad.any(), "mrcnn test_forward [%s]: element %d: %d seg_preds voxels differ away from a rounding tie" % (case, e, int(bad.sum()))
        else:
            assert not diff[e].any()
    assert int(diff.sum()) <= max(4, int(1e-5 * diff.size)), "%d of %d seg_preds voxels differ (ties)" % (int(diff.sum()), diff.size)


@pytest.mark.parametrize("case", ["small", "bench_patch"])
def test_reference_retina_unet_test_forward_equals_this_repos_test_forward(ref, cuda, case):
    """reference retina_unet.py `net.test_forward` (:459-475): forward (:478-513) -> refine_detections (:187-272, nms_3D per element and class)
    -> get_results (:275-338, seg_preds = arg-max of the soft-max over the full-resolution segmentation logits)"""
    from medicaldetectiontoolkit_amd.models import retina_unet as my_ru
    ru, _lib = ref["retina_unet"], ref["lib"]
    cf = _infer_cf("retina_unet", case, os.path.join(REF_PY, "models/backbone.py"))
    rnet = ru.net(cf, _log()).cuda().eval()
    si.fill_by_name(rnet)
    mnet = my_ru.net(_infer_cf("retina_unet", case), device=cuda).eval()
    si.fill_by_name(mnet)
    batch = {"data": _infer_image(case)}
    seg_logits = []
    orig = ru.get_results

    def spy(cf_, shape, detections, logits, *a, **k):
        seg_logits.append(logits.detach().float().cpu().numpy())
        return orig(cf_, shape, detections, logits, *a, **k)
    ru.get_results = spy
    _lib.count_calls(True)
    try:
        with torch04(), torch.no_grad():
            rres = rnet.test_forward(batch)
        ref_calls = dict(_lib.CALLS)
    finally:
        _lib.count_calls(False)
        ru.get_results = orig
    mres = mnet.test_forward(batch)
    nb = INFER_CASES[case][1]
    assert ref_calls.get("mdt_nms_3d", 0) >= nb, ref_calls
    _compare_detections(rres, mres, "retina_unet test_forward [%s]" % case, min_total=nb)
    rs, ms = rres["seg_preds"], mres["seg_preds"]
    assert rs.shape == ms.shape == (nb, 1) + tuple(INFER_CASES[case][0]) and rs.dtype == ms.dtype == np.uint8
    assert len(np.unique(rs)) >= 2, "the case must produce a non-trivial label map"
    diff = (rs != ms)[:, 0]
    if diff.any():
        lg = np.sort(seg_logits[0], axis=1)                    # the soft-max is monotone: the arg-max margin is the logit margin
        margin = lg[:, -1] - lg[:, -2]
        assert not (diff & (margin > 1e-4)).any(), "retina_unet test_forward [%s]: %d label voxels differ away from a tie" % (
            case, int((diff & (margin > 1e-4)).sum()))
    assert int(diff.sum()) <= max(4, int(1e-5 * diff.size)), "%d of %d seg_preds voxels differ (ties)" % (int(diff.sum()), diff.size)


# ------------------------------------------------------------------------------------------------------------------------------
# Metric-level parity FROM A REFERENCE-FORMAT CHECKPOINT (VERDICT r5 missing 3 / next 7): the reference's own net, trained
# K steps on the GPU (drop-in ops) with torch.optim.Adam as exec.py:39,68-74 does, is saved BY THE REFERENCE'S OWN WRITER
# (utils/exp_utils.py:147-192 ModelSelector.run_model_selection: `<epoch>_best_checkpoint/params.pth` = bare state dict,
# `last_checkpoint/params.pth` = {epoch, state_dict, optimizer}); this repo's net + FlatAdam load both files; patch-tiled prediction of two
# synthetic volumes through this repo's predictor with either net behind it; then weighted box clustering and the ROI AP of evaluator.py.
# ------------------------------------------------------------------------------------------------------------------------------
class _RefNetAdapter(object):
    """what predictor.collect_raw_boxes needs from a net (`device_`, `test_forward(batch)`), over the reference's net: the patch batch goes in
    as the numpy array its test_forward expects (mrcnn.py:980-981)"""

    def __init__(self, rnet, dev):
        self.rnet, self.device_ = rnet, dev

    def test_forward(self, batch, return_masks=False):
        data = batch["data"]
        with torch04(), torch.no_grad():
            return self.rnet.test_forward({"data": data.detach().cpu().numpy() if torch.is_tensor(data) else data}, return_masks=return_masks)


def test_reference_checkpoint_loads_and_gives_the_same_detections_wbc_and_ap(ref, cuda, tmp_path):
    import types
    from medicaldetectiontoolkit_amd import evaluator, predictor, training
    from medicaldetectiontoolkit_amd.models import mrcnn as my_mrcnn
    from medicaldetectiontoolkit_amd.utils import exp_utils as my_exp
    from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch
    mr = ref["mrcnn"]
    ref_exp = sys.modules["utils.exp_utils"]
    assert os.path.realpath(ref_exp.__file__).startswith(os.path.realpath(REF_PY))
    patch, nb, K = [64, 64, 32], 2, 20
    cf = si.make_cf("mrcnn")
    cf.backbone_path = os.path.join(REF_PY, "models/backbone.py")
    rnet = mr.net(cf, _log()).cuda()
    si.fill_by_name(rnet)
    ropt = torch.optim.Adam(rnet.parameters(), lr=cf.learning_rate[0], weight_decay=cf.weight_decay)      # exec.py:39
    for it in range(K):                                 # exec.py:68-74
        np.random.seed(it)
        torch.manual_seed(it)
        with torch04():
            res = rnet.train_forward(make_batch(patch, nb, seed=200 + it))
        ropt.zero_grad()
        res["torch_loss"].backward()
        ropt.step()
    assert np.isfinite(float(res["torch_loss"].item()))
    # ---- the reference's own checkpoint writer
    cf.fold_dir = str(tmp_path)
    cf.save_n_models, cf.min_save_thresh, cf.model_selection_criteria = 1, 0, ["malignant_ap"]
    selector = ref_exp.ModelSelector(cf, _log())
    selector.run_model_selection(rnet, ropt, {"val": {"malignant_ap": [None, 0.5]}, "train": {"malignant_ap": [None, 0.4]}}, 1)
    best, last = os.path.join(cf.fold_dir, "1_best_checkpoint"), os.path.join(cf.fold_dir, "last_checkpoint")
    assert os.path.exists(os.path.join(best, "params.pth")) and os.path.exists(os.path.join(last, "params.pth"))
    # ---- loaded by this repo: bare state dict into the net; {epoch, state_dict, optimizer} into net + FlatAdam
    mcf = si.make_cf("mrcnn")
    mnet = my_mrcnn.net(mcf, device=cuda)
    start, _ = my_exp.load_checkpoint(best, mnet, map_location="cpu")
    assert start == 1
    for (n1, p1), (n2, p2) in zip(rnet.state_dict().items(), mnet.state_dict().items()):
        assert n1 == n2 and torch.equal(p1.detach().cpu(), p2.detach().cpu()), n1
    mnet2 = my_mrcnn.net(si.make_cf("mrcnn"), device=cuda)
    mopt2 = training.build_optimizer(mnet2, mnet2.cf, flat=True)
    start, metrics = my_exp.load_checkpoint(last, mnet2, mopt2, map_location="cpu")
    assert start == 2 and metrics["val"]["malignant_ap"][1] == 0.5
    rs, ms = ropt.state_dict()["state"], mopt2.state_dict()["state"]
    assert set(rs) == set(ms)
    for k in rs:
        assert float(rs[k]["step"]) == float(ms[k]["step"]) and torch.equal(rs[k]["exp_avg"].cpu(), ms[k]["exp_avg"].cpu())
    # one more step from the loaded optimizer state == one more reference step (FlatAdam adopts per-parameter counters and moments)
    b = make_batch(patch, nb, seed=300)
    np.random.seed(77)
    torch.manual_seed(77)
    with torch04():
        res = rnet.train_forward(b)
    ropt.zero_grad()
    res["torch_loss"].backward()
    ropt.step()
    training.train_step(mnet2, mopt2, b, monitor=False)
    worst = max(float((p1.detach() - p2.detach()).abs().max() / (p1.detach().abs().max() + 1e-12))
                for (_, p1), (_, p2) in zip(rnet.state_dict().items(), mnet2.state_dict().items()))
    assert worst <= 2e-3, "one step after loading the reference checkpoint: parameters differ by %.3g (relative to the tensor's max)" % worst
    rnet.load_state_dict(torch.load(os.path.join(best, "params.pth")))        # back to the checkpointed weights for the inference comparison
    # ---- patch-tiled inference of two synthetic volumes with either net behind this repo's predictor (predictor.py:279-455 semantics)
    rnet.eval()
    mnet.eval()
    mcf.class_dict = {1: "benign", 2: "malignant"}
    mcf.ap_match_ious = [0.1]
    results = {"ref": [], "mine": []}
    n_raw = 0
    for v in range(2):
        vol = np.random.default_rng(900 + v).standard_normal((1, 96, 96, 48)).astype(np.float32)
        raw_r, info_r = predictor.collect_raw_boxes(_RefNetAdapter(rnet, cuda), vol, mcf)
        raw_m, info_m = predictor.collect_raw_boxes(mnet, vol, mcf)
        assert info_r["n_patches"] == info_m["n_patches"] >= 8
        key = lambda bx: (bx["patch_id"], bx["box_pred_class_id"], tuple(np.round(bx["box_coords"]).astype(int)))
        raw_r, raw_m = sorted(raw_r, key=key), sorted(raw_m, key=key)
        assert len(raw_r) == len(raw_m) and len(raw_r) > 0, (len(raw_r), len(raw_m))
        n_raw += len(raw_r)
        for a, c in zip(raw_r, raw_m):
            assert a["patch_id"] == c["patch_id"] and a["box_pred_class_id"] == c["box_pred_class_id"]
            assert np.abs(np.asarray(a["box_coords"]) - np.asarray(c["box_coords"])).max() <= 1e-4
            assert abs(a["box_score"] - c["box_score"]) <= 1e-5
            assert abs(a["box_patch_center_factor"] - c["box_patch_center_factor"]) <= 1e-9 and a["box_n_overlaps"] == c["box_n_overlaps"]
        wbc_r = predictor.apply_wbc_to_patient(raw_r, mcf, info_r["n_passes"], device=cuda)
        wbc_m = predictor.apply_wbc_to_patient(raw_m, mcf, info_m["n_passes"], device=cuda)
        assert len(wbc_r) == len(wbc_m) > 0
        for a, c in zip(wbc_r, wbc_m):
            assert a["box_pred_class_id"] == c["box_pred_class_id"]
            assert np.abs(np.asarray(a["box_coords"]) - np.asarray(c["box_coords"])).max() <= 1e-3 and abs(a["box_score"] - c["box_score"]) <= 1e-5
        # ground truth for the AP: the two best consolidated boxes of the reference's prediction become GT objects of their class, a third GT
        # object sits where nothing was predicted -- tp, fp and fn all occur
        top = sorted(wbc_r, key=lambda bx: -bx["box_score"])[:2]
        gts = [{"box_type": "gt", "box_coords": np.asarray(t["box_coords"]), "box_label": t["box_pred_class_id"]} for t in top]
        gts.append({"box_type": "gt", "box_coords": np.array([1.0, 1.0, 4.0, 4.0, 1.0, 3.0]), "box_label": 2})
        results["ref"].append([[wbc_r + gts], "vol_%d" % v])
        results["mine"].append([[wbc_m + gts], "vol_%d" % v])
    ecf = types.SimpleNamespace(ap_match_ious=[0.1], class_dict={1: "benign", 2: "malignant"}, fold=0)
    df_r, df_m = evaluator.evaluate_predictions(results["ref"], ecf, "test"), evaluator.evaluate_predictions(results["mine"], ecf, "test")
    assert list(df_r.det_type) == list(df_m.det_type) and list(df_r.class_label) == list(df_m.class_label)
    aps = {}
    for cl in (1, 2):
        a_r = evaluator.get_roi_ap_from_df(df_r[df_r.pred_class == cl], 0.1, False)
        a_m = evaluator.get_roi_ap_from_df(df_m[df_m.pred_class == cl], 0.1, False)
        # (a class without a GT object in either volume has no AP: the mean over an empty list, NaN, in both)
        assert a_r == a_m or (np.isnan(a_r) and np.isnan(a_m)), (cl, a_r, a_m)
        aps[cl] = a_r
    assert n_raw >= 16 and any(0.0 < v <= 1.0 for v in aps.values() if not np.isnan(v)), aps
