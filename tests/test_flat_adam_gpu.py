"""GPU: training.FlatAdam (one launch of csrc/adam.hip over flat parameter / gradient / moment buffers) == torch.optim.Adam
(exec.py:39), incl. channels-last weights, weight decay, state_dict round trips and the training step of the real model."""
import copy

import pytest
import torch
import torch.nn as nn

from medicaldetectiontoolkit_amd import training
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch

pytestmark = pytest.mark.gpu


def _toy(cuda, seed=0):
    torch.manual_seed(seed)
    net = nn.Sequential(nn.Conv3d(2, 5, 3, padding=1), nn.ReLU(), nn.Conv3d(5, 3, 1), nn.Flatten(), nn.Linear(3 * 4 * 4 * 4, 6)).to(cuda)
    net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last_3d)     # dense, permuted strides
    return net


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adam_equals_torch_adam(wd, cuda):
    """12 steps on a toy net (one channels-last weight, odd total size: the scalar tail of the kernel runs): parameters equal to
    torch.optim.Adam's to 1e-6 relative, state_dict in torch's format"""
    a, b = _toy(cuda), _toy(cuda)
    assert sum(p.numel() for p in a.parameters()) % 4 != 0
    oa = torch.optim.Adam(a.parameters(), lr=1e-2, weight_decay=wd)
    ob = training.FlatAdam(b.parameters(), lr=1e-2, weight_decay=wd)
    g = torch.Generator(device=cuda).manual_seed(1)
    for it in range(12):
        x = torch.randn((3, 2, 4, 4, 4), device=cuda, generator=g)
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            net(x).square().mean().backward()
            opt.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert pb.stride() == pa.stride()
            assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6), (it, float((pa - pb).abs().max()))
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0]["params"] == sb["param_groups"][0]["params"]
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 12.0
        # the two nets' GRADIENTS agree only to the run-to-run noise of MIOpen's atomics-based weight-gradient solvers (one ulp of the
        # largest terms): an entry of the running mean that nearly cancels sees that noise amplified, so the absolute bar scales with
        # the tensor (1e-6 of its max), on top of the 5e-5 relative one
        ea, eb = sa["state"][k]["exp_avg"], sb["state"][k]["exp_avg"]
        assert torch.allclose(ea, eb, rtol=5e-5, atol=1e-6 * float(ea.abs().max()) + 1e-12), float((ea - eb).abs().max())
        va, vb = sa["state"][k]["exp_avg_sq"], sb["state"][k]["exp_avg_sq"]
        assert torch.allclose(va, vb, rtol=5e-5, atol=1e-6 * float(va.abs().max()) + 1e-15), float((va - vb).abs().max())


def test_flat_adam_state_dict_round_trip_and_adoption(cuda):
    """a torch.optim.Adam checkpoint loads into FlatAdam (and back): training continues on the same trajectory"""
    a, b = _toy(cuda, 3), _toy(cuda, 3)
    oa = torch.optim.Adam(a.parameters(), lr=1e-2)
    g = torch.Generator(device=cuda).manual_seed(2)
    xs = [torch.randn((3, 2, 4, 4, 4), device=cuda, generator=g) for _ in range(8)]
    for x in xs[:4]:
        oa.zero_grad()
        a(x).square().mean().backward()
        oa.step()
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    ob = training.FlatAdam(b.parameters(), lr=1e-2)
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))
    for x in xs[4:]:
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            net(x).square().mean().backward()
            opt.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6)
    assert float(ob.state_dict()["state"][0]["step"]) == 8.0
    oc = torch.optim.Adam(a.parameters(), lr=1e-2)
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))            # and back into torch's Adam


def test_flat_adam_missing_gradient_is_skipped_like_torch_or_zero_like_torch_0_4(cuda):
    """a parameter without a gradient in a step (p.grad None): absent_grad="skip" (default) leaves it alone like torch.optim.Adam --
    parameters, moments and its own step counter; absent_grad="zero_after_first" updates it with a zero gradient like torch 0.4.1
    (whose zero_grad() leaves zero tensors) -- equal to torch.optim.Adam fed explicit zeros"""
    a, b, c, d = _toy(cuda, 5), _toy(cuda, 5), _toy(cuda, 5), _toy(cuda, 5)
    oa = torch.optim.Adam(a.parameters(), lr=1e-2)
    ob = training.FlatAdam(b.parameters(), lr=1e-2)
    oc = torch.optim.Adam(c.parameters(), lr=1e-2)
    od = training.FlatAdam(d.parameters(), lr=1e-2, absent_grad="zero_after_first")
    x = torch.randn((2, 2, 4, 4, 4), device=cuda)
    for it in range(5):
        a.zero_grad()
        a(x).square().mean().backward()
        grads = [p.grad.clone() for p in a.parameters()]
        for net, opt in ((b, ob), (c, oc), (d, od)):
            opt.zero_grad()
            for p, g in zip(net.parameters(), grads):          # the SAME gradient tensors for all four: the comparison is the optimizer's
                p.grad = g.clone()
        if it in (2, 3):                 # the last layer gets no gradient in steps 2 and 3
            for net in (a, b, d):
                for p in net[4].parameters():
                    p.grad = None
            for p in c[4].parameters():
                p.grad = torch.zeros_like(p)
        for o in (oa, ob, oc, od):
            o.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.equal(pa, pb), (it, float((pa - pb).abs().max()))
        for pc, pd in zip(c.parameters(), d.parameters()):
            assert torch.equal(pc, pd), (it, float((pc - pd).abs().max()))
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    assert [float(sa[k]["step"]) for k in sorted(sa)] == [float(sb[k]["step"]) for k in sorted(sb)] == [5.0, 5.0, 5.0, 5.0, 3.0, 3.0]
    for k in sa:
        assert torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) and torch.equal(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"])


def test_flat_adam_is_bit_equal_to_torch_adam_on_identical_gradients(cuda):
    """arith = 1 (default): the fma pattern of torch's foreach Adam kernels -- parameters and both moments BIT-EQUAL to
    torch.optim.Adam over 12 steps when both are fed the same gradient tensors (with and without weight decay)"""
    for wd in (0.0, 0.01):
        a, b = _toy(cuda, 7), _toy(cuda, 7)
        oa = torch.optim.Adam(a.parameters(), lr=1e-2, weight_decay=wd)
        ob = training.FlatAdam(b.parameters(), lr=1e-2, weight_decay=wd)
        g = torch.Generator(device=cuda).manual_seed(3)
        for it in range(12):
            x = torch.randn((3, 2, 4, 4, 4), device=cuda, generator=g)
            a.zero_grad()
            a(x).square().mean().backward()
            ob.zero_grad()
            for pa, pb in zip(a.parameters(), b.parameters()):
                pb.grad = pa.grad.clone()
            oa.step()
            ob.step()
            for pa, pb in zip(a.parameters(), b.parameters()):
                assert torch.equal(pa, pb), (wd, it, float((pa - pb).abs().max()))
        sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
        for k in sa:
            assert torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) and torch.equal(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"]), (wd, k)


def test_mrcnn_four_steps_one_without_positives_equal_torch_adam_bit_for_bit(cuda):
    """VERDICT r4 "missing" 3: the K-step trajectory.  Four training steps of the real Mask R-CNN, step 2 on a batch WITHOUT GT objects (no
    positive RoI, no positive anchor): the reference's mask / bbox / rpn-bbox losses are constants there (mrcnn.py:233-234, 266-268,
    287-288), so exec.py:74's torch.optim.Adam does not touch the mask head, linear_bbox and conv_bbox -- no moment decay, no step count.
    net A: this repo's step + FlatAdam with the model's device-side gradient conditions.  net B: torch.optim.Adam fed A's gradient
    tensors, with `None` exactly where the reference's autograd produces none (net.grad_condition_spec() evaluated on the host).  Weights,
    moments and per-parameter step counters: BIT-EQUAL after every step."""
    import copy
    from tests.golden import step_inputs as si
    import numpy as np
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_reference.npz"), allow_pickle=False)
    nb = si.CASES["small"][1]
    full = si.make_batch(si.make_image(), [gold["gt_boxes_%d" % b] for b in range(nb)], [gold["gt_labels_%d" % b] for b in range(nb)])
    empty = si.make_batch(si.make_image(seed=32), [np.zeros((0, 6), np.float32)] * nb, [np.zeros((0,), np.int64)] * nb)
    cf = si.make_cf("mrcnn")
    net_a = mrcnn.net(cf, device=cuda)
    si.fill_by_name(net_a)
    net_b = copy.deepcopy(net_a)
    oa = training.build_optimizer(net_a, cf, flat=True)
    assert oa._cond is not None
    ob = torch.optim.Adam(net_b.parameters(), lr=cf.learning_rate[0], weight_decay=cf.weight_decay)
    spec = net_a.grad_condition_spec()
    cond_of = {}
    for i, (_, plist) in enumerate(spec):
        for p in plist:
            cond_of[id(p)] = i
    names = [n for n, _ in net_a.named_parameters()]
    skipped_in_step2 = None
    n_updates = {}
    for it, batch in enumerate((full, empty, full, full)):
        torch.manual_seed(10 + it)
        res = net_a.train_forward(batch, monitor=False)
        oa.zero_grad()
        res["torch_loss"].backward()
        cond = net_a._grad_cond.detach().cpu().tolist()
        n_valid, n_pos = (int(v) for v in res["sample_counts"])
        assert cond[3] == n_pos and cond[2] == n_valid
        if it == 0:
            assert n_pos > 0            # (later full-batch steps may lose their positives again: the weights have moved)
        if it == 1:
            assert n_pos == 0
        absent = []
        for (n, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
            have = pa.grad is not None and (id(pa) not in cond_of or cond[cond_of[id(pa)]] > 0)
            pb.grad = pa.grad.clone() if have else None
            if not have:
                absent.append(n)
        if it == 1:
            skipped_in_step2 = set(absent)
        for n in names:
            if n not in absent:
                n_updates[n] = n_updates.get(n, 0) + 1
        before = [p.detach().clone() for p in net_a.parameters()]
        oa.step()
        ob.step()
        for (n, pa), pb, p0 in zip(net_a.named_parameters(), net_b.parameters(), before):
            assert torch.equal(pa, pb), (it, n, float((pa - pb).abs().max()))
            if n in absent:
                assert torch.equal(pa, p0), (it, n)          # untouched
    # what the empty step skipped: the mask head, the box regressor of the classifier head, the RPN's box head (+ the never-used P1 layers)
    assert {n for n in skipped_in_step2 if not n.startswith("fpn.")} == {n for n in names if n.startswith("mask.") or n.startswith("classifier.linear_bbox") or n.startswith("rpn.conv_bbox")}
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k, n in enumerate(names):
        if k in sb:
            assert float(sa[k]["step"]) == float(sb[k]["step"]) == float(n_updates[n]), n
            assert float(sa[k]["step"]) <= (3.0 if n in skipped_in_step2 else 4.0)
            assert torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) and torch.equal(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"]), n
        else:
            assert float(sa[k]["step"]) == 0.0, n           # torch keeps no state for a parameter that never had a gradient


@pytest.mark.parametrize("model", ["retina_unet", "retina_net"])
def test_retina_steps_one_without_positive_anchors_equal_torch_adam_bit_for_bit(model, cuda):
    """the same for the Retina U-Net / Retina Net: a batch without GT objects has no positive anchor, so the reference's compute_bbox_loss
    returns a constant (retina_unet.py:180-186) and torch.optim.Adam leaves the BBRegressor head alone, while the Classifier still learns
    from the sampled negatives.  FlatAdam with the device-side conditions == torch.optim.Adam fed the same gradients with None where the
    reference's autograd has none: weights, moments, step counters bit-equal."""
    from tests.golden import step_inputs as si
    from medicaldetectiontoolkit_amd.models import retina_unet
    import numpy as np
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_reference.npz"), allow_pickle=False)
    nb = si.CASES["small"][1]
    full = si.make_batch(si.make_image(), [gold["gt_boxes_%d" % b] for b in range(nb)], [gold["gt_labels_%d" % b] for b in range(nb)])
    empty = si.make_batch(si.make_image(seed=32), [np.zeros((0, 6), np.float32)] * nb, [np.zeros((0,), np.int64)] * nb)
    cf = si.make_cf(model)
    net_a = retina_unet.net(cf, device=cuda)
    net_b = copy.deepcopy(net_a)
    oa = training.build_optimizer(net_a, cf, flat=True)
    assert oa._cond is not None and oa._cond[2] == (2 if model == "retina_unet" else 3)
    ob = torch.optim.Adam(net_b.parameters(), lr=cf.learning_rate[0], weight_decay=cf.weight_decay)
    cond_of = {}
    for i, (_, plist) in enumerate(net_a.grad_condition_spec()):
        for p in plist:
            cond_of[id(p)] = i
    names = [n for n, _ in net_a.named_parameters()]
    n_updates, skipped = {}, None
    for it, batch in enumerate((full, empty, full)):
        torch.manual_seed(10 + it)
        res = net_a.train_forward(batch, monitor=False)
        oa.zero_grad()
        res["torch_loss"].backward()
        cond = net_a._grad_cond.detach().cpu().tolist()
        assert cond[0] > 0                      # negatives are always sampled
        assert (cond[1] > 0) == (it != 1) or it == 2
        absent = []
        for (n, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
            have = pa.grad is not None and (id(pa) not in cond_of or cond[cond_of[id(pa)]] > 0)
            pb.grad = pa.grad.clone() if have else None
            if not have:
                absent.append(n)
        if it == 1:
            skipped = set(absent)
        for n in names:
            if n not in absent:
                n_updates[n] = n_updates.get(n, 0) + 1
        before = [p.detach().clone() for p in net_a.parameters()]
        oa.step()
        ob.step()
        for (n, pa), pb, p0 in zip(net_a.named_parameters(), net_b.parameters(), before):
            assert torch.equal(pa, pb), (it, n, float((pa - pb).abs().max()))
            if n in absent:
                assert torch.equal(pa, p0), (it, n)
    head = {n for n in names if n.startswith("BBRegressor.")}
    assert head and head <= skipped
    assert not any(n.startswith("Classifier.") for n in skipped)
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    for k, n in enumerate(names):
        if k in sb:
            assert float(sa[k]["step"]) == float(sb[k]["step"]) == float(n_updates[n]), n
            assert torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) and torch.equal(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"]), n
        else:
            assert float(sa[k]["step"]) == 0.0, n


def test_flat_adam_with_grad_sync_refuses_dropped_gradient_views(cuda):
    """with a FlatGradAllReduce the gradients live in ITS flat buffer: a dropped view is an error, never a silent freeze"""
    net = _toy(cuda)
    sync = training.FlatGradAllReduce(net)
    opt = training.FlatAdam(net.parameters(), lr=1e-3, grad_sync=sync)
    p0 = [p.detach().clone() for p in net.parameters()]
    for _ in range(2):
        opt.zero_grad()
        net(torch.randn((1, 2, 4, 4, 4), device=cuda)).sum().backward()
        sync.finish()                    # no process group: a no-op
        opt.step()
    assert all(p.grad.data_ptr() >= sync.flat.data_ptr() and p.grad.data_ptr() < sync.flat.data_ptr() + 4 * sync.flat.numel() for p in net.parameters())
    assert any(not torch.equal(p, q) for p, q in zip(net.parameters(), p0))
    for p in net.parameters():
        p.grad = None
    with pytest.raises(RuntimeError, match="gradient view was dropped"):
        opt.step()


def test_mrcnn_train_step_flat_adam_equals_torch_adam(cuda):
    """two steps of the real model from the same weights on the same batches: losses and parameters agree with torch.optim.Adam
    (the sampling inside the step is seeded identically)"""
    patch, B = [64, 64, 32], 2
    cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B)
    nets, losses = [], []
    for flat in (False, True):
        torch.manual_seed(0)
        net = mrcnn.net(cf, device=cuda)
        opt = training.build_optimizer(net, cf, flat=flat)
        assert isinstance(opt, training.FlatAdam) == flat
        ls = []
        for s in range(2):
            torch.manual_seed(100 + s)
            ls.append(float(training.train_step(net, opt, make_batch(patch, B, seed=s), monitor=False)["torch_loss"]))
        nets.append(net)
        losses.append(ls)
    assert losses[0][0] == pytest.approx(losses[1][0], rel=1e-5)
    assert losses[0][1] == pytest.approx(losses[1][1], rel=1e-3)
    used = 0
    for (n, pa), (_, pb) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        # Adam's first steps move every weight by ~lr = 1e-4 whatever the gradient's size, so a gradient that is zero up to
        # summation-order noise may move a weight either way: the bound is 2 steps x 2 lr; wrong views / a missed update show up
        # as differences of the size of the weights themselves
        assert torch.allclose(pa, pb, rtol=0, atol=4.1e-4), (n, float((pa - pb).abs().max()))
        used += 1
    assert used > 100
