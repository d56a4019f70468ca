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


def test_flat_adam_missing_gradient_is_a_zero_gradient(cuda):
    """a parameter without a gradient in a step (p.grad None) is updated with a ZERO gradient (documented convention; torch skips
    it): equal to torch.optim.Adam fed explicit zeros, and last step's gradient is not applied twice"""
    a, b = _toy(cuda, 5), _toy(cuda, 5)
    oa = torch.optim.Adam(a.parameters(), lr=1e-2)
    ob = training.FlatAdam(b.parameters(), lr=1e-2)
    x = torch.randn((2, 2, 4, 4, 4), device=cuda)
    for it in range(4):
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            net(x).square().mean().backward()
        if it >= 2:                      # the last layer gets no gradient in steps 2 and 3
            for p in a[4].parameters():
                p.grad = torch.zeros_like(p)
            for p in b[4].parameters():
                p.grad = None
        oa.step()
        ob.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6)


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
