"""Whole-step hipGraph probe (round 4, first GPU call): does torch.cuda.graph capture train_forward + backward of the 3D Mask R-CNN
step (MIOpen convolutions, this repo's ctypes-launched kernels, autograd worker threads, device RNG) on this stack, are the
replayed losses / gradients identical to the eager step on the same batch, and what do host and GPU time per step become?
NOT product code (the product path is training.GraphedTrainStep); one JSON line.  usage: python tools/graph_step_probe.py [steps]"""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()
import torch  # noqa: E402
from medicaldetectiontoolkit_amd import training  # noqa: E402
from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402
from medicaldetectiontoolkit_amd.models import mrcnn  # noqa: E402
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
patch, B = [128, 128, 128], 8
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf, flat=True)
batch = to_device(make_batch(patch, B, seed=1), dev)
rec = {"probe": "whole-step capture: train_forward + backward, 8 x 128^3 Mask R-CNN"}

for _ in range(4):
    training.train_step(net, opt, batch, monitor=False)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    training.train_step(net, opt, batch, monitor=False)
th = time.time() - t0
torch.cuda.synchronize()
rec["eager_host_ms"], rec["eager_ms"] = round(th / steps * 1e3, 2), round((time.time() - t0) / steps * 1e3, 2)

# GT table built once, outside the capture (its pinned staging + H2D copy are host-side work)
gt_dev = mrcnn.GtOnDevice(batch["bb_target"], batch["roi_labels"], cf.dim, dev)
orig_gt = mrcnn.GtOnDevice
mrcnn.GtOnDevice = lambda *a, **k: gt_dev
params = [p for p in net.parameters() if p.requires_grad]


def fwd_bwd():
    res = net.train_forward(batch, monitor=False)
    for p in params:
        p.grad = None
    res["torch_loss"].backward()
    return res


try:
    # eager reference on a fixed RNG state
    torch.cuda.synchronize()
    rng_state = torch.cuda.get_rng_state(dev)
    res_e = fwd_bwd()
    loss_e = {k: float(v) for k, v in res_e["loss_terms"].items()}
    grads_e = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    # warm-up on a side stream (the documented protocol), then capture
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    t0 = time.time()
    with torch.cuda.graph(g):
        res_g = fwd_bwd()
    torch.cuda.synchronize()
    rec["capture_s"] = round(time.time() - t0, 2)
    rec["captured"] = True
    static_grads = [p.grad for p in params]
    g.replay()
    torch.cuda.synchronize()
    loss_g = {k: float(v) for k, v in res_g["loss_terms"].items()}
    rec["loss_eager"], rec["loss_graph"] = loss_e, loss_g
    grads_g = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    rec["grads_max_rel_diff"] = max(float((grads_e[n] - grads_g[n]).abs().max() / grads_e[n].abs().max().clamp(min=1e-30)) for n in grads_e if n in grads_g)
    rec["params_with_grad"] = [len(grads_e), len(grads_g)]

    def graph_step():
        g.replay()
        for p, gr in zip(params, static_grads):
            p.grad = gr
        opt.step()

    for _ in range(3):
        graph_step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        graph_step()
    th = time.time() - t0
    torch.cuda.synchronize()
    rec["graph_host_ms"], rec["graph_ms"] = round(th / steps * 1e3, 2), round((time.time() - t0) / steps * 1e3, 2)
    # replay only (no optimizer): the graph's own GPU time
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        g.replay()
    th = time.time() - t0
    torch.cuda.synchronize()
    rec["replay_only_host_ms"], rec["replay_only_ms"] = round(th / steps * 1e3, 2), round((time.time() - t0) / steps * 1e3, 2)
    rec["mem_allocated_GB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
except Exception as e:
    rec["captured"] = rec.get("captured", False)
    rec["error"] = repr(e)[:800]
    rec["trace"] = traceback.format_exc()[-1500:]
finally:
    mrcnn.GtOnDevice = orig_gt
print(json.dumps(rec))
