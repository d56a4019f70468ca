"""hipGraph capture of the static-shape segment of the training step (image -> FPN -> RPN heads -> row-major pyramid copies):
does torch.cuda.make_graphed_callables capture it with this repo's ctypes-launched kernels and MIOpen inside, are outputs and
parameter gradients identical to eager, and what does it do to the HOST time of the segment (the step needs 38 ms
of host time per 43 ms of GPU time)?  NOT part of the product; written at the end of round 3 to be run first thing in round 4.
One JSON line.  usage: python tools/graph_probe.py [steps]
First run (end of round 3, profiles/r03_graph_probe_first_try.json): eager segment 39.7 ms GPU / 29.3 ms host of the 43 / 38 ms step;
capture stopped at the FPN's two unused parameters (allow_unused_input was off) -- fixed here, not yet re-run."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
import torch.nn as nn
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
patch, B = [128, 128, 128], 8
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)


class Segment(nn.Module):
    """what mrcnn.net.forward does before the first data-dependent step (proposal_layer)"""

    def __init__(self, net):
        super().__init__()
        self.fpn, self.rpn, self.levels, self.mf = net.fpn, net.rpn, list(net.cf.pyramid_levels), net.memory_format

    def forward(self, img):
        if self.mf is not None:
            img = img.contiguous(memory_format=self.mf)
        outs = self.fpn(img)
        maps = [outs[i] for i in self.levels]
        row_major = [m.contiguous() for m in maps]
        logits, probs, deltas = [torch.cat(list(o), dim=1) for o in zip(*[self.rpn(p) for p in maps])]
        return tuple(row_major) + (logits, probs, deltas)


seg = Segment(net)
img = torch.randn((B, 1) + tuple(patch), device=dev)


def run(fn, x):
    outs = fn(x)
    loss = sum(o.float().square().mean() for o in outs)
    for p in seg.parameters():
        p.grad = None
    loss.backward()
    return [o.detach().clone() for o in outs], {n: p.grad.detach().clone() for n, p in seg.named_parameters() if p.grad is not None}


def timed(fn, x, n):
    for _ in range(2):
        run(fn, x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        run(fn, x)
    t_host = time.time() - t0
    torch.cuda.synchronize()
    return round(t_host / n * 1e3, 2), round((time.time() - t0) / n * 1e3, 2)


rec = {"segment": "image -> FPN -> RPN (+ row-major pyramid copies), fwd + bwd, 8 x 128^3"}
rec["eager_host_ms"], rec["eager_ms"] = timed(seg, img, steps)
ref_out, ref_grad = run(seg, img)
try:
    graphed = torch.cuda.make_graphed_callables(seg, (img,), num_warmup_iters=3, allow_unused_input=True)   # P1_conv1/2 are never used (backbone.py:112,118)
    rec["graphed_host_ms"], rec["graphed_ms"] = timed(graphed, img, steps)
    img2 = torch.randn_like(img)               # new input values through the static input buffer
    o_e, g_e = run(seg, img2)
    o_g, g_g = run(graphed, img2)
    rec["outputs_max_abs_diff"] = max(float((a - b).abs().max()) for a, b in zip(o_e, o_g))
    rec["grads_max_rel_diff"] = max(float((g_e[n] - g_g[n]).abs().max() / g_e[n].abs().max().clamp(min=1e-30)) for n in g_e)
    rec["params_with_grad"] = [len(g_e), len(g_g)]
    rec["captured"] = True
except Exception as e:  # the answer next round starts from
    rec["captured"] = False
    rec["error"] = repr(e)[:600]
print(json.dumps(rec))
