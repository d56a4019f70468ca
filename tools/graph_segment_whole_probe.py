"""Whole-capture (torch.cuda.graph, forward + loss + backward in ONE graph) of the static segment image -> FPN -> RPN of the
3D Mask R-CNN step: host time and total time of a replay against the eager segment.  Separates the cost of a hipGraph replay of
MIOpen / own-kernel nodes from the overheads of torch.cuda.make_graphed_callables (tools/graph_probe.py).  One JSON line."""
import json, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
what = sys.argv[2] if len(sys.argv) > 2 else "fpn_rpn"
patch, B = [128, 128, 128], 8
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
img = torch.randn((B, 1) + tuple(patch), device=dev).contiguous(memory_format=torch.channels_last_3d)
params = [p for p in list(net.fpn.parameters()) + list(net.rpn.parameters())]


def fwd_bwd():
    outs = net.fpn(img)
    if what == "fpn_rpn":
        maps = [outs[i] for i in cf.pyramid_levels]
        heads = [torch.cat(list(o), dim=1) for o in zip(*[net.rpn(p) for p in maps])]
        outs = [m.contiguous() for m in maps] + heads
    loss = sum(o.float().square().mean() for o in outs)
    for p in params:
        p.grad = None
    loss.backward()
    return loss


rec = {"segment": what + " fwd + loss + bwd, one torch.cuda.graph, 8 x 128^3"}
try:
    for _ in range(3):
        fwd_bwd()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        fwd_bwd()
    th = time.time() - t0
    torch.cuda.synchronize()
    rec["eager_host_ms"], rec["eager_ms"] = round(th / steps * 1e3, 2), round((time.time() - t0) / steps * 1e3, 2)
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
        loss = fwd_bwd()
    torch.cuda.synchronize()
    rec["capture_s"] = round(time.time() - t0, 2)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        g.replay()
    th = time.time() - t0
    torch.cuda.synchronize()
    rec["graph_host_ms"], rec["graph_ms"] = round(th / steps * 1e3, 2), round((time.time() - t0) / steps * 1e3, 2)
    rec["loss"] = float(loss)
    rec["captured"] = True
except Exception as e:
    rec["captured"] = False
    rec["error"] = repr(e)[:600]
    rec["trace"] = traceback.format_exc()[-1200:]
print(json.dumps(rec))
