"""Which part of the Mask R-CNN training step breaks hipGraph capture (tools/graph_step_probe.py: segfault in capture_end)?
Captures ONE prefix / piece of the step per process (a crash kills the process): usage  python tools/graph_bisect_probe.py <mode>
modes: proposals | forward | targets | heads | match | losses | full_nobwd | full"""
import json, os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn
from medicaldetectiontoolkit_amd.utils import model_utils as mutils
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device

mode = sys.argv[1]
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
patch, B = [128, 128, 128], 8
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
batch = to_device(make_batch(patch, B, seed=1), dev)
gt_dev = mrcnn.GtOnDevice(batch["bb_target"], batch["roi_labels"], cf.dim, dev)
mrcnn.GtOnDevice = lambda *a, **k: gt_dev
img = batch["data"].float()
params = [p for p in net.parameters() if p.requires_grad]
state = {}


def piece():
    if mode in ("full", "full_nobwd"):
        res = net.train_forward(batch, monitor=False)
        if mode == "full":
            for p in params:
                p.grad = None
            res["torch_loss"].backward()
        return res["torch_loss"]
    with torch.no_grad():
        if mode == "proposals":
            x = img.contiguous(memory_format=net.memory_format)
            outs = net.fpn(x)
            maps = [outs[i] for i in cf.pyramid_levels]
            lo = [net.rpn(p) for p in maps]
            logits, probs, deltas = [torch.cat(list(o), dim=1) for o in zip(*lo)]
            return mrcnn.proposal_layer(probs, deltas, cf.post_nms_rois_training, net.anchors, cf)[0]
        if mode == "forward":
            return net.forward(img, with_masks=False)[3]
        if mode == "forward_masks":
            return net.forward(img, with_masks=True)[5]
        if "fwd" not in state:
            state["fwd"] = net.forward(img, with_masks=False)
        if mode == "targets":
            return mrcnn.detection_target_layer(net.rpn_rois_batch_info, net.batch_mrcnn_class_scores, batch["roi_labels"], batch["bb_target"],
                                                batch["roi_masks_device"], cf, B, gt_dev=gt_dev)[0]
        if mode == "heads":
            return net.loss_samples_forward(batch["roi_labels"], batch["bb_target"], batch["roi_masks_device"], B, gt_dev=gt_dev)[0]
        if mode in ("match", "losses"):
            ms, ams = [], []
            for b in range(B):
                gt_t = gt_dev.px[b, :gt_dev.n_all[b]] if gt_dev.n_all[b] > 0 else None
                m, am, _, _ = mutils.anchor_match_labels(net.anchors_f64, gt_t, None, 0.01, float(cf.anchor_matching_iou))
                ms.append(m); ams.append(am)
            rm, ra = torch.stack(ms), torch.stack(ams)
            if mode == "match":
                return rm
            return mrcnn.compute_rpn_losses(rm, ra, state["fwd"][0], state["fwd"][1], net.anchors_f64, batch["bb_target"], cf, gt_dev=gt_dev)[0]
    raise SystemExit("unknown mode")


rec = {"mode": mode}
try:
    for _ in range(3):
        piece()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        piece()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = piece()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        g.replay()
    th = time.time() - t0
    torch.cuda.synchronize()
    rec.update(captured=True, graph_host_ms=round(th / 5 * 1e3, 3), graph_ms=round((time.time() - t0) / 5 * 1e3, 3), out_sum=float(out.double().sum()))
except Exception as e:
    rec.update(captured=False, error=repr(e)[:500], trace=traceback.format_exc()[-800:])
print(json.dumps(rec), flush=True)
