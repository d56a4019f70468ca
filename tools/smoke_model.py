import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch
from medicaldetectiontoolkit_amd import training
ps = [int(v) for v in os.environ.get("MDT_PATCH", "64,64,32").split(",")]
B = int(os.environ.get("MDT_B", 2))
cf = Configs(dim=len(ps), model="mrcnn", patch_size=ps, batch_size=B)
torch.manual_seed(0)
net = mrcnn.net(cf)
print("params", sum(p.numel() for p in net.parameters()), "anchors", tuple(net.anchors.shape))
opt = training.build_optimizer(net, cf)
batches = [make_batch(ps, B, seed=s, with_empty=(s == 1)) for s in range(3)]
for it in range(int(os.environ.get("MDT_STEPS", 6))):
    torch.cuda.synchronize(); t = time.time()
    r = training.train_step(net, opt, batches[it % 3], monitor=(it < 2))
    torch.cuda.synchronize(); dt = time.time() - t
    print("step", it, "loss %.4f" % r["torch_loss"].item(), "%.1f ms" % (dt * 1e3), r.get("logger_string", "")[:120], flush=True)
res = net.test_forward(batches[0], return_masks=True)
print("test_forward boxes per element", [len(b) for b in res["boxes"]], res["seg_preds"].shape)
