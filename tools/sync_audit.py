"""One warm training step under torch.cuda.set_sync_debug_mode("warn"): prints every host<->device synchronisation torch
sees inside train_step (pageable uploads, .item(), nonzero ...).  The step is meant to have none."""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import training  # noqa: E402
from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402
from medicaldetectiontoolkit_amd.models import mrcnn, retina_unet  # noqa: E402
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "mrcnn"
patch = [64, 64, 32]
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cf = Configs(dim=3, model=model, patch_size=patch, batch_size=2, channels_last=True)
net = (mrcnn if model == "mrcnn" else retina_unet).net(cf, device=dev)
opt = training.build_optimizer(net, cf)
pool = [to_device(make_batch(patch, 2, seed=i), dev) for i in range(2)]
for i in range(3):
    training.train_step(net, opt, pool[i % 2], monitor=False)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    training.train_step(net, opt, pool[1], monitor=False)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("synchronising calls in one %s train_step: %d" % (model, len(w)))
for x in w:
    print("  %s:%d  %s" % (x.filename.replace(os.getcwd() + "/", ""), x.lineno, str(x.message)[:120]))
