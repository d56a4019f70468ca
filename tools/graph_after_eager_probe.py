"""Does training.GraphedTrainStep capture AFTER eager training steps of the same net in the same process (the order that segfaulted in
hipStreamEndCapture in the first r04_final run), now that capture() drops the net's references to the last eager step?  One line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
from medicaldetectiontoolkit_amd import training
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
patch, B = ([64, 64, 32], 2) if "small" in sys.argv else ([128, 128, 128], 8)
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf, flat=True)
b = to_device(make_batch(patch, B, seed=1), dev)
for _ in range(3):
    r = training.train_step(net, opt, b, monitor=False)
del r
torch.cuda.synchronize()
print("eager steps done", flush=True)
g = training.GraphedTrainStep(net, opt, gmax=8)
for _ in range(3):
    r = g(b)
torch.cuda.synchronize()
print("CAPTURE_AFTER_EAGER_OK", float(r["torch_loss"]), flush=True)
