"""How many steps a FRESH process on a FRESH box needs before the exec-form step runs at its steady rate: ms per step in groups of 5 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
from medicaldetectiontoolkit_amd import training
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
mode = sys.argv[1] if len(sys.argv) > 1 else "exec"
cf = Configs(dim=3, model="mrcnn", patch_size=[128, 128, 128], batch_size=8, channels_last=True)
cf.run_detection_mask_head_in_training = mode == "exec"
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf, flat=True)
pool = [to_device(make_batch([128, 128, 128], 8, seed=1000 + i), dev) for i in range(3)]
mon = "deferred" if mode == "exec" else False
out = []
for g in range(int(os.environ.get("GROUPS", "14"))):
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(5):
        training.train_step(net, opt, pool[i % 3], monitor=mon)
    torch.cuda.synchronize()
    out.append(round((time.time() - t0) / 5 * 1e3, 2))
print(mode, "ms/step per group of 5:", out, "reserved GB", round(torch.cuda.memory_reserved() / 2 ** 30, 1), flush=True)
