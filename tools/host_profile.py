"""Host-side profile (cProfile) of the eager training step at the benchmarked configuration: where the ~32 ms of host work per step go.
The autograd engine runs backward nodes on its own thread; Python Function.backward bodies show up here through the GIL only as wall time of
`run_backward`, so the backward's Python share is listed separately by wrapping the custom Functions.   Usage: host_profile.py [steps=10] [patch=128,128,128]
(a small patch, e.g. 64,64,32, takes the GPU out of the picture: at the benchmarked size the host blocks on the full launch queue inside random ops)"""
import cProfile
import io
import os
import pstats
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()
import torch  # noqa: E402
from medicaldetectiontoolkit_amd import training  # noqa: E402
from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402
from medicaldetectiontoolkit_amd.models import mrcnn  # noqa: E402
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
MON = "deferred" if (len(sys.argv) > 3 and sys.argv[3] == "exec") else False          # third argument "exec": the exec-form step (read-out + detection mask head)
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
patch = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [128, 128, 128]
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=8, channels_last=True)
cf.run_detection_mask_head_in_training = bool(MON)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf, flat=True)
pool = [to_device(make_batch(patch, 8, seed=i), dev) for i in range(2)]
for i in range(6):
    training.train_step(net, opt, pool[i % 2], monitor=MON)
torch.cuda.synchronize()
t0 = time.time()
for i in range(steps):
    training.train_step(net, opt, pool[i % 2], monitor=MON)
th = time.time() - t0
torch.cuda.synchronize()
print("unprofiled: host issue %.2f ms/step, wall %.2f ms/step" % (th / steps * 1e3, (time.time() - t0) / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    training.train_step(net, opt, pool[i % 2], monitor=MON)
pr.disable()
torch.cuda.synchronize()
for key, n in (("tottime", 45), ("cumulative", 60)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(n)
    lines = s.getvalue().splitlines()
    print("\n".join(l[:170] for l in lines[4:]))
