import os, sys, traceback, collections
sys.path.insert(0, os.getcwd())
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import torch
from medicaldetectiontoolkit_amd import training, _lib
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device
dev = torch.device("cuda:0")
cf = Configs(dim=3, model="mrcnn", patch_size=[128, 128, 128], batch_size=8, channels_last=True)
cf.run_detection_mask_head_in_training = True
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf, flat=True)
b = to_device(make_batch([128, 128, 128], 8, seed=1000), dev)
for _ in range(2):
    training.train_step(net, opt, b, monitor="deferred")
L = _lib.lib()
log = collections.Counter()
def wrap(name, nidx, cidx, ridx):
    fn = getattr(L, name)
    def f(*a):
        if a[nidx] >= (1 << 22):
            st = [s for s in traceback.extract_stack() if "medicaldetectiontoolkit_amd" in s.filename][-4:]
            log[(name, a[nidx], a[cidx], a[ridx], " < ".join("%s:%d %s" % (os.path.basename(s.filename), s.lineno, s.name) for s in reversed(st)))] += 1
        return fn(*a)
    setattr(L, name, f)
wrap("mdt_bias_act_forward", 4, 5, 7)
wrap("mdt_bias_act_backward", 4, 5, 7)
training.train_step(net, opt, b, monitor="deferred")
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[0][1]):
    print(v, k)
