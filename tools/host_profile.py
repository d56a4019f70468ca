"""Where the HOST time of a training step goes (the step is GPU-bound only while the host issues it faster than the GPU runs it:
38 ms vs 43 ms at the end of round 3).  cProfile over 5 resident-batch steps, top functions by own time and by cumulative time."""
import cProfile, io, os, pstats, sys, time
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
patch = [128, 128, 128]
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=8, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf)
pool = [to_device(make_batch(patch, 8, seed=i), dev) for i in range(2)]
for i in range(4):
    training.train_step(net, opt, pool[i % 2], monitor=False)
torch.cuda.synchronize()
# phase split without a profiler: forward+loss / backward / optimizer, host time only (no syncs inside)
n = 6
t_f = t_b = t_o = 0.0
for i in range(n):
    t0 = time.time()
    opt.zero_grad(set_to_none=False) if False else None
    r = net.train_forward(pool[i % 2], monitor=False)
    t1 = time.time()
    opt.zero_grad()
    r["torch_loss"].backward()
    t2 = time.time()
    opt.step()
    t3 = time.time()
    t_f += t1 - t0; t_b += t2 - t1; t_o += t3 - t2
torch.cuda.synchronize()
print("host ms per step: train_forward %.1f, backward %.1f, optimizer %.1f" % (t_f / n * 1e3, t_b / n * 1e3, t_o / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    training.train_step(net, opt, pool[i % 2], monitor=False)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(40)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[:60]))
