"""torch.profiler attribution of one bench-shaped training step: device time per aten op (which torch ops the
elementwise / copy kernels of the rocprof breakdown belong to).  usage: python tools/op_profile.py [steps]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import training  # noqa: E402
from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402
from medicaldetectiontoolkit_amd.models import mrcnn  # noqa: E402
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
patch = [128, 128, 128]
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=8, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf)
pool = [to_device(make_batch(patch, 8, seed=i), dev) for i in range(2)]
for i in range(3):
    training.train_step(net, opt, pool[i % 2], monitor=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(steps):
        training.train_step(net, opt, pool[i % 2], monitor=False)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0)
    if dt > 0 and (e.key.startswith("aten::") or e.key.startswith("_") and not e.key.startswith("_ZN")):
        rows.append((dt / steps, e.count / steps, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print("self device time per step (us), calls per step, op, input shapes")
for r in rows[:70]:
    print("%9.1f  %6.1f  %-38s %s" % r)

# the glue: ops whose launches average < 40 us, grouped by op name over all shapes ("torch elementwise + fills")
small = {}
tot_small = 0.0
for dt, cnt, key, shapes in rows:
    if cnt > 0 and dt / cnt < 40.0 and key.startswith("aten::"):
        a = small.setdefault(key, [0.0, 0.0])
        a[0] += dt
        a[1] += cnt
        tot_small += dt
print("\nsmall aten ops (< 40 us per call), by name: us per step, calls per step   [total %.0f us per step]" % tot_small)
for key, (dt, cnt) in sorted(small.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%9.1f  %6.1f  %s" % (dt, cnt, key))
