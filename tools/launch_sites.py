"""Which SOURCE LINES of this package the step's torch kernel launches come from (round 6: cutting the small-launch tail).  The eager exec-form
training step runs under torch.profiler with Python stacks; every top-level aten op is attributed to the innermost frame inside
medicaldetectiontoolkit_amd/ and its device kernels are counted.  Launches through the C ABI (ctypes, no aten op) are counted by name with
_lib.count_calls.  Usage: launch_sites.py [steps=3] [patch=128,128,128] [batch=8]"""
import collections
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from medicaldetectiontoolkit_amd import _lib, training  # noqa: E402
from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402
from medicaldetectiontoolkit_amd.models import mrcnn  # noqa: E402
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
patch = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [128, 128, 128]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=B, channels_last=True)
cf.run_detection_mask_head_in_training = True
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf, flat=True)
pool = [to_device(make_batch(patch, B, seed=i), dev) for i in range(2)]
for i in range(5):
    training.train_step(net, opt, pool[i % 2], monitor="deferred")
torch.cuda.synchronize()
_lib.count_calls(True)
for i in range(steps):
    training.train_step(net, opt, pool[i % 2], monitor="deferred")
torch.cuda.synchronize()
calls = dict(_lib.CALLS)
_lib.count_calls(False)
print("# C-ABI launches per step (ctypes, csrc/*.hip):  total %.1f" % (sum(calls.values()) / steps))
for k, v in sorted(calls.items(), key=lambda kv: -kv[1]):
    print("%7.1f  %s" % (v / steps, k))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(steps):
        training.train_step(net, opt, pool[i % 2], monitor="deferred")
    torch.cuda.synchronize()
sites = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
n_k = 0
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or ev.cpu_parent is not None:
        continue
    ks = [k for k in ev.kernels]
    if not ks:
        continue
    where = "(no package frame: autograd engine / backward of built-in ops)"
    for fr in ev.stack:
        if "medicaldetectiontoolkit_amd" in fr or "bench.py" in fr:
            where = fr.split("medicaldetectiontoolkit_amd/")[-1]
            break
    s = sites[where]
    s[0] += len(ks)
    s[1] += sum(k.duration for k in ks)
    s[2][ev.name] += len(ks)
    n_k += len(ks)
# ---- the forward's aten ops by source line (TorchDispatchMode sees the calling thread: forward + glue; the backward of built-in ops runs on
# the autograd engine's device thread and mirrors these)
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
kern_per_op = collections.defaultdict(lambda: [0, 0])
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.cpu_parent is None:
        kern_per_op[ev.name][0] += 1
        kern_per_op[ev.name][1] += len(ev.kernels)


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.ops = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        where = "?"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "medicaldetectiontoolkit_amd" in fr.filename and "_python_dispatch" not in fr.filename:
                where = "%s:%d %s" % (fr.filename.split("medicaldetectiontoolkit_amd/")[-1], fr.lineno, fr.name)
                break
        self.ops[where][str(func).replace("aten.", "aten::").split(".")[0]] += 1
        return func(*args, **(kwargs or {}))


with Sites() as sm:
    training.train_step(net, opt, pool[0], monitor="deferred")
torch.cuda.synchronize()
rows = []
for where, ops in sm.ops.items():
    est = 0.0
    for name, n in ops.items():
        c = kern_per_op.get(name)
        est += n * (c[1] / c[0] if c and c[0] else 1.0)
    rows.append((est, sum(ops.values()), where, ops))
print("# calling-thread aten ops of ONE step by source line: %d ops, ~%.0f kernel launches (ops x mean kernels per op of that name)" % (
    sum(r[1] for r in rows), sum(r[0] for r in rows)))
print("# ~launches  ops  site  {op: count}")
for est, n, where, ops in sorted(rows, key=lambda r: -r[0]):
    print("%7.1f %5d  %s  %s" % (est, n, where, dict(ops.most_common(8))))
print("# torch-op kernel launches per step by source line: total %.1f" % (n_k / steps))
print("# launches/step  us/step  site  {op: launches/step}")
for where, (n, us, ops) in sorted(sites.items(), key=lambda kv: -kv[1][0]):
    print("%7.1f %8.1f  %s  %s" % (n / steps, us / steps, where, {k: round(v / steps, 1) for k, v in ops.most_common(6)}))
