"""Host-side cost of a training step: how long the CPU takes to ISSUE one step (the GPU-bound step stays GPU-bound only while this is
shorter than the GPU time) for resident and for host numpy batches, and the host cost of each upload of the numpy path.
One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medicaldetectiontoolkit_amd import miopen_env
miopen_env.setup()
import numpy as np
import torch
from medicaldetectiontoolkit_amd import training
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import mrcnn
from medicaldetectiontoolkit_amd.utils import model_utils as mutils
from medicaldetectiontoolkit_amd.utils.synthetic_data import make_batch, to_device

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
patch = [128, 128, 128]
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=8, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf)
host = [make_batch(patch, 8, seed=i) for i in range(2)]
res = [to_device(b, dev) for b in host]
out = {}
for name, pool in (("resident", res), ("host_numpy", host)):
    for i in range(3):
        training.train_step(net, opt, pool[i % 2], monitor=False)
    torch.cuda.synchronize()
    n = 10
    t0 = time.time()
    issue = []
    for i in range(n):
        t1 = time.time()
        training.train_step(net, opt, pool[i % 2], monitor=False)
        issue.append(time.time() - t1)
    t_issue = time.time() - t0
    torch.cuda.synchronize()
    t_all = time.time() - t0
    out[name] = {"ms_per_step": round(t_all / n * 1e3, 2), "host_issue_ms_per_step": round(t_issue / n * 1e3, 2),
                 "host_issue_ms_each": [round(v * 1e3, 1) for v in issue]}


def cpu_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    dt = (time.time() - t0) / reps
    torch.cuda.synchronize()
    return round(dt * 1e3, 2)


b = host[0]
masks = [m for m in b["roi_masks"] if len(m) > 0]
out["uploads_host_ms"] = {
    "image 67 MB: mutils.upload": cpu_ms(lambda: mutils.upload(b["data"], dev)),
    "masks %.0f MB: StagedUpload(...).get()" % (sum(m.nbytes for m in masks) / 1e6): cpu_ms(lambda: mutils.StagedUpload(masks, dev).get()),
    "masks: torch.cat + upload (round-3 mid state)": cpu_ms(lambda: mutils.upload(torch.cat([torch.as_tensor(np.ascontiguousarray(m)) for m in masks], 0), dev)),
    "GtOnDevice": cpu_ms(lambda: mrcnn.GtOnDevice(b["bb_target"], b["roi_labels"], 3, dev)),
    "pinned alloc 67 MB (cached)": cpu_ms(lambda: torch.empty((8, 1, 128, 128, 128), dtype=torch.float32, pin_memory=True)),
}
print(json.dumps(out))
