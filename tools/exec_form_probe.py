"""Where the exec-form step loses time against the no-read-out step: the four combinations of {read-out: off / deferred / synchronous} x {mask head over the
detections: off / on}, same net, same batches, 20 timed steps each.  Usage: exec_form_probe.py [steps=20]"""
import os
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
patch = [128, 128, 128]
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=8, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf, flat=True)
pool = [to_device(make_batch(patch, 8, seed=1000 + i), dev) for i in range(3)]
for mon, mh in ((False, False), (False, True), ("deferred", False), ("deferred", True), (True, True), (False, False)):
    cf.run_detection_mask_head_in_training = mh
    for i in range(4):
        training.train_step(net, opt, pool[i % 3], monitor=mon)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(steps):
        r = training.train_step(net, opt, pool[i % 3], monitor=mon)
        if "logger_string" in r:
            _ = len(r["logger_string"]) + len(r["boxes"])
    th = time.time() - t0
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("monitor=%-9s detection_mask_head=%-5s  %.2f ms/step  (host issue %.2f)  %.1f patches/s" % (mon, mh, dt / steps * 1e3, th / steps * 1e3, 8 * steps / dt), flush=True)
