"""Training steps with the RoI heads FULL, for rocprofv3 (round 5; VERDICT r4 "weak" 3: the roofline headline of the bench line is the
mask head's pyramid RoIAlign backward AS IT RUNS inside such steps -- this gives it a rocprofv3 kernel trace and PMC passes).

    rocprofv3 --kernel-trace --stats ... -- python tools/instep_heads_full.py [steps]
    rocprofv3 --pmc WRITE_SIZE   ...    -- python tools/instep_heads_full.py [steps]        (FETCH_SIZE: its own pass)

The batch: GT boxes = two large disjoint proposals of the net's own RPN per element (utils/synthetic_data.batch_with_gt_from_proposals,
the construction bench.py's `heads_full_step` / in-step roofline use).  Every step is preceded by a plain `zero_()` of a
8 x 36 x 32 x 32 x 128 fp32 tensor: the 150 994 944-byte calibration write MI355X_MICROARCH.md asks for (WRITE_SIZE is uncalibrated).
Autograd runs the mask head's RoIAlign backward (pool 14x14x5) BEFORE the classifier's (7x7x3) -- the mask head is the later forward
-- so in every step the FIRST crop_bwd_gather_kernel launch is the one the bench line's roofline names.  tools/instep_extract.py
reads the traces."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()
import torch  # noqa: E402

from medicaldetectiontoolkit_amd import training  # noqa: E402
from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402
from medicaldetectiontoolkit_amd.models import mrcnn  # noqa: E402
from medicaldetectiontoolkit_amd.utils.synthetic_data import batch_with_gt_from_proposals, make_batch, to_device  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
cf = Configs(dim=3, model="mrcnn", patch_size=[128, 128, 128], batch_size=8, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf, flat=True)
b = to_device(make_batch([128, 128, 128], 8, seed=1000), dev)
for _ in range(3):
    training.train_step(net, opt, b, monitor=False)
b48 = batch_with_gt_from_proposals(net, cf, b, dev)
cal = torch.empty((8, 36, 32, 32, 128), device=dev)
counts = []
for i in range(2 + steps):
    cal.zero_()
    counts.append(training.train_step(net, opt, b48, monitor=False)["sample_counts"])
torch.cuda.synchronize()
print(json.dumps({"steps": steps, "warmup_steps_on_the_batch": 2, "valid_rois": [int(a) for a, _ in counts[2:]], "positive_rois": [int(p) for _, p in counts[2:]]}))
