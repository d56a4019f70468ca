"""How sparse is the gradient that reaches the FPN outputs (= the output gradient of P2_conv2 .. P5_conv2) in training steps?
It is non-zero only inside the sampled RoI boxes routed to a level (RoIAlign backward) and at the sampled RPN rows (rpn_at_anchors), so the
post-convolutions' input- and weight-gradient kernels mostly multiply zero tiles.  Prints, per level and step: fraction of
voxels with a non-zero gradient row, and of tiles of 1 x 4 x 32 voxels (y, x, z: the tile of csrc/conv3x3x3_small.hip) / 8 x 8 x 16 voxels
that contain any.   Usage: fpn_grad_sparsity.py [steps=3]   (bench configuration: 128^3, batch 8; random GT batches, then GT from proposals)"""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd import miopen_env  # noqa: E402
miopen_env.setup()
import torch  # noqa: E402
from medicaldetectiontoolkit_amd import training  # noqa: E402
from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402
from medicaldetectiontoolkit_amd.models import mrcnn  # noqa: E402
from medicaldetectiontoolkit_amd.utils.synthetic_data import batch_with_gt_from_proposals, make_batch, to_device  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
patch = [128, 128, 128]
cf = Configs(dim=3, model="mrcnn", patch_size=patch, batch_size=8, channels_last=True)
torch.manual_seed(0)
net = mrcnn.net(cf, device=dev)
opt = training.build_optimizer(net, cf, flat=True)
stats = []


def tile_frac(nz, ty, tx, tz):
    B, Y, X, Z = nz.shape
    ty, tx, tz = min(ty, Y), min(tx, X), min(tz, Z)
    t = nz[:, :Y // ty * ty, :X // tx * tx, :Z // tz * tz].reshape(B, Y // ty, ty, X // tx, tx, Z // tz, tz)
    return float(t.any(6).any(4).any(2).float().mean())


fpn_forward = net.fpn.forward


def hooked(x):
    outs = fpn_forward(x)
    for li, o in enumerate(outs):
        if o.requires_grad:
            def h(g, li=li):
                nz = (g != 0).any(dim=1)
                stats.append((li, float(nz.float().mean()), tile_frac(nz, 1, 4, 32), tile_frac(nz, 8, 8, 16)))
            o.register_hook(h)
    return outs


net.fpn.forward = hooked
for name, batches in (("random GT boxes (the timed loop's batches)", [to_device(make_batch(patch, 8, seed=i), dev) for i in range(steps)]),
                      ("GT from the net's own proposals (RoI heads full)", None)):
    if batches is None:
        batches = [batch_with_gt_from_proposals(net, cf, to_device(make_batch(patch, 8, seed=0), dev), dev)] * steps
    for b in batches:
        del stats[:]
        res = training.train_step(net, opt, b, monitor=False)
        torch.cuda.synchronize()
        print(name, "| " + " | ".join("P%d voxels %.4f tiles(1x4x32) %.3f tiles(8x8x16) %.3f" % (li + 2, a, t1, t2) for li, a, t1, t2 in sorted(stats)), flush=True)
