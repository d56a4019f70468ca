"""Times FPN+RPN forward+backward at the bench shape under different MIOpen / layout settings.
usage: conv_probe.py <mode>   mode in: default | benchmark | cl3d | cl3d_benchmark"""
import os, sys, time
mode = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from medicaldetectiontoolkit_amd.configs import Configs
from medicaldetectiontoolkit_amd.models import backbone as bb
from medicaldetectiontoolkit_amd.models.mrcnn import RPN
from medicaldetectiontoolkit_amd.utils.model_utils import NDConvGenerator
if "benchmark" in mode:
    torch.backends.cudnn.benchmark = True
B = int(os.environ.get("MDT_B", 8))
cf = Configs(dim=3, model="mrcnn", patch_size=[128, 128, 128], batch_size=B)
dev = torch.device("cuda:0")
conv = NDConvGenerator(3)
fpn, rpn = bb.FPN(cf, conv).to(dev), RPN(cf, conv).to(dev)
x = torch.randn(B, 1, 128, 128, 128, device=dev)
if "cl3d" in mode:
    fpn = fpn.to(memory_format=torch.channels_last_3d); rpn = rpn.to(memory_format=torch.channels_last_3d)
    x = x.contiguous(memory_format=torch.channels_last_3d)
for it in range(4):
    torch.cuda.synchronize(); t = time.time()
    outs = fpn(x)
    loss = sum(sum(o.float().mean() for o in rpn(p)) for p in outs)
    loss.backward()
    torch.cuda.synchronize(); print(mode, "iter", it, "%.1f ms" % ((time.time() - t) * 1e3), flush=True)
