"""RoIAlign-3D backward roofline variants only (bench.roialign_bwd_roofline without the training loop): warm / step-cache-state / cold
for the survey and train-realistic boxes -- for A/B runs of the zero-role geometry (MDT_BWD_TUNE=1 MDT_BWD3_ZERO_CHUNK_ROWS=k ...)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from medicaldetectiontoolkit_amd.configs import Configs
cf = Configs(dim=3, model="mrcnn", patch_size=[128, 128, 128], batch_size=8)
r = bench.roialign_bwd_roofline(cf, 8, torch.device("cuda:0"), None)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("MDT_BWD")}}
for k, v in r["variants"].items():
    out[k] = [v["avg_us"], v["frac"]]
print(json.dumps(out))
