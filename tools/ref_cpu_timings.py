"""Times the REFERENCE's numpy functions (imported from /root/reference -- build container only) on this host's
CPU for the BASELINE.md CPU-baseline table.  Writes profiles/r01_reference_numpy_cpu_timings.json."""
import json, logging, os, sys, time, warnings
warnings.filterwarnings("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import numpy as np
import utils.model_utils as mutils
import predictor as P
sys.path.remove("/root/reference")
from medicaldetectiontoolkit_amd.configs import Configs
log = logging.getLogger("x"); log.addHandler(logging.NullHandler())
rng = np.random.default_rng(0); out = {"host": "build container, %d cores" % os.cpu_count()}
for model in ("mrcnn", "retina_unet"):
    cf = Configs(dim=3, model=model); cf.rpn_train_anchors_per_image = 6
    t = time.time(); a = mutils.generate_pyramid_anchors(log, cf); out["generate_pyramid_anchors_3d_%s_s" % model] = round(time.time() - t, 4)
    for G in (1, 3, 8):
        c = rng.uniform(30, 100, size=(G, 3)); s = rng.uniform(8, 30, size=(G, 3))
        gt = np.stack([c[:, 0] - s[:, 0], c[:, 1] - s[:, 1], c[:, 0] + s[:, 0], c[:, 1] + s[:, 1], c[:, 2] - s[:, 2] / 2, c[:, 2] + s[:, 2] / 2], 1)
        t = time.time(); mutils.gt_anchor_matching(cf, a, gt, rng.integers(1, 3, size=G)); out["gt_anchor_matching_A%d_G%d_s" % (a.shape[0], G)] = round(time.time() - t, 4)
for n in (2000, 45000):
    true = rng.uniform(40, 400, size=(20, 3)); which = rng.integers(0, 20, size=n)
    c = true[which] + rng.normal(0, 2.0, size=(n, 3)); s = rng.uniform(6, 20, size=(n, 3))
    dets = np.concatenate([np.stack([c[:, 0] - s[:, 0], c[:, 1] - s[:, 1], c[:, 0] + s[:, 0], c[:, 1] + s[:, 1], c[:, 2] - s[:, 2], c[:, 2] + s[:, 2]], 1),
                           rng.permutation(np.linspace(0.02, 0.99, n))[:, None], rng.uniform(0.2, 1, (n, 1)), rng.integers(1, 5, (n, 1)).astype(float)], 1)
    pid = np.array(["%d" % v for v in rng.integers(0, 1500, size=n)])
    t = time.time(); ks, kc = P.weighted_box_clustering(dets, pid, 1e-5, 20); out["weighted_box_clustering_n%d_s" % n] = round(time.time() - t, 4)
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_reference_numpy_cpu_timings.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
