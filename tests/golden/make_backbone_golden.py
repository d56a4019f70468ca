"""Writes tests/golden/backbone_reference.npz: outputs of the REFERENCE's FPN (/root/reference/models/backbone.py:22-206,
built with its NDConvGenerator, utils/model_utils.py:732-781) run by torch on the CPU in fp32, with the name-seeded
weights of tests/golden/backbone_inputs.py.  12 000 evenly spaced elements of each pyramid level are kept (plus max-abs per level).
Run once in the build container:  timeout 600 python tests/golden/make_backbone_golden.py"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import backbone_inputs as bi  # noqa: E402

for name in ["cuda_functions", "cuda_functions.nms_2D", "cuda_functions.nms_2D.pth_nms", "cuda_functions.nms_3D",
             "cuda_functions.nms_3D.pth_nms", "cuda_functions.roi_align_2D", "cuda_functions.roi_align_2D.roi_align",
             "cuda_functions.roi_align_2D.roi_align.crop_and_resize", "cuda_functions.roi_align_3D",
             "cuda_functions.roi_align_3D.roi_align", "cuda_functions.roi_align_3D.roi_align.crop_and_resize"]:
    m = types.ModuleType(name)
    m.nms_gpu = None
    m.CropAndResizeFunction = None
    sys.modules[name] = m
sys.path.insert(0, REF)


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


mu = load("utils/model_utils.py", "ref_mu")
bb = load("models/backbone.py", "ref_bb")
torch.set_num_threads(8)
out = {}
for tag, kw in (("mrcnn", {}), ("retina_unet", {"operate_stride1": True})):
    cf = bi.make_cf(**kw)
    fpn = bb.FPN(cf, mu.NDConvGenerator(cf.dim), operate_stride1=kw.get("operate_stride1", False)) if kw else bb.FPN(cf, mu.NDConvGenerator(cf.dim))
    fpn.eval()
    bi.fill_by_name(fpn)
    with torch.no_grad():
        outs = fpn(torch.from_numpy(bi.make_input()))
    for i, o in enumerate(outs):
        a = o.numpy().reshape(-1)
        out["%s_level%d" % (tag, i)] = a[bi.sample_index(a.size)].copy()
        out["%s_level%d_shape" % (tag, i)] = np.array(o.shape)
        out["%s_level%d_maxabs" % (tag, i)] = np.float32(np.abs(a).max())
        print(tag, i, tuple(o.shape), float(np.abs(a).max()), float(a.std()))
np.savez_compressed(os.path.join(HERE, "backbone_reference.npz"), **out)
print(os.path.getsize(os.path.join(HERE, "backbone_reference.npz")))
