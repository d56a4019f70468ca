"""Writes tests/golden/state_dict_keys.json: parameter names + shapes of the REFERENCE's modules (imported from
/root/reference with the four cuda_functions modules stubbed), for checkpoint compatibility tests.
Run once in the build container:  timeout 300 python tests/golden/make_state_dict_golden.py"""
import importlib.util
import json
import os
import sys
import types
import warnings

warnings.filterwarnings("ignore")
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from medicaldetectiontoolkit_amd.configs import Configs  # noqa: E402

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
mr = load("models/mrcnn.py", "ref_mrcnn")
ru = load("models/retina_unet.py", "ref_retina")
out = {}
for tag, kw in (("mrcnn3d", dict(dim=3, model="mrcnn")), ("mrcnn2d", dict(dim=2, model="mrcnn")),
                ("retina_unet3d", dict(dim=3, model="retina_unet")), ("retina_net2d", dict(dim=2, model="retina_net"))):
    cf = Configs(**kw)
    conv = mu.NDConvGenerator(cf.dim)
    mods = {}
    if "mrcnn" in tag:
        mods = {"fpn": bb.FPN(cf, conv), "rpn": mr.RPN(cf, conv), "classifier": mr.Classifier(cf, conv), "mask": mr.Mask(cf, conv)}
    else:
        mods = {"Fpn": bb.FPN(cf, conv, operate_stride1=cf.operate_stride1), "Classifier": ru.Classifier(cf, conv),
                "BBRegressor": ru.BBRegressor(cf, conv)}
        if cf.model == "retina_unet":
            mods["final_conv"] = conv(cf.end_filts, cf.num_seg_classes, ks=1, pad=0, norm=None, relu=None)
    keys = {}
    for prefix, m in mods.items():
        for k, v in m.state_dict().items():
            keys[prefix + "." + k] = list(v.shape)
    out[tag] = keys
    print(tag, len(keys), "tensors", sum(int(__import__("numpy").prod(s)) for s in keys.values()), "parameters")
json.dump(out, open(os.path.join(HERE, "state_dict_keys.json"), "w"), indent=0, sort_keys=True)
