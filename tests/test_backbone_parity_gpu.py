"""Backbone numerics: the FPN / ResNet conv path of this repo on MIOpen / CK (fp32,
with and without channels_last_3d, with the exhaustive solver search the bench uses) against the REFERENCE's FPN run by
torch on the CPU in fp32 (tests/golden/backbone_reference.npz, tests/golden/make_backbone_golden.py), same name-seeded
weights.  Bar: 1e-3 of the level's max-abs (fp32 accumulation-order differences through ~50 conv layers)."""
import os

import numpy as np
import pytest
import torch

from tests.golden import backbone_inputs as bi

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "backbone_reference.npz"))


@pytest.mark.parametrize("tag,channels_last,benchmark", [("mrcnn", False, False), ("mrcnn", True, True), ("retina_unet", True, True)])
def test_fpn_vs_reference_cpu(cuda, tag, channels_last, benchmark):
    from medicaldetectiontoolkit_amd.models import backbone as bb
    from medicaldetectiontoolkit_amd.utils.model_utils import NDConvGenerator
    prev = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = benchmark
    try:
        kw = {"operate_stride1": True} if tag == "retina_unet" else {}
        cf = bi.make_cf(**kw)
        fpn = bb.FPN(cf, NDConvGenerator(cf.dim), **kw).eval()
        bi.fill_by_name(fpn)
        fpn = fpn.to(cuda)
        x = torch.from_numpy(bi.make_input()).to(cuda)
        if channels_last:
            fpn = fpn.to(memory_format=torch.channels_last_3d)
            x = x.contiguous(memory_format=torch.channels_last_3d)
        with torch.no_grad():
            outs = fpn(x)
        assert len(outs) == sum(1 for k in G.files if k.startswith(tag + "_level") and k.endswith("_shape"))
        for i, o in enumerate(outs):
            assert list(o.shape) == G["%s_level%d_shape" % (tag, i)].tolist()
            a = o.float().contiguous().cpu().numpy().reshape(-1)
            got = a[bi.sample_index(a.size)]
            want = G["%s_level%d" % (tag, i)]
            bound = 1e-3 * float(G["%s_level%d_maxabs" % (tag, i)])
            assert np.abs(got - want).max() <= bound, (tag, i, np.abs(got - want).max(), bound)
    finally:
        torch.backends.cudnn.benchmark = prev
