"""Host-side packing logic that needs no GPU."""
import numpy as np
import torch


def test_gt_on_device_packing_matches_padded_tables():
    """models/mrcnn.GtOnDevice (one upload for all GT boxes / class ids) against a direct construction of the padded
    tables the target layer and the RPN losses consume (mrcnn.py:487-500 semantics: elements without a foreground class id
    are treated as having no GT in the target layer but keep their boxes for the anchor matching)."""
    from medicaldetectiontoolkit_amd.models.mrcnn import GtOnDevice
    rng = np.random.default_rng(0)
    boxes = [rng.uniform(0, 60, size=(3, 6)), np.zeros((0, 6)), rng.uniform(0, 60, size=(1, 6)), rng.uniform(0, 60, size=(2, 6))]
    cls = [np.array([1, 2, 1]), np.array([], dtype=np.int64), np.array([0]), np.array([2, 1])]
    g = GtOnDevice(boxes, cls, 3, torch.device("cpu"))
    assert g.n_all == [3, 0, 1, 2] and g.counts == [3, 0, 0, 2]
    assert g.px.shape == (4, 3, 6) and g.px.dtype == torch.float64 and g.px.is_contiguous()
    for b in range(4):
        n = len(boxes[b])
        assert np.array_equal(g.px[b, :n].numpy(), boxes[b])
        assert (g.px[b, n:] == 0).all()
        assert g.cls[b, :n].tolist() == cls[b].tolist() and g.cls_i32.dtype == torch.int32
    assert g.valid.tolist() == [[True, True, True], [False] * 3, [False] * 3, [True, True, False]]
    # index into the stacked GT masks: runs over ALL objects of the batch, -1 where not valid
    assert g.gidx.tolist() == [[0, 1, 2], [-1, -1, -1], [-1, -1, -1], [4, 5, -1]]


def test_const_tensor_is_cached_per_value_dtype_device():
    from medicaldetectiontoolkit_amd.utils import model_utils as mutils
    a = mutils.const_tensor([0.1, 0.2], torch.float32, torch.device("cpu"))
    b = mutils.const_tensor([0.1, 0.2], torch.float32, torch.device("cpu"))
    c = mutils.const_tensor([0.1, 0.2], torch.float64, torch.device("cpu"))
    d = mutils.const_tensor([[0.1], [0.2]], torch.float32, torch.device("cpu"))
    assert a is b and c is not a and c.dtype == torch.float64 and d.shape == (2, 1) and d is not a
    assert torch.equal(a, torch.tensor([0.1, 0.2]))


def test_conv_reformulations_are_exact_identities_in_float64():
    """the two MIOpen-problem reformulations of utils/fused_epilogue (input gradient as a forward convolution with the
    flipped/transposed filter; stem in space-to-depth form) are identities: float64 on the CPU shows it to 1e-12"""
    import torch.nn.functional as F
    from medicaldetectiontoolkit_amd.utils.fused_epilogue import _ConvStem221, _ConvStride1
    torch.manual_seed(0)
    for nd, ks, pad in ((3, 3, 1), (3, 1, 0), (2, 3, 1), (2, 5, 2)):
        x = torch.randn((2, 5) + (9, 8, 7)[:nd], dtype=torch.float64, requires_grad=True)
        w = torch.randn((4, 5) + (ks,) * nd, dtype=torch.float64, requires_grad=True)
        conv = F.conv3d if nd == 3 else F.conv2d
        y1, y2 = _ConvStride1.apply(x, w, (pad,) * nd), conv(x, w, None, 1, pad)
        g = torch.randn_like(y2)
        a, b = torch.autograd.grad(y1, (x, w), g), torch.autograd.grad(y2, (x, w), g)
        assert torch.equal(y1, y2) and (a[0] - b[0]).abs().max() < 1e-12 and (a[1] - b[1]).abs().max() < 1e-12
    for cin, k in ((1, 7), (2, 3), (3, 5)):
        x = torch.randn(2, cin, 16, 12, 10, dtype=torch.float64, requires_grad=True)
        w = torch.randn(5, cin, k, k, k, dtype=torch.float64, requires_grad=True)
        y1, y2 = _ConvStem221.apply(x, w), F.conv3d(x, w, None, (2, 2, 1), k // 2)
        g = torch.randn_like(y2)
        a, b = torch.autograd.grad(y1, (x, w), g), torch.autograd.grad(y2, (x, w), g)
        assert (y1 - y2).abs().max() < 1e-12 and (a[0] - b[0]).abs().max() < 1e-12 and (a[1] - b[1]).abs().max() < 1e-12
