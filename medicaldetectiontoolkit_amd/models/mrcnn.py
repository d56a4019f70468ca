"""Mask R-CNN (2D / 3D) on the gfx950 hot-path kernels.

Mirror of the reference's models/mrcnn.py: same module names / state_dict keys (fpn, rpn, classifier,
mask), same `net(cf, logger)` with train_forward / test_forward / forward returning the same
results_dict.  What changed is the glue between the kernels:
  * proposal_layer (mrcnn.py:297-369): top-k, fused decode+clip+score kernel, ONE batched
    device-resident NMS over the batch with early stop at proposal_count; no D2H, no per-element sync.
  * pyramid_roi_align (:373-457): ONE forward launch and ONE backward launch for all pyramid levels (every RoI is
    pooled on its own level and written to its own row) -- no per-level loop, no nonzero(), no gather/sort-back.
  * all GT boxes / class ids of a step go to the device in one pinned asynchronous copy before the backbone is
    launched (GtOnDevice); small constants are cached on the device: nothing inside a step waits for the stream.
  * detection_target_layer (:461-613), refine_detections (:620-714) and the RPN losses (:176-240) are
    expressed on fixed-size, masked tensors (random keys + top-k for the random sub-sampling, SHEM via
    sorted pools), so a training step has no host synchronisation before the final loss read-out.
Random sub-sampling uses torch's device RNG instead of numpy / CPU randperm, so individual samples
differ from the reference while the sampling distribution is the same.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from ..cuda_functions import _roi_align_impl as _rai_mod
from ..cuda_functions._roi_align_impl import pyramid_crop_and_resize
from ..cuda_functions.roi_align_2D.roi_align.crop_and_resize import CropAndResizeFunction as ra2D
from ..cuda_functions.roi_align_3D.roi_align.crop_and_resize import CropAndResizeFunction as ra3D
from ..utils import fused_epilogue
from ..utils import model_utils as mutils
from . import backbone as backbone_module


############################################################
# Networks on top of backbone (state_dict-compatible with the reference)
############################################################
HEAD_AS_LINEAR = True    # module switch (A/B: bench.py --head-as-linear 0): classifier-head convolutions as matrix products
MERGE_RPN_HEADS = True   # module switch (A/B: bench.py --merge-rpn-heads 0): conv_class and conv_bbox of the RPN as ONE 1x1 convolution
SHARED_PYRAMID_GRAD = True   # module switch (A/B: bench.py --shared-pyramid-grad 0): the RoI heads' RoIAlign backward launches and the sampled-anchor RPN scatter write
                             # ONE gradient buffer per pyramid map (cuda_functions/_roi_align_impl.PyramidGradAccumulator) instead of three that autograd adds
FUSED_GLUE = True        # module switch (A/B: bench.py --fused-glue 0): level rule, RPN sampling, box targets and the detection target layer as single
                         # launches of csrc/glue.hip instead of chains of small tensor operations (same arithmetic, same random keys)
SPARSE_RPN_LOSS = True   # module switch (A/B: bench.py --sparse-rpn-loss 0): RPN losses differentiate through the SAMPLED anchors only


class RPN(nn.Module):
    """Region Proposal Network (mrcnn.py:40-86)."""

    def __init__(self, cf, conv):
        super(RPN, self).__init__()
        self.dim = conv.dim
        self.conv_shared = conv(cf.end_filts, cf.n_rpn_features, ks=3, stride=cf.rpn_anchor_stride, pad=1, relu=cf.relu)
        self.conv_class = conv(cf.n_rpn_features, 2 * len(cf.rpn_anchor_ratios), ks=1, stride=1, relu=None)
        self.conv_bbox = conv(cf.n_rpn_features, 2 * self.dim * len(cf.rpn_anchor_ratios), ks=1, stride=1, relu=None)

    def forward(self, x):
        x = self.conv_shared(x)
        axes = (0, 2, 3, 1) if self.dim == 2 else (0, 2, 3, 4, 1)
        if MERGE_RPN_HEADS and isinstance(self.conv_class, fused_epilogue.ConvBias) and isinstance(self.conv_bbox, fused_epilogue.ConvBias) \
                and x.is_cuda and x.dtype == torch.float32:
            # both heads are 1x1 convolutions of the SAME 128-channel map (mrcnn.py:70-77; 537 MB on P2 at 8 x 128^3): as two layers
            # the map is read twice forward, twice for the weight gradients, and its gradient is produced twice (two 537 MB
            # tensors + an add).  One convolution with the concatenated filters [2A + 2*dim*A, C] reads / writes it once each way;
            # the parameters stay separate (state-dict keys unchanged), autograd splits the gradient of the concatenation.
            nc = self.conv_class.out_channels
            w = torch.cat([self.conv_class.weight, self.conv_bbox.weight], 0)
            b = torch.cat([self.conv_class.bias, self.conv_bbox.bias], 0)
            y = fused_epilogue.bias_act(fused_epilogue.conv_unit_stride(x, w, self.conv_class.padding), b)
            y = y.permute(*axes)
            rpn_class_logits = y[..., :nc].contiguous().view(x.size(0), -1, 2)
            rpn_bbox = y[..., nc:].contiguous().view(x.size(0), -1, self.dim * 2)
        else:
            rpn_class_logits = self.conv_class(x).permute(*axes).contiguous().view(x.size(0), -1, 2)
            rpn_bbox = self.conv_bbox(x).permute(*axes).contiguous().view(x.size(0), -1, self.dim * 2)
        rpn_probs = F.softmax(rpn_class_logits, dim=2)
        return [rpn_class_logits, rpn_probs, rpn_bbox]


RPN_HEADS_FUSED = True   # module switch (A/B: bench.py --rpn-heads-fused 0): the dense, gradient-free RPN forward runs both heads on the raw conv_shared output in one
                         # launch per level that writes straight into the concatenated outputs (mdt_rpn_heads_forward)


def rpn_levels_fused(rpn, feature_maps):
    """[logits, probs, deltas] of the RPN over all pyramid levels, concatenated along the anchor axis as mrcnn.py:1030-1033 builds them -- without autograd
    graph: per level conv_shared's raw output goes through ONE kernel (bias + ReLU of conv_shared, both 1x1 heads, their biases, the slicing and the
    concatenation; csrc/conv1x1_fwd.hip rpn_heads_mfma_kernel), then one softmax over the concatenated logits.  None when the case is not the kernel's
    (the caller runs RPN.forward per level)."""
    from .. import _lib
    cs = rpn.conv_shared
    if not (isinstance(cs, fused_epilogue.ConvBiasReLU) and isinstance(rpn.conv_class, fused_epilogue.ConvBias) and isinstance(rpn.conv_bbox, fused_epilogue.ConvBias)):
        return None
    conv = cs[0]
    nd = rpn.dim
    mf = torch.channels_last_3d if nd == 3 else torch.channels_last
    hidden, ncl, nbox = int(conv.out_channels), int(rpn.conv_class.out_channels), int(rpn.conv_bbox.out_channels)
    L = _lib.lib()
    if conv.bias is None or not all(int(v) == 1 for v in conv.stride) or not L.mdt_rpn_heads_forward_supported(hidden, ncl, nbox) \
            or not all(int(k) == 1 for k in rpn.conv_class.kernel_size) or not all(int(k) == 1 for k in rpn.conv_bbox.kernel_size):
        return None
    if not all(m.is_cuda and m.dtype == torch.float32 and m.dim() == nd + 2 and m.is_contiguous(memory_format=mf) and not m.is_contiguous() for m in feature_maps) \
            or torch.is_autocast_enabled():
        return None
    A = ncl // 2
    B = int(feature_maps[0].shape[0])
    with torch.no_grad():
        vox = [int(m.shape[2:].numel()) for m in feature_maps]
        total = sum(vox) * A
        dev = feature_maps[0].device
        logits = torch.empty(B, total, 2, dtype=torch.float32, device=dev)
        deltas = torch.empty(B, total, nbox // A, dtype=torch.float32, device=dev)
        w = torch.cat([rpn.conv_class.weight.reshape(ncl, hidden), rpn.conv_bbox.weight.reshape(nbox, hidden)], 0)
        b = torch.cat([rpn.conv_class.bias, rpn.conv_bbox.bias], 0)
        off = 0
        for m, v in zip(feature_maps, vox):
            h = fused_epilogue._conv(conv, m)
            if not h.is_contiguous(memory_format=mf):
                h = h.contiguous(memory_format=mf)
            rc = L.mdt_rpn_heads_forward(h.data_ptr(), conv.bias.data_ptr(), w.data_ptr(), b.data_ptr(), logits.data_ptr(), deltas.data_ptr(), B, v, hidden, ncl, nbox,
                                         total, off, _lib.raw_stream())
            if rc == _lib.MDT_ERR_UNSUPPORTED:
                return None
            if rc != 0:
                _lib.check(rc, "mdt_rpn_heads_forward")
            off += v * A
        probs = F.softmax(logits, dim=2)
    return [logits, probs, deltas]


def rpn_sparse_supported(rpn):
    """rpn_at_anchors restates the RPN for single voxels: 3^dim unit-stride conv_shared (+ ReLU / LeakyReLU, no norm layer) and 1^dim heads"""
    cs = rpn.conv_shared
    if not isinstance(cs, nn.Sequential) or len(cs) != 2 or not isinstance(cs[1], (nn.ReLU, nn.LeakyReLU)):
        return False
    Conv = nn.Conv2d if rpn.dim == 2 else nn.Conv3d

    def plain(c, k, pad):
        return isinstance(c, Conv) and c.groups == 1 and not isinstance(c.padding, str) and all(int(v) == k for v in c.kernel_size) \
            and all(int(v) == 1 for v in c.stride) and all(int(v) == 1 for v in c.dilation) and all(int(v) == pad for v in c.padding) \
            and c.padding_mode == "zeros"
    return plain(cs[0], 3, 1) and plain(rpn.conv_class, 1, 0) and plain(rpn.conv_bbox, 1, 0)


class _GatherRows(torch.autograd.Function):
    """rows of a [V, C] matrix by index; backward = index_add_ into zeros (advanced indexing's backward sorts the indices first: a dozen
    launches for a few thousand rows)"""

    @staticmethod
    def forward(ctx, flat, rows):
        ctx.save_for_backward(rows)
        ctx.n = flat.shape[0]
        return flat.index_select(0, rows)

    @staticmethod
    def backward(ctx, g):
        rows, = ctx.saved_tensors
        out = torch.zeros((ctx.n, g.shape[1]), dtype=g.dtype, device=g.device)
        out.index_add_(0, rows, g.contiguous())
        return out, None


class _RpnPatches(torch.autograd.Function):
    """the 3^dim x C neighbourhoods of the sampled anchors from channels-last pyramid maps: ONE gather launch (mdt_rpn_patch_gather) instead of
    ~50 index-arithmetic / masked-gather tensor operations, ONE scatter launch backward (float atomics into zero-filled gradient maps)"""

    @staticmethod
    def forward(ctx, idx_flat, A, n_per_elem, *maps):
        import ctypes
        L = _lib.lib()
        dim = maps[0].dim() - 2
        C = int(maps[0].shape[1])
        S = int(idx_flat.shape[0])
        T = 3 ** dim
        dev = maps[0].device
        patches = torch.empty((S, T, C), dtype=torch.float32, device=dev)
        k_anchor = torch.empty(S, dtype=torch.int64, device=dev)
        Y = (ctypes.c_int * len(maps))(*[int(m.shape[2]) for m in maps])
        X = (ctypes.c_int * len(maps))(*[int(m.shape[3]) for m in maps])
        Z = (ctypes.c_int * len(maps))(*[int(m.shape[4]) if dim == 3 else 1 for m in maps])
        ptrs = (ctypes.c_void_p * len(maps))(*[_lib.ptr(m) for m in maps])
        with torch.cuda.device(dev):
            rc = L.mdt_rpn_patch_gather(len(maps), ptrs, Y, X, Z, dim, C, int(A), _lib.ptr(idx_flat), S, int(n_per_elem), _lib.ptr(patches), _lib.ptr(k_anchor),
                                        _lib.current_stream_ptr())
        _lib.check(rc, "mdt_rpn_patch_gather")
        ctx.save_for_backward(idx_flat)
        ctx.meta = (int(A), int(n_per_elem), dim, C, [tuple(m.shape) for m in maps], (Y, X, Z))
        ctx.mark_non_differentiable(k_anchor)
        acc = _rai_mod.PyramidGradAccumulator.CURRENT        # one gradient buffer per map for all consumers of the step (see there)
        ctx.acc = None
        if acc is not None and acc.matches(maps) and any(ctx.needs_input_grad[3:]):
            acc.register()
            ctx.acc = acc
        return patches, k_anchor

    @staticmethod
    def backward(ctx, g, _gk):
        import ctypes
        idx_flat, = ctx.saved_tensors
        A, n_per_elem, dim, C, shapes, (Y, X, Z) = ctx.meta
        mf = torch.channels_last_3d if dim == 3 else torch.channels_last
        g = g.contiguous()

        def scatter(outs, row_major):
            # deterministic: one writer per voxel, rows that share a voxel added in row order (mdt_rpn_patch_scatter_add_ordered) -- float atomics made the
            # step's gradients differ in the last bit from run to run (neighbourhoods of sampled anchors overlap)
            ptrs = (ctypes.c_void_p * len(outs))(*[_lib.ptr(o) for o in outs])
            ids = torch.empty(int(idx_flat.shape[0]) * (3 ** dim), dtype=torch.int64, device=g.device)
            with torch.cuda.device(g.device):
                rc = _lib.lib().mdt_rpn_patch_scatter_add_ordered(len(outs), ptrs, 1 if row_major else 0, Y, X, Z, dim, C, A, _lib.ptr(idx_flat),
                                                                  int(idx_flat.shape[0]), n_per_elem, _lib.ptr(g), _lib.ptr(ids), _lib.current_stream_ptr())
            _lib.check(rc, "mdt_rpn_patch_scatter_add_ordered")
        if ctx.acc is not None:         # into the step's shared (row-major) buffers, after a RoIAlign backward has written them
            outs = ctx.acc.sparse(lambda bufs: scatter(bufs, True))
            return (None, None, None) + (tuple(outs) if outs is not None else (None,) * len(shapes))
        outs = [torch.empty(sh, dtype=torch.float32, device=g.device, memory_format=mf).zero_() for sh in shapes]
        scatter(outs, False)
        return (None, None, None) + tuple(outs)


def _rpn_patches_fused_ok(feature_maps, dim):
    mf = torch.channels_last_3d if dim == 3 else torch.channels_last
    C = int(feature_maps[0].shape[1])
    return FUSED_GLUE and len(feature_maps) <= 8 and C % 4 == 0 and all(
        m.is_cuda and m.dtype == torch.float32 and m.dim() == dim + 2 and int(m.shape[1]) == C and m.is_contiguous(memory_format=mf) for m in feature_maps)


def rpn_at_anchors(rpn, feature_maps, idx, n_anchors_per_voxel):
    """The RPN (mrcnn.py:40-86) evaluated at CHOSEN anchors only: idx [B, n] indexes the anchors in the order of the concatenated pyramid
    levels, as RPN.forward lays them out ((y, x, z, anchor) row-major per level).  Returns class logits [B, n, 2] and box deltas
    [B, n, 2 dim], differentiable w.r.t. the feature maps and the RPN's parameters.

    Why: the RPN losses (mrcnn.py:176-240) read the dense outputs at rpn_train_anchors_per_image = 6 anchors per element -- 48 of
    3.6 M anchors at 8 x 128^3 -- so the gradient that the dense graph sends back through the heads and conv_shared is zero at all but
    <= 48 voxels, and the reference (and this repo until round 4) still runs the full input-gradient and weight-gradient convolutions
    of conv_shared on every level for it: 6.6 ms on P2 alone, a sixth of the step.  Here the dense forward runs without a graph (the
    proposals and the SHEM ranking need every anchor, mrcnn.py:243-330 detaches them anyway) and the sampled anchors are recomputed
    from their 3^dim x C neighbourhoods: gather -> [S, 3^dim C] x [3^dim C, F] -> activation -> the two 1x1 heads.  Same function of
    the same parameters, so the same gradients up to fp32 summation order (test_models_gpu.py pins both against the dense graph)."""
    dim = rpn.dim
    conv = rpn.conv_shared[0]
    B, n = idx.shape
    dev = idx.device
    A = int(n_anchors_per_voxel)
    sizes = [tuple(int(v) for v in m.shape[2:]) for m in feature_maps]
    vox = [int(np.prod(sz)) for sz in sizes]
    starts = np.concatenate([[0], np.cumsum([v * A for v in vox])])
    idx = idx.long()
    if _rpn_patches_fused_ok(feature_maps, dim):
        S, T, C = B * n, 3 ** dim, conv.in_channels
        patches, k_anchor = _RpnPatches.apply(idx.reshape(-1).contiguous(), A, n, *feature_maps)
        return _rpn_heads_on_patches(rpn, conv, patches, k_anchor, B, n, S, T, C, A, dim)
    b_ix = torch.arange(B, device=dev)[:, None].expand(B, n).reshape(-1)
    flat_idx = idx.reshape(-1)
    S = B * n
    T = 3 ** dim
    C = conv.in_channels
    offs = mutils.const_tensor([[t // (3 ** (dim - 1 - d)) % 3 - 1 for d in range(dim)] for t in range(T)], torch.int64, dev)      # [T, dim], (ky, kx[, kz]) order
    # index arithmetic ONCE for all levels (per-sample level, extents and start through small lookup tables) ...
    L = len(sizes)
    starts_t = mutils.const_tensor([int(v) for v in starts], torch.int64, dev)
    size_t = mutils.const_tensor([list(sz) for sz in sizes], torch.int64, dev)  # [L, dim]
    level = torch.bucketize(flat_idx, starts_t[1:L], right=True)               # [S] in 0 .. L-1
    local = flat_idx - starts_t[level]
    sz = size_t[level]                                                          # [S, dim]
    v = local // A
    k_anchor = local - v * A
    coords = []
    for d in range(dim - 1, -1, -1):
        q = v // sz[:, d]
        coords.append(v - q * sz[:, d])
        v = q
    coords = torch.stack(coords[::-1], 1)                                       # [S, dim]
    nb = coords[:, None, :] + offs[None, :, :]                                 # [S, T, dim]
    lim = sz[:, None, :]
    ok = ((nb >= 0) & (nb < lim)).all(-1)                                       # the zero padding of the convolution
    nbc = torch.minimum(nb.clamp(min=0), lim - 1)
    row = b_ix[:, None]
    for d in range(dim):
        row = row * sz[:, d:d + 1] + nbc[..., d]                                # [S, T] row inside the sample's own level
    # ... then one masked gather per level (rows of the other levels read row 0 and are multiplied by zero)
    patches = None
    for l, fm in enumerate(feature_maps):
        m = ok & (level == l)[:, None]
        perm = (0, 2, 3, 1) if dim == 2 else (0, 2, 3, 4, 1)
        flat = fm.permute(*perm).reshape(-1, C)                                # a view for channels-last maps
        g = _GatherRows.apply(flat, (row * m).reshape(-1)).view(S, T, C) * m.unsqueeze(-1).to(fm.dtype)
        patches = g if patches is None else patches + g
    return _rpn_heads_on_patches(rpn, conv, patches, k_anchor, B, n, S, T, C, A, dim)


def _rpn_heads_on_patches(rpn, conv, patches, k_anchor, B, n, S, T, C, A, dim):
    """conv_shared as a matrix product over the gathered (tap, channel) patches, activation, the two 1x1 heads, the sampled anchor's rows"""
    wperm = (0, 2, 3, 1) if dim == 2 else (0, 2, 3, 4, 1)
    w = conv.weight.permute(*wperm).reshape(conv.out_channels, T * C)          # (tap, channel) order of the patches
    h = F.linear(patches.reshape(S, T * C), w, conv.bias)
    act = rpn.conv_shared[1]
    h = F.leaky_relu(h, act.negative_slope) if isinstance(act, nn.LeakyReLU) else F.relu(h)
    Fh = conv.out_channels
    lc = F.linear(h, rpn.conv_class.weight.reshape(-1, Fh), rpn.conv_class.bias).view(S, A, 2)
    lb = F.linear(h, rpn.conv_bbox.weight.reshape(-1, Fh), rpn.conv_bbox.bias).view(S, A, 2 * dim)
    logits = torch.gather(lc, 1, k_anchor.view(S, 1, 1).expand(S, 1, 2)).view(B, n, 2)
    deltas = torch.gather(lb, 1, k_anchor.view(S, 1, 1).expand(S, 1, 2 * dim)).view(B, n, 2 * dim)
    return logits, deltas


def _forward_valid_rows(fn, feature_maps, rois):
    """The glue keeps fixed-size tensors whose padding rows carry batch_ix = -1 and pool to all-zero features.  With
    cf.norm == 'batch_norm' those rows would enter the heads' batch statistics (the reference only ever forwards real
    samples, mrcnn.py:1075-1076), so in that configuration the heads run on the valid rows only and the outputs are
    scattered back -- at the price of one host sync (nonzero); norm=None / instance_norm never take this path."""
    keep = torch.nonzero(rois[:, -1] >= 0)[:, 0]
    outs = fn(feature_maps, rois[keep])
    full = []
    for o in outs:
        z = torch.zeros((rois.shape[0],) + tuple(o.shape[1:]), dtype=o.dtype, device=o.device)
        full.append(z.index_copy(0, keep, o))
    return full


class Classifier(nn.Module):
    """Classification + box-refinement head (mrcnn.py:89-127)."""

    def __init__(self, cf, conv):
        super(Classifier, self).__init__()
        self.dim = conv.dim
        self.in_channels = cf.end_filts
        self.pool_size = cf.pool_size
        self.pyramid_levels = cf.pyramid_levels
        norm = cf.norm if cf.norm != "instance_norm" else None
        self.conv1 = conv(cf.end_filts, cf.end_filts * 4, ks=self.pool_size, stride=1, norm=norm, relu=cf.relu)
        self.conv2 = conv(cf.end_filts * 4, cf.end_filts * 4, ks=1, stride=1, norm=norm, relu=cf.relu)
        self.linear_class = nn.Linear(cf.end_filts * 4, cf.head_classes)
        self.linear_bbox = nn.Linear(cf.end_filts * 4, cf.head_classes * 2 * self.dim)
        self.compact_rows = norm == "batch_norm"

    def forward(self, x, rois):
        if self.compact_rows:
            return _forward_valid_rows(self._forward, x, rois)
        return self._forward(x, rois)

    def _forward(self, x, rois):
        x = pyramid_roi_align(x, rois, self.pool_size, self.pyramid_levels, self.dim)
        if HEAD_AS_LINEAR and isinstance(self.conv1, fused_epilogue.ConvBiasReLU) and isinstance(self.conv2, fused_epilogue.ConvBiasReLU) \
                and tuple(self.conv1[0].kernel_size) == tuple(x.shape[2:]) and not any(self.conv1[0].padding):
            # conv1's kernel IS the pooled extent (mrcnn.py:104) and conv2 is 1x1(x1) on a single voxel: both are matrix products
            # [n, C * ph * pw * pd] x [.., 4C] -- MIOpen's convolution kernels take 1.2 ms for conv1's backward on 48 RoIs
            # (tools/op_profile.py), the GEMM path microseconds.  Same sums up to fp32 order; weights keep their conv shape.
            c1, c2 = self.conv1[0], self.conv2[0]
            x = F.relu(F.linear(x.flatten(1), c1.weight.flatten(1), c1.bias))
            x = F.relu(F.linear(x, c2.weight.flatten(1), c2.bias))
        else:
            x = self.conv2(self.conv1(x))
        x = x.view(-1, self.in_channels * 4)
        mrcnn_class_logits = self.linear_class(x)
        mrcnn_bbox = self.linear_bbox(x)
        return [mrcnn_class_logits, mrcnn_bbox.view(mrcnn_bbox.size(0), -1, self.dim * 2)]


class Mask(nn.Module):
    """Mask head (mrcnn.py:130-169)."""

    def __init__(self, cf, conv):
        super(Mask, self).__init__()
        self.pool_size = cf.mask_pool_size
        self.pyramid_levels = cf.pyramid_levels
        self.dim = conv.dim
        self.conv1 = conv(cf.end_filts, cf.end_filts, ks=3, stride=1, pad=1, norm=cf.norm, relu=cf.relu)
        self.conv2 = conv(cf.end_filts, cf.end_filts, ks=3, stride=1, pad=1, norm=cf.norm, relu=cf.relu)
        self.conv3 = conv(cf.end_filts, cf.end_filts, ks=3, stride=1, pad=1, norm=cf.norm, relu=cf.relu)
        self.conv4 = conv(cf.end_filts, cf.end_filts, ks=3, stride=1, pad=1, norm=cf.norm, relu=cf.relu)
        Deconv = nn.ConvTranspose2d if conv.dim == 2 else nn.ConvTranspose3d
        self.deconv = Deconv(cf.end_filts, cf.end_filts, kernel_size=2, stride=2)
        self.relu = nn.ReLU(inplace=True) if cf.relu == "relu" else nn.LeakyReLU(inplace=True)
        self.conv5 = conv(cf.end_filts, cf.head_classes, ks=1, stride=1, relu=None)
        self.sigmoid = nn.Sigmoid()
        self.compact_rows = cf.norm == "batch_norm"

    def forward(self, x, rois):
        if self.compact_rows:
            return _forward_valid_rows(lambda a, b: [self._forward(a, b)], x, rois)[0]
        return self._forward(x, rois)

    def _forward(self, x, rois):
        x = pyramid_roi_align(x, rois, self.pool_size, self.pyramid_levels, self.dim)
        x = self.conv4(self.conv3(self.conv2(self.conv1(x))))
        dc = self.deconv
        if isinstance(self.relu, nn.ReLU) and dc.bias is not None and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
            # bias + ReLU of the transposed convolution in one in-place pass (torch runs the bias add and the ReLU as two passes over the
            # [N, 36, 28, 28, 10] map; same parameters, same state-dict keys)
            fn = F.conv_transpose3d if self.dim == 3 else F.conv_transpose2d
            x = fused_epilogue.bias_act(fn(x, dc.weight, None, dc.stride, dc.padding, dc.output_padding, dc.groups, dc.dilation), dc.bias, None, True)
        else:
            x = self.relu(self.deconv(x))
        return self.sigmoid(self.conv5(x))


############################################################
#  Helper layers
############################################################
def _masked_mean(values, mask):
    """mean of `values` over entries where mask is True; 0 if none (mirrors the `0 not in ...size()` guards)."""
    m = mask.to(values.dtype)
    while m.dim() < values.dim():
        m = m.unsqueeze(-1)
    m = m.expand_as(values)
    return (values * m).sum() / m.sum().clamp(min=1.0)


def proposal_layer(rpn_pred_probs, rpn_pred_deltas, proposal_count, anchors, cf):
    """mrcnn.py:297-369.  anchors: [A, 2*dim] fp32 device.  Returns
    batch_normalized_boxes [B, proposal_count, 2*dim] and batch_out_proposals [B, proposal_count, 2*dim+1]
    (pixel boxes + RPN fg score; zero rows pad missing proposals) -- both DEVICE tensors."""
    L = _lib.lib()
    B, A = rpn_pred_probs.shape[0], rpn_pred_probs.shape[1]
    dim = rpn_pred_deltas.shape[-1] // 2
    dev = rpn_pred_probs.device
    pre_nms_limit = min(cf.pre_nms_limit, A)
    # top pre_nms_limit anchors by fg score, best first (:328-333)
    scores, order = torch.topk(rpn_pred_probs[:, :, 1].detach(), pre_nms_limit, dim=1, sorted=True)
    deltas = rpn_pred_deltas.detach()
    dets = torch.empty((B, pre_nms_limit, 2 * dim + 1), dtype=torch.float32, device=dev)
    for b in range(B):   # fused gather + decode + clip + score append (:337-344); launches only, no sync
        dets[b] = mutils.decode_clip_boxes(anchors, deltas[b], cf.rpn_bbox_std_dev, cf.window, order=order[b], scores=scores[b])
    keep = torch.empty((B, proposal_count), dtype=torch.int64, device=dev)
    num = torch.empty(B, dtype=torch.int32, device=dev)
    wsb = B * L.mdt_nms_workspace_bytes(pre_nms_limit)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    fn = L.mdt_nms_3d_batched if dim == 3 else L.mdt_nms_2d_batched
    import ctypes
    with torch.cuda.device(dev):
        rc = fn(_lib.ptr(dets), B, pre_nms_limit, ctypes.c_float(cf.rpn_nms_threshold), _lib.NMS_RULE_GT, proposal_count,
                _lib.ptr(keep), proposal_count, _lib.ptr(num), _lib.ptr(ws), wsb, _lib.current_stream_ptr())
    _lib.check(rc, "mdt_nms_batched")
    valid = keep >= 0                                            # rows beyond num_out are -1: zero-padded (:352-358)
    gathered = torch.gather(dets, 1, keep.clamp(min=0).unsqueeze(-1).expand(-1, -1, 2 * dim + 1))
    batch_out_proposals = gathered * valid.unsqueeze(-1).to(gathered.dtype)
    norm = mutils.const_tensor(cf.scale, torch.float32, dev)
    batch_normalized_boxes = batch_out_proposals[:, :, :2 * dim] / norm
    return batch_normalized_boxes, batch_out_proposals


def pyramid_roi_align(feature_maps, rois, pool_size, pyramid_levels, dim):
    """mrcnn.py:373-457.  rois [n, 2*dim + 1] (normalised box, batch_ix; batch_ix < 0 marks a padding row).
    Level rule :403 (h*w of the normalised box only).  All levels are pooled by ONE kernel launch (and one backward
    launch): every RoI reads its own level's map and writes its own output row, so there is no per-level loop, no
    nonzero()/gather/sort-back and no host sync."""
    if FUSED_GLUE and rois.is_cuda and rois.dtype == torch.float32 and rois.dim() == 2 and rois.shape[1] == 2 * dim + 1:
        r = rois.detach().contiguous()
        n = int(r.shape[0])
        boxes = torch.empty((n, 2 * dim), dtype=torch.float32, device=r.device)
        ints = torch.empty((2, n), dtype=torch.int32, device=r.device)
        with torch.cuda.device(r.device):
            rc = _lib.lib().mdt_roi_levels(_lib.ptr(r), n, dim, int(pyramid_levels[0]), int(pyramid_levels[-1]), 1 if len(pyramid_levels) == 5 else 0,
                                           _lib.ptr(boxes), _lib.ptr(ints[0]), _lib.ptr(ints[1]), _lib.current_stream_ptr())
        _lib.check(rc, "mdt_roi_levels")
        return pyramid_crop_and_resize(list(feature_maps), boxes, ints[0], ints[1], pool_size)
    boxes = rois[:, :dim * 2].detach()
    batch_ixs = rois[:, dim * 2]
    h = boxes[:, 2] - boxes[:, 0]
    w = boxes[:, 3] - boxes[:, 1]
    roi_level = (4 + mutils.log2(torch.sqrt(h * w))).round().int().clamp(pyramid_levels[0], pyramid_levels[-1])
    if len(pyramid_levels) == 5:
        roi_level = torch.where(h * w > 0.65, torch.full_like(roi_level, 5), roi_level)
    return pyramid_crop_and_resize(list(feature_maps), boxes, batch_ixs.to(torch.int32), roi_level - int(pyramid_levels[0]),
                                   pool_size)


class GtOnDevice(object):
    """GT boxes / class ids of a batch on the device, sent with ONE pinned, asynchronous upload at the start of the step.
    (Every separate upload from pageable numpy memory waits for the stream: the per-element uploads of the matching and
    target code used to drain the launch queue a dozen times per step.)
      px     [B, Gmax, 2*dim] f64 pixel boxes of ALL objects (anchor matching, RPN delta targets)
      cls    [B, Gmax] i64 class ids of all objects
      valid  [B, Gmax] bool, gidx [B, Gmax] i32 (index into the stacked GT masks): set for elements with at least one
             foreground class id, like the target layer expects (mrcnn.py:487)
      n_gt   [B] i32 DEVICE tensor: objects per element (read by the batched matching kernel -- no host count enters a launch)
      n_all  python list: objects per element;  counts: 0 for elements without a foreground class id else n_all
    The whole content is one float64 table [B, Gmax, 2*dim + 4] (box, class id, valid flag, mask index, n_all of the element):
    `stage()` builds it in pinned host memory, `GtOnDevice(table=...)` derives the views from a table that is already on the device
    (training.GraphedTrainStep copies the staged table into a STATIC device table and captures the derivation).  `gmax` fixes the
    padded object count (static shapes for hipGraph capture); default: the batch's own maximum."""

    @staticmethod
    def stage(batch_gt_boxes, batch_gt_class_ids, dim, gmax=None, pin=True):
        B = len(batch_gt_boxes)
        n_all = [0 if g is None else len(g) for g in batch_gt_boxes]
        counts = [0 if (n_all[b] == 0 or not np.any(np.asarray(batch_gt_class_ids[b]) > 0)) else n_all[b] for b in range(B)]
        if gmax is None:
            gmax = max(1, max(n_all))
        elif max(n_all) > gmax:
            raise ValueError("a batch element has %d GT objects, the fixed-size GT table holds %d (cf.max_gt_per_element)" % (max(n_all), gmax))
        stage = torch.zeros((B, gmax, 2 * dim + 4), dtype=torch.float64, pin_memory=bool(pin))
        a = stage.numpy()
        a[:, :, 2 * dim + 2] = -1.0
        run = 0
        for b in range(B):
            n = n_all[b]
            a[b, :, 2 * dim + 3] = n
            if n:
                a[b, :n, :2 * dim] = np.asarray(batch_gt_boxes[b], dtype=np.float64)
                a[b, :n, 2 * dim] = np.asarray(batch_gt_class_ids[b])
            if counts[b]:
                a[b, :n, 2 * dim + 1] = 1.0
                a[b, :n, 2 * dim + 2] = run + np.arange(n)
            run += n
        return stage, n_all, counts

    def __init__(self, batch_gt_boxes, batch_gt_class_ids, dim, dev, gmax=None, table=None):
        if table is None:
            stage, self.n_all, self.counts = self.stage(batch_gt_boxes, batch_gt_class_ids, dim, gmax, pin=False)
            if torch.device(dev).type == "cuda":
                # through the table's own persistent pinned ring (a fresh pinned block per step = a hipHostMalloc and an implicit device
                # sync per step as soon as the host runs ahead of the GPU: torch's caching host allocator re-issues a block only
                # after its copy has executed)
                table = mutils.stage_pinned(stage, dev, "gt_table").to(dev, non_blocking=True)
                mutils.stage_release(dev, "gt_table")
            else:
                table = stage
        else:
            self.n_all = self.counts = None       # host-side counts are not known (and not needed) for a device table
        d = table
        self.table = d
        self.px = d[:, :, :2 * dim].contiguous()
        self.cls = d[:, :, 2 * dim].long()
        self.cls_i32 = d[:, :, 2 * dim].int()
        self.valid = d[:, :, 2 * dim + 1] > 0
        self.gidx = d[:, :, 2 * dim + 2].int()
        self.n_gt = d[:, 0, 2 * dim + 3].int()


def _pad_gt(batch_gt_boxes, batch_gt_class_ids, scale, dim, dev, gt_dev=None):
    """padded device tensors [B, Gmax, ...] of the target layer: normalised fp32 boxes, class ids, validity, mask index"""
    g = gt_dev if gt_dev is not None else GtOnDevice(batch_gt_boxes, batch_gt_class_ids, dim, dev)
    return g.px.float() / scale, g.cls, g.valid, g.gidx, g.counts


def _detection_targets_fused(batch_proposals, scores, batch_gt_masks, cf, B, pc, P, pool_max, Nn, generator, g):
    """detection_target_layer in ONE launch (mdt_detection_targets: overlaps, sampling, class / box targets) + the GT-mask crop; the same two
    torch.rand draws as the tensor form ([B, pc] and [B, pool_max]), so both forms sample the same RoIs"""
    L = _lib.lib()
    dev = batch_proposals.device
    dim = cf.dim
    S = P + Nn
    rois = batch_proposals.detach()
    rois = rois if rois.is_contiguous() else rois.contiguous()
    sc = scores.detach()
    sc = sc if sc.is_contiguous() else sc.contiguous()
    rand_pos = torch.rand((B, pc), device=dev, generator=generator)
    rand_pool = torch.rand((B, pool_max), device=dev, generator=generator)
    sample_indices = torch.empty(B * S, dtype=torch.int64, device=dev)
    flags = torch.empty((2, B * S), dtype=torch.bool, device=dev)
    tcls = torch.empty(B * S, dtype=torch.int64, device=dev)
    tdel = torch.empty((B * S, 2 * dim), dtype=torch.float32, device=dev)
    pos_rois = torch.empty((B * P, 2 * dim), dtype=torch.float32, device=dev)
    box_ids = torch.empty(B * P, dtype=torch.int32, device=dev)
    scale = mutils.const_tensor(cf.scale, torch.float32, dev)
    std = mutils.const_tensor(cf.bbox_std_dev, torch.float32, dev)
    pos_thr, neg_thr = (0.5, 0.1) if dim == 2 else (0.3, 0.01)
    import ctypes
    gvalid = g.valid if g.valid.is_contiguous() else g.valid.contiguous()
    with torch.cuda.device(dev):
        rc = L.mdt_detection_targets(_lib.ptr(rois), int(rois.shape[1]), _lib.ptr(sc), int(sc.shape[1]), _lib.ptr(g.px), _lib.ptr(scale),
                                     _lib.ptr(g.cls), _lib.ptr(gvalid), _lib.ptr(g.gidx), _lib.ptr(rand_pos), _lib.ptr(rand_pool), _lib.ptr(std),
                                     B, pc, int(g.px.shape[1]), dim, P, pool_max, Nn, int(cf.shem_poolsize),
                                     ctypes.c_float(pos_thr), ctypes.c_float(neg_thr), ctypes.c_float(1.0 / cf.roi_positive_ratio),
                                     _lib.ptr(sample_indices), _lib.ptr(flags[0]), _lib.ptr(flags[1]), _lib.ptr(tcls), _lib.ptr(tdel),
                                     _lib.ptr(pos_rois), _lib.ptr(box_ids), None, _lib.current_stream_ptr())
    _lib.check(rc, "mdt_detection_targets")
    target_masks = torch.zeros((B, S) + tuple(cf.mask_shape), dtype=torch.float32, device=dev)
    if batch_gt_masks is not None and batch_gt_masks.shape[0] > 0:
        ra = ra2D(cf.mask_shape[0], cf.mask_shape[1], 0) if dim == 2 else ra3D(cf.mask_shape[0], cf.mask_shape[1], cf.mask_shape[2], 0)
        with torch.no_grad():
            gm = batch_gt_masks if batch_gt_masks.dtype in (torch.uint8, torch.float32) else batch_gt_masks.float()
            masks = ra(gm, pos_rois, box_ids).squeeze(1)
            torch.round(masks.view((B, P) + tuple(cf.mask_shape)), out=target_masks[:, :P])
    return sample_indices, flags[0], flags[1], tcls, tdel, target_masks.view((-1,) + tuple(cf.mask_shape))


def detection_target_layer(batch_proposals, batch_mrcnn_class_scores, batch_gt_class_ids, batch_gt_boxes,
                           batch_gt_masks, cf, B, generator=None, gt_dev=None):
    """mrcnn.py:461-613 on fixed-size masked tensors.
    batch_proposals [B*pc, 2*dim+1]; batch_gt_masks: device float/uint8 tensor [sum_G, 1, Y, X, (Z)] with the GT
    masks of all batch elements stacked in order (the reference gathers per-RoI copies, :551).
    Returns sample_indices [B*S], valid [B*S], target_class_ids [B*S], target_deltas [B*S, 2*dim],
    target_masks [B*S, *mask_shape] with S = train_rois_per_image slots per element (positives then negatives)."""
    dev = batch_proposals.device
    dim = cf.dim
    pc = batch_proposals.shape[0] // B
    scale = mutils.const_tensor(cf.scale, torch.float32, dev)
    n_pos_max = int(cf.train_rois_per_image * cf.roi_positive_ratio)
    n_neg_max = max(int((1.0 / cf.roi_positive_ratio) * n_pos_max - n_pos_max), 1)
    Pq, pool_q = min(n_pos_max, pc), min(cf.shem_poolsize * n_neg_max, pc)
    Nq = min(n_neg_max, pool_q)
    if FUSED_GLUE and batch_proposals.is_cuda and batch_proposals.dtype == torch.float32 and batch_mrcnn_class_scores.dtype == torch.float32:
        if gt_dev is None:
            gt_dev = GtOnDevice(batch_gt_boxes, batch_gt_class_ids, dim, dev)
        if _lib.lib().mdt_detection_targets_supported(pc, int(gt_dev.px.shape[1]), Pq, pool_q, Nq):
            return _detection_targets_fused(batch_proposals, batch_mrcnn_class_scores, batch_gt_masks, cf, B, pc, Pq, pool_q, Nq, generator, gt_dev)
    gt_boxes, gt_cls, gt_valid, gt_gidx, counts = _pad_gt(batch_gt_boxes, batch_gt_class_ids, scale, dim, dev, gt_dev)
    proposals = batch_proposals[:, :2 * dim].detach().view(B, pc, 2 * dim)
    has_gt = gt_valid.any(1)                                                          # [B]

    overlaps = mutils.bbox_overlaps(proposals, gt_boxes)                              # [B, pc, Gmax], all elements at once
    overlaps = torch.where(gt_valid[:, None, :], overlaps, torch.full_like(overlaps, -1.0))
    roi_iou_max, roi_gt_assign = overlaps.max(dim=2)                                  # [B, pc]
    pos_thr, neg_thr = (0.5, 0.1) if dim == 2 else (0.3, 0.01)
    positive = (roi_iou_max >= pos_thr) & has_gt[:, None]
    negative = torch.where(has_gt[:, None], roi_iou_max < neg_thr, torch.ones_like(positive))

    n_pos_max = int(cf.train_rois_per_image * cf.roi_positive_ratio)
    n_neg_max = max(int((1.0 / cf.roi_positive_ratio) * n_pos_max - n_pos_max), 1)
    # positives: random subset of size <= n_pos_max (randperm in the reference, :532-534)
    key = torch.where(positive, torch.rand(positive.shape, device=dev, generator=generator), torch.full(positive.shape, -1.0, device=dev))
    pkey, pidx = torch.topk(key, min(n_pos_max, pc), dim=1)
    pvalid = pkey >= 0
    pos_count = pvalid.sum(1)                                                         # [B]
    # negatives via SHEM (:579-585): pool = best poolsize*count negatives by max fg prob, `count` random ones of it
    r = 1.0 / cf.roi_positive_ratio
    neg_count = torch.clamp((r * pos_count.float() - pos_count.float()).long(), min=1)
    fg = batch_mrcnn_class_scores.detach()[:, 1:].max(1)[0].view(B, pc)
    nscore = torch.where(negative, fg, torch.full_like(fg, -1.0))
    pool_max = min(cf.shem_poolsize * n_neg_max, pc)
    pool_score, pool_idx = torch.topk(nscore, pool_max, dim=1)                        # sorted, best first
    rank = torch.arange(pool_max, device=dev)[None, :]
    in_pool = (pool_score >= 0) & (rank < (cf.shem_poolsize * neg_count)[:, None])
    key2 = torch.where(in_pool, torch.rand(in_pool.shape, device=dev, generator=generator), torch.full(in_pool.shape, -1.0, device=dev))
    nkey, nsel = torch.topk(key2, min(n_neg_max, pool_max), dim=1)
    nidx = torch.gather(pool_idx, 1, nsel)
    nvalid = (nkey >= 0) & (torch.arange(nkey.shape[1], device=dev)[None, :] < neg_count[:, None])

    # positive targets
    pos_rois = torch.gather(proposals, 1, pidx.unsqueeze(-1).expand(-1, -1, 2 * dim))  # [B, P, 2dim]
    pos_assign = torch.gather(roi_gt_assign, 1, pidx)                                 # [B, P]
    pos_gt_boxes = torch.gather(gt_boxes, 1, pos_assign.unsqueeze(-1).expand(-1, -1, 2 * dim))
    pos_cls = torch.gather(gt_cls, 1, pos_assign)
    dummy = mutils.const_tensor([0., 0., 1., 1., 0., 1.] if dim == 3 else [0., 0., 1., 1.], torch.float32, dev)   # keeps log() finite
    safe_rois = torch.where(pvalid.unsqueeze(-1), pos_rois, dummy.expand_as(pos_rois))
    safe_gt = torch.where(pvalid.unsqueeze(-1), pos_gt_boxes, safe_rois)
    std = mutils.const_tensor(cf.bbox_std_dev, torch.float32, dev)
    deltas = mutils.box_refinement(safe_rois.view(-1, 2 * dim), safe_gt.view(-1, 2 * dim)) / std   # [B*P, 2dim]
    # mask targets (:551-563): crop the assigned GT mask with the positive RoI, threshold at 0.5
    P = pidx.shape[1]
    gsel = torch.gather(gt_gidx, 1, pos_assign)
    box_ids = torch.where(pvalid, gsel, torch.full_like(gsel, -1)).view(-1).to(torch.int32)
    ra = ra2D(cf.mask_shape[0], cf.mask_shape[1], 0) if dim == 2 else ra3D(cf.mask_shape[0], cf.mask_shape[1], cf.mask_shape[2], 0)
    if batch_gt_masks is not None and batch_gt_masks.shape[0] > 0:
        with torch.no_grad():
            # uint8 masks are read as uint8 (csrc/roi_align.hip u8 instantiation): no fp32 copy of the stacked 128^3 masks
            gm = batch_gt_masks if batch_gt_masks.dtype in (torch.uint8, torch.float32) else batch_gt_masks.float()
            masks = torch.round(ra(gm, pos_rois.view(-1, 2 * dim).contiguous(), box_ids).squeeze(1))
    else:
        masks = torch.zeros((B * P,) + tuple(cf.mask_shape), device=dev)

    Nn = nidx.shape[1]
    base = (torch.arange(B, device=dev) * pc)[:, None]
    sample_indices = torch.cat([(pidx + base), (nidx + base)], 1).view(-1)            # [B*(P+Nn)]
    valid = torch.cat([pvalid, nvalid], 1).view(-1)
    is_pos = torch.cat([pvalid, torch.zeros_like(nvalid)], 1).view(-1)
    target_class_ids = torch.cat([torch.where(pvalid, pos_cls, torch.zeros_like(pos_cls)),
                                  torch.zeros((B, Nn), dtype=pos_cls.dtype, device=dev)], 1).view(-1)
    target_deltas = torch.cat([deltas.view(B, P, 2 * dim), torch.zeros((B, Nn, 2 * dim), device=dev)], 1).view(-1, 2 * dim)
    target_masks = torch.cat([masks.view((B, P) + tuple(cf.mask_shape)),
                              torch.zeros((B, Nn) + tuple(cf.mask_shape), device=dev)], 1).view((-1,) + tuple(cf.mask_shape))
    target_deltas = target_deltas * is_pos.unsqueeze(-1).to(target_deltas.dtype)
    return sample_indices, valid, is_pos, target_class_ids, target_deltas, target_masks


def _refine_detections_fused(rois, probs, deltas, cf, B, pc, fg, std, M):
    """refine_detections as decode + sort (one launch), the batched NMS, top-M + row assembly (two launches): csrc/glue.hip"""
    import ctypes
    L = _lib.lib()
    dev = rois.device
    dim = cf.dim
    nc = int(probs.shape[1])
    rois_c = rois.detach().contiguous()
    probs_c = probs.detach().contiguous()
    deltas_c = deltas.detach().contiguous()
    G = B * fg
    scale = np.ascontiguousarray(cf.scale, dtype=np.float32)
    win = np.ascontiguousarray(cf.window, dtype=np.float32)
    thr = ctypes.c_float(cf.model_min_confidence)
    dets = torch.empty((G, pc, 2 * dim + 1), dtype=torch.float32, device=dev)
    keep = torch.empty((G, pc), dtype=torch.int64, device=dev)
    num = torch.empty(G + B, dtype=torch.int32, device=dev)              # NMS counts | any_valid scratch
    wsb = G * L.mdt_nms_workspace_bytes(pc)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    result = torch.empty((B * M, 2 * dim + 3), dtype=torch.float32, device=dev)
    valid = torch.empty(B * M, dtype=torch.bool, device=dev)
    fn = L.mdt_nms_3d_batched if dim == 3 else L.mdt_nms_2d_batched
    with torch.cuda.device(dev):
        st = _lib.current_stream_ptr()
        rc = L.mdt_refine_detections_pre(_lib.ptr(rois_c), _lib.ptr(probs_c), _lib.ptr(deltas_c), std.ctypes.data, scale.ctypes.data, win.ctypes.data, thr,
                                         B, pc, dim, nc, _lib.ptr(dets), st)
        _lib.check(rc, "mdt_refine_detections_pre")
        rc = fn(_lib.ptr(dets), G, pc, ctypes.c_float(cf.detection_nms_threshold), _lib.NMS_RULE_GT, 0,
                _lib.ptr(keep), pc, _lib.ptr(num), _lib.ptr(ws), wsb, st)
        _lib.check(rc, "mdt_nms_batched")
        rc = L.mdt_refine_detections_post(_lib.ptr(rois_c), _lib.ptr(probs_c), _lib.ptr(deltas_c), std.ctypes.data, scale.ctypes.data, win.ctypes.data, thr,
                                          B, pc, dim, nc, M, _lib.ptr(dets), _lib.ptr(keep), _lib.ptr(result), _lib.ptr(valid), _lib.ptr(num[G:]), st)
        _lib.check(rc, "mdt_refine_detections_post")
    return result, valid


def refine_detections(rois, probs, deltas, batch_ixs, cf, B):
    """mrcnn.py:620-714 on fixed-size tensors.  Returns detections [B*M, 2*dim+3] = (pixel box rounded, batch_ix,
    class_id, score) with M = model_max_instances_per_batch_element slots per element, and a validity mask.
    One batched device NMS over all (element, class) groups replaces the per-element x per-class loop."""
    L = _lib.lib()
    import ctypes
    dev = rois.device
    dim = cf.dim
    n = rois.shape[0]
    pc = n // B
    fg = cf.head_classes - 1
    std = np.asarray(cf.rpn_bbox_std_dev, dtype=np.float32)       # quirk 8: rpn_bbox_std_dev, not bbox_std_dev (:650)
    M = cf.model_max_instances_per_batch_element
    if FUSED_GLUE and rois.is_cuda and rois.dtype == torch.float32 and probs.dtype == torch.float32 and deltas.dtype == torch.float32 \
            and L.mdt_refine_detections_supported(pc, int(probs.shape[1]), min(M, fg * pc)):
        return _refine_detections_fused(rois, probs, deltas, cf, B, pc, fg, std, min(M, fg * pc))
    scale = mutils.const_tensor(cf.scale, torch.float32, dev)
    win = [float(v) for v in cf.window]
    no_clip = [-3e38, -3e38, 3e38, 3e38] + ([-3e38, 3e38] if dim == 3 else [])
    groups_boxes, groups_scores = [], []
    for c in range(1, fg + 1):
        dec = mutils.decode_clip_boxes(rois, deltas[:, c, :].contiguous(), std, no_clip) * scale
        dec = torch.round(mutils.clip_boxes(dec, win))
        groups_boxes.append(dec.view(B, pc, 2 * dim))
        groups_scores.append(probs[:, c].view(B, pc))
    boxes = torch.stack(groups_boxes, 1).view(B * fg, pc, 2 * dim)          # group g = b*fg + (c-1)
    scores = torch.stack(groups_scores, 1).view(B * fg, pc)
    ok = scores >= cf.model_min_confidence
    s_sorted, order = torch.sort(torch.where(ok, scores, torch.full_like(scores, -1.0)), dim=1, descending=True, stable=True)
    b_sorted = torch.gather(boxes, 1, order.unsqueeze(-1).expand(-1, -1, 2 * dim))
    dets = torch.cat([b_sorted, s_sorted.unsqueeze(-1)], 2).contiguous()
    G = B * fg
    keep = torch.empty((G, pc), dtype=torch.int64, device=dev)
    num = torch.empty(G, dtype=torch.int32, device=dev)
    wsb = G * L.mdt_nms_workspace_bytes(pc)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    fn = L.mdt_nms_3d_batched if dim == 3 else L.mdt_nms_2d_batched
    with torch.cuda.device(dev):
        rc = fn(_lib.ptr(dets), G, pc, ctypes.c_float(cf.detection_nms_threshold), _lib.NMS_RULE_GT, 0,
                _lib.ptr(keep), pc, _lib.ptr(num), _lib.ptr(ws), wsb, _lib.current_stream_ptr())
    _lib.check(rc, "mdt_nms_batched")
    kept = torch.zeros((G, pc + 1), dtype=torch.bool, device=dev)
    kept.scatter_(1, torch.where(keep >= 0, keep, torch.full_like(keep, pc)), True)
    kept = kept[:, :pc] & (s_sorted >= cf.model_min_confidence)
    # top-k per batch element over its classes (:701-703)
    M = cf.model_max_instances_per_batch_element
    cand = torch.where(kept, s_sorted, torch.full_like(s_sorted, -1.0)).view(B, fg * pc)
    top_s, top_i = torch.topk(cand, min(M, fg * pc), dim=1)
    valid = top_s >= 0
    cls_ids = (top_i // pc + 1).float()
    top_b = torch.gather(b_sorted.view(B, fg * pc, 2 * dim), 1, top_i.unsqueeze(-1).expand(-1, -1, 2 * dim))
    bix = torch.arange(B, device=dev, dtype=torch.float32)[:, None].expand_as(top_s)
    result = torch.cat([top_b, bix.unsqueeze(-1), cls_ids.unsqueeze(-1), top_s.unsqueeze(-1)], 2)
    result = result * valid.unsqueeze(-1).to(result.dtype)
    # mrcnn.py:708-709: when NO roi of the whole batch reaches model_min_confidence the reference keeps index 0 of its repeated arrays --
    # roi 0 of element 0 with class 1 and its (sub-threshold) score -- and get_results (:717-799) emits it unfiltered.  Same here, decided
    # on the device (no host sync): slot 0 takes that row when nothing is valid.
    none = ~valid.any()
    fallback = torch.cat([groups_boxes[0][0, 0], mutils.const_tensor([0.0, 1.0], torch.float32, dev), groups_scores[0][0, :1]])
    result = result.view(-1, 2 * dim + 3)
    valid = valid.reshape(-1).clone()
    result[0] = torch.where(none, fallback, result[0])
    valid[0] = valid[0] | none
    return result, valid


############################################################
#  Loss functions (masked, fixed-size)
############################################################
def _rpn_sample_fused(rpn_match, rpn_class_logits, n_pos_max, poolsize, generator):
    """the sampling of compute_rpn_losses in two launches of csrc/glue.hip (mdt_rpn_sample) -- the SAME two torch.rand draws as the tensor
    form below (shapes [B, A] and [B, kpool]), so both forms pick the same anchors"""
    L = _lib.lib()
    dev = rpn_match.device
    B, A = rpn_match.shape
    K = int(rpn_class_logits.shape[-1])
    kpool = min(poolsize * n_pos_max, A)
    rand_pos = torch.rand((B, A), device=dev, generator=generator)
    rand_pool = torch.rand((B, kpool), device=dev, generator=generator)
    idx = torch.empty((3, B, n_pos_max), dtype=torch.int64, device=dev)          # pidx, nidx, tgt_pos
    flags = torch.empty((2, B, n_pos_max), dtype=torch.bool, device=dev)         # pvalid, nvalid
    pos_count = torch.empty(B, dtype=torch.int64, device=dev)
    wsb = L.mdt_rpn_sample_workspace_bytes(B, A, n_pos_max, kpool)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    logits = rpn_class_logits.detach()
    if not logits.is_contiguous():
        logits = logits.contiguous()
    match = rpn_match if rpn_match.is_contiguous() else rpn_match.contiguous()
    with torch.cuda.device(dev):
        rc = L.mdt_rpn_sample(_lib.ptr(match), _lib.ptr(logits), K, _lib.ptr(rand_pos), _lib.ptr(rand_pool), B, A, n_pos_max, poolsize, kpool,
                              _lib.ptr(idx[0]), _lib.ptr(flags[0]), _lib.ptr(idx[1]), _lib.ptr(flags[1]), _lib.ptr(pos_count), _lib.ptr(idx[2]),
                              _lib.ptr(ws), wsb, _lib.current_stream_ptr())
    _lib.check(rc, "mdt_rpn_sample")
    return idx[0], flags[0], idx[1], flags[1], pos_count, idx[2]


def _rpn_losses_from_samples(samples, rpn_match, rpn_argmax, rpn_class_logits, rpn_pred_deltas, anchors_f64, gt_boxes_list, cf, gt_dev, sparse_eval,
                             n_pos_max):
    """the loss terms of compute_rpn_losses on the anchors mdt_rpn_sample drew; box targets from mdt_anchor_delta_targets"""
    pidx, pvalid, nidx, nvalid, pos_count, tgt_pos = samples
    dev = rpn_class_logits.device
    B, A = rpn_match.shape
    dim = cf.dim
    K = rpn_class_logits.shape[-1]
    if sparse_eval is None:
        logits_pos = torch.gather(rpn_class_logits, 1, pidx.unsqueeze(-1).expand(-1, -1, K))
        logits_neg = torch.gather(rpn_class_logits, 1, nidx.unsqueeze(-1).expand(-1, -1, K))
        pred = torch.gather(rpn_pred_deltas, 1, pidx.unsqueeze(-1).expand(-1, -1, 2 * dim))
    else:
        ls, ds = sparse_eval(torch.cat([pidx, nidx], 1))                       # ONE evaluation for positives and negatives
        logits_pos, logits_neg, pred = ls[:, :n_pos_max], ls[:, n_pos_max:], ds[:, :n_pos_max]
    # both cross-entropies in one call: rows [positives | negatives], targets [class id | 0]
    tgt = torch.cat([tgt_pos, torch.zeros_like(tgt_pos)], 1)
    ce = F.cross_entropy(torch.cat([logits_pos, logits_neg], 1).reshape(-1, K), tgt.view(-1), reduction="none").view(B, 2 * n_pos_max)
    w = torch.cat([pvalid, nvalid], 1).to(ce.dtype)
    sums = (ce * w).view(B, 2, n_pos_max).sum(2)                               # [B, 2]: positive / negative sums
    cnt = w.view(B, 2, n_pos_max).sum(2).clamp(min=1)
    class_loss = ((sums / cnt).sum(1) / 2).mean()
    if gt_dev is not None:
        gt_pad = gt_dev.px
    else:
        gmax = max(1, max(len(g) for g in gt_boxes_list))
        gt_np = np.zeros((B, gmax, 2 * dim), dtype=np.float64)
        for b, g in enumerate(gt_boxes_list):
            if len(g) > 0:
                gt_np[b, :len(g)] = np.asarray(g, dtype=np.float64)
        gt_pad = torch.from_numpy(gt_np).to(dev, non_blocking=True)
    tgt_d = torch.empty((B, n_pos_max, 2 * dim), dtype=torch.float32, device=dev)
    std = mutils.const_tensor(cf.rpn_bbox_std_dev, torch.float64, dev)
    argmax = rpn_argmax if (rpn_argmax.dtype == torch.int32 and rpn_argmax.is_contiguous()) else rpn_argmax.to(torch.int32).contiguous()
    gt_pad = gt_pad if gt_pad.is_contiguous() else gt_pad.contiguous()
    with torch.cuda.device(dev):
        rc = _lib.lib().mdt_anchor_delta_targets(_lib.ptr(anchors_f64), _lib.ptr(gt_pad), _lib.ptr(argmax), _lib.ptr(pidx), _lib.ptr(pvalid), _lib.ptr(std),
                                                 B, A, int(gt_pad.shape[1]), n_pos_max, dim, _lib.ptr(tgt_d), _lib.current_stream_ptr())
    _lib.check(rc, "mdt_anchor_delta_targets")
    sl1 = F.smooth_l1_loss(pred, tgt_d, reduction="none")
    bbox_loss_b = (sl1 * pvalid.unsqueeze(-1)).sum((1, 2)) / (pos_count.clamp(min=1) * 2 * dim)
    bbox_loss = bbox_loss_b.mean()
    return class_loss, bbox_loss, (pidx, pvalid, nidx, nvalid)


def compute_rpn_losses(rpn_match, rpn_argmax, rpn_class_logits, rpn_pred_deltas, anchors_f64, gt_boxes_list, cf, generator=None,
                       shem_poolsize=None, gt_dev=None, sparse_eval=None):
    """compute_rpn_class_loss (mrcnn.py:176-214) + compute_rpn_bbox_loss (:217-240), batched over B; with K-class
    logits and class-id matches it is also retina_unet.compute_class_loss / compute_bbox_loss (retina_unet.py:126-189).
    rpn_match [B, A] int32 (-1 / 0 / >0) as returned by the matching kernel BEFORE sub-sampling; the
    sub-sampling of surplus positives (model_utils.py:566-571) and SHEM are done here with random keys.
    sparse_eval (idx [B, n] -> logits [B, n, K], deltas [B, n, 2 dim]; rpn_at_anchors): the dense tensors then carry no graph and only
    rank / select, the loss terms differentiate through the re-evaluated samples."""
    dev = rpn_class_logits.device
    B, A = rpn_match.shape
    dim = cf.dim
    n_pos_max = max(cf.rpn_train_anchors_per_image // 2, 1)
    K = rpn_class_logits.shape[-1]
    poolsize = cf.shem_poolsize if shem_poolsize is None else shem_poolsize
    fused = None
    if FUSED_GLUE and rpn_match.is_cuda and rpn_match.dtype == torch.int32 and rpn_class_logits.dtype == torch.float32 and \
            _lib.lib().mdt_rpn_sample_supported(A, n_pos_max, min(poolsize * n_pos_max, A)):
        fused = _rpn_sample_fused(rpn_match, rpn_class_logits, n_pos_max, poolsize, generator)
    if fused is not None:
        return _rpn_losses_from_samples(fused, rpn_match, rpn_argmax, rpn_class_logits, rpn_pred_deltas, anchors_f64, gt_boxes_list, cf, gt_dev, sparse_eval,
                                        n_pos_max)
    pos = rpn_match > 0
    key = torch.where(pos, torch.rand(pos.shape, device=dev, generator=generator), torch.full(pos.shape, -1.0, device=dev))
    pkey, pidx = torch.topk(key, n_pos_max, dim=1)
    pvalid = pkey >= 0                                                        # [B, n_pos_max]
    pos_count = pvalid.sum(1)
    K = rpn_class_logits.shape[-1]
    poolsize = cf.shem_poolsize if shem_poolsize is None else shem_poolsize
    if sparse_eval is None:
        logits_pos = torch.gather(rpn_class_logits, 1, pidx.unsqueeze(-1).expand(-1, -1, K))
    tgt_pos = torch.gather(rpn_match, 1, pidx).clamp(min=0).long()          # 1 for the RPN, class id for Retina
    # negatives: SHEM over anchors labelled -1
    neg = rpn_match == -1
    neg_count = pos_count.clamp(min=1)
    fgp = F.softmax(rpn_class_logits.detach(), dim=2)[:, :, 1:].max(dim=2)[0]
    pool_max = poolsize * n_pos_max
    pool_score, pool_idx = torch.topk(torch.where(neg, fgp, torch.full_like(fgp, -1.0)), min(pool_max, A), dim=1)
    rank = torch.arange(pool_score.shape[1], device=dev)[None, :]
    in_pool = (pool_score >= 0) & (rank < (poolsize * neg_count)[:, None])
    key2 = torch.where(in_pool, torch.rand(in_pool.shape, device=dev, generator=generator), torch.full(in_pool.shape, -1.0, device=dev))
    nkey, nsel = torch.topk(key2, n_pos_max, dim=1)
    nidx = torch.gather(pool_idx, 1, nsel)
    nvalid = (nkey >= 0) & (torch.arange(n_pos_max, device=dev)[None, :] < neg_count[:, None])
    if sparse_eval is None:
        logits_neg = torch.gather(rpn_class_logits, 1, nidx.unsqueeze(-1).expand(-1, -1, K))
        pred = torch.gather(rpn_pred_deltas, 1, pidx.unsqueeze(-1).expand(-1, -1, 2 * dim))
    else:
        ls, ds = sparse_eval(torch.cat([pidx, nidx], 1))                       # ONE evaluation for positives and negatives
        logits_pos, logits_neg, pred = ls[:, :n_pos_max], ls[:, n_pos_max:], ds[:, :n_pos_max]
    ce_pos = F.cross_entropy(logits_pos.reshape(-1, K), tgt_pos.view(-1), reduction="none").view(B, -1)
    pos_loss = (ce_pos * pvalid).sum(1) / pos_count.clamp(min=1)              # 0 when no positive
    ce_neg = F.cross_entropy(logits_neg.reshape(-1, K), torch.zeros(B * n_pos_max, dtype=torch.long, device=dev), reduction="none").view(B, -1)
    neg_loss = (ce_neg * nvalid).sum(1) / nvalid.sum(1).clamp(min=1)
    class_loss = ((pos_loss + neg_loss) / 2).mean()                           # mean over batch == sum(loss_b / B)

    # bbox: smooth-L1 between predicted deltas of the kept positives and their targets
    if gt_dev is not None:
        gt_pad = gt_dev.px
    else:
        gmax = max(1, max(len(g) for g in gt_boxes_list))
        gt_pad = np.zeros((B, gmax, 2 * dim), dtype=np.float64)
        for b, g in enumerate(gt_boxes_list):
            if len(g) > 0:
                gt_pad[b, :len(g)] = np.asarray(g, dtype=np.float64)
        gt_pad = torch.from_numpy(gt_pad).to(dev, non_blocking=True)
    a_pos = anchors_f64[pidx.view(-1)]                                        # [B*n, 2dim] f64
    g_assign = torch.gather(rpn_argmax.long(), 1, pidx)
    g_pos = torch.gather(gt_pad, 1, g_assign.unsqueeze(-1).expand(-1, -1, 2 * dim)).view(-1, 2 * dim)
    pv = pvalid.view(-1)
    g_pos = torch.where(pv.unsqueeze(-1), g_pos, a_pos)                       # keep invalid rows finite
    tgt = mutils.anchor_delta_targets(a_pos, g_pos, cf.rpn_bbox_std_dev).float().view(B, n_pos_max, 2 * dim)
    sl1 = F.smooth_l1_loss(pred, tgt, reduction="none")
    bbox_loss_b = (sl1 * pvalid.unsqueeze(-1)).sum((1, 2)) / (pos_count.clamp(min=1) * 2 * dim)
    bbox_loss = bbox_loss_b.mean()
    return class_loss, bbox_loss, (pidx, pvalid, nidx, nvalid)


def compute_mrcnn_class_loss(target_class_ids, pred_class_logits, valid):
    ce = F.cross_entropy(pred_class_logits, target_class_ids.long(), reduction="none")
    return _masked_mean(ce, valid)


def compute_mrcnn_bbox_loss(mrcnn_target_deltas, mrcnn_pred_deltas, target_class_ids, is_pos):
    idx = torch.arange(mrcnn_pred_deltas.shape[0], device=mrcnn_pred_deltas.device)
    pred = mrcnn_pred_deltas[idx, target_class_ids.long().clamp(min=0)]
    sl1 = F.smooth_l1_loss(pred, mrcnn_target_deltas.detach(), reduction="none")
    return _masked_mean(sl1, is_pos)


def compute_mrcnn_mask_loss(target_masks, pred_masks, target_class_ids, is_pos):
    idx = torch.arange(pred_masks.shape[0], device=pred_masks.device)
    y_pred = pred_masks[idx, target_class_ids.long().clamp(min=0)]
    bce = F.binary_cross_entropy(y_pred, target_masks.detach(), reduction="none")
    return _masked_mean(bce, is_pos)


############################################################
#  Output handler
############################################################
def get_results(cf, img_shape, detections, det_valid, detection_masks, box_results_list=None, return_masks=True):
    """mrcnn.py:717-799: restore the batch dimension, unmold, fill the results dict."""
    dv = det_valid if isinstance(det_valid, np.ndarray) else det_valid.detach().cpu().numpy()
    det = (detections if isinstance(detections, np.ndarray) else detections.detach().cpu().numpy())[dv]
    dim = cf.dim
    if box_results_list is None:
        box_results_list = [[] for _ in range(img_shape[0])]
    masks_np = None
    if return_masks and detection_masks is not None:
        perm = (0, 2, 3, 1) if dim == 2 else (0, 2, 3, 4, 1)
        masks_np = detection_masks.permute(*perm).detach().cpu().numpy()[dv]
    batch_ixs = det[:, dim * 2] if det.shape[0] else np.zeros(0)
    seg_preds = []
    for ix in range(img_shape[0]):
        sel = batch_ixs == ix
        d = det[sel]
        final_masks = None           # stays None unless masks are pasted: the all-zero volume is then made once, as uint8 (the reference
        if d.shape[0] > 0:           # builds an fp64 volume per element and rounds / casts the stack: 35 ms per 8 x 128^3 step)
            boxes = d[:, :2 * dim].astype(np.int32)
            class_ids = d[:, 2 * dim + 1].astype(np.int32)
            scores = d[:, 2 * dim + 2]
            ext = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
            if dim == 3:
                ext = ext * (boxes[:, 5] - boxes[:, 4])
            keep = ext > 0                                            # zero-area detections are dropped (:757-766)
            boxes, class_ids, scores = boxes[keep], class_ids[keep], scores[keep]
            if return_masks and masks_np is not None and boxes.shape[0] > 0:
                from scipy.ndimage import zoom
                m = masks_np[sel][keep]
                full = []
                for i in range(boxes.shape[0]):
                    mi = m[i][..., class_ids[i]]
                    bb = boxes[i]
                    tgt = (bb[2] - bb[0], bb[3] - bb[1]) + ((bb[5] - bb[4],) if dim == 3 else ())
                    fm = np.zeros(img_shape[2:])
                    zm = zoom(mi, [t / s for t, s in zip(tgt, mi.shape)], order=1)
                    sl = (slice(bb[0], bb[2]), slice(bb[1], bb[3])) + ((slice(bb[4], bb[5]),) if dim == 3 else ())
                    fm[sl] = zm
                    full.append(fm)
                final_masks = np.max(np.array(full), 0) if full else final_masks
            for i2, score in enumerate(scores):
                box_results_list[ix].append({"box_coords": boxes[i2], "box_score": score, "box_type": "det",
                                             "box_pred_class_id": class_ids[i2]})
        seg_preds.append(final_masks)
    seg = np.zeros((img_shape[0], 1) + tuple(img_shape[2:]), dtype=np.uint8)
    for ix, fm in enumerate(seg_preds):
        if fm is not None:
            seg[ix, 0] = np.round(fm).astype("uint8")
    return {"boxes": box_results_list, "seg_preds": seg}


############################################################
#  Mask R-CNN
############################################################
class net(nn.Module):
    def __init__(self, cf, logger=None, device=None):
        super(net, self).__init__()
        self.cf = cf
        self.logger = logger
        self.device_ = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.build()

    def build(self):
        h, w = self.cf.patch_size[:2]
        if h / 2 ** 5 != int(h / 2 ** 5) or w / 2 ** 5 != int(w / 2 ** 5):
            raise Exception("Image size must be dividable by 2 at least 5 times to avoid fractions when downscaling and upscaling.")
        if len(self.cf.patch_size) == 3 and self.cf.patch_size[2] / 2 ** 3 != int(self.cf.patch_size[2] / 2 ** 3):
            raise Exception("Image z dimension must be dividable by 2 at least 3 times to avoid fractions when downscaling and upscaling.")
        conv = mutils.NDConvGenerator(self.cf.dim)
        # anchors live on the device: float64 table for the matching kernel, fp32 copy for the proposal layer
        self.anchors_f64, self.anchors = mutils.generate_pyramid_anchors(self.logger, self.cf, device=self.device_, return_f32=True)
        self.fpn = backbone_module.FPN(self.cf, conv)
        self.rpn = RPN(self.cf, conv)
        self.classifier = Classifier(self.cf, conv)
        self.mask = Mask(self.cf, conv)
        self.to(self.device_)
        # optional NDHWC ("channels last") weights/activations for the MIOpen/CK conv path: removes the
        # NCDHW<->NDHWC transposes CK's grouped-conv kernels otherwise insert
        self.memory_format = None
        if getattr(self.cf, "channels_last", False):
            self.memory_format = torch.channels_last_3d if self.cf.dim == 3 else torch.channels_last
            self.to(memory_format=self.memory_format)

    @property
    def np_anchors(self):
        return self.anchors_f64.cpu().numpy()

    # ------------------------------------------------------------------ which parameters have a gradient only under a condition of the step
    def grad_condition_spec(self):
        """[(condition, [parameters])] for training.FlatAdam.attach_conditions.  The reference's loss helpers return CONSTANTS when a step
        samples nothing for them -- compute_rpn_class_loss without sampled anchors (mrcnn.py:193-209), compute_rpn_bbox_loss without a
        positive anchor (:233-234), compute_mrcnn_class_loss without a sampled RoI (:247-248), compute_mrcnn_bbox_loss / _mask_loss without a
        positive RoI (:266-268, :287-288) -- so autograd hands the parameters below no gradient in such a step and torch.optim.Adam
        (exec.py:74) does not touch them.  The fixed-size masked step here gives them an exact zero gradient instead; the counts that decide
        are written to `grad_cond` by train_forward_device, in this order."""
        r, c, m = self.rpn, self.classifier, self.mask
        spec = [("rpn_samples", list(r.conv_shared.parameters()) + list(r.conv_class.parameters())),
                ("rpn_positive_anchors", list(r.conv_bbox.parameters())),
                ("sampled_rois", list(c.conv1.parameters()) + list(c.conv2.parameters()) + list(c.linear_class.parameters())),
                ("positive_rois", list(c.linear_bbox.parameters()) + ([] if self.cf.frcnn_mode else list(m.parameters()))),
                ("any_sample", list(self.fpn.parameters()))]
        return spec

    def set_grad_cond_buffer(self, t):
        """float32 [5] device tensor (or None) that train_forward_device fills with the counts of grad_condition_spec()"""
        self._grad_cond = t

    # ------------------------------------------------------------------ forward passes
    def forward(self, img, is_training=True, with_masks=True, rpn_graph=True, keep_rpn_maps=None):
        """mrcnn.py:987-1050.  `is_training` selects the proposal count exactly as the reference does (:1014); note that the reference's
        own test_forward (:982) calls `self.forward(img)` with the DEFAULT True, so its inference runs on post_nms_rois_training proposals
        (75 in 3D) and post_nms_rois_inference is never reached -- test_forward below follows that (cf.test_forward_proposals).
        keep_rpn_maps (default: is_training) keeps the FPN outputs for rpn_at_anchors.  with_masks=False skips the mask head over the detections (the reference always runs it and
        drops the result when return_masks is off, mrcnn.py:1046-1048 / :984): box-only inference.  rpn_graph=False: the dense RPN
        outputs carry no autograd graph (train_forward_device differentiates the RPN losses through rpn_at_anchors instead)."""
        cf = self.cf
        B = img.shape[0]
        if self.memory_format is not None:
            img = img.contiguous(memory_format=self.memory_format)
        fpn_outs = self.fpn(img)
        rpn_feature_maps = [fpn_outs[i] for i in cf.pyramid_levels]
        # round 5: channels-last maps (what the convolution path produces) are pooled AS THEY ARE by mdt_pyramid_roi_align_forward_cl -- a
        # corner voxel is 36 contiguous floats serving every channel; no row-major copy of the pyramid per forward, no copy back in the
        # backward.  Otherwise the RoIAlign kernels read [B, C, spatial] row-major maps: convert ONCE per forward (every head call would
        # re-copy all levels)
        from ..cuda_functions import _roi_align_impl as _rai
        if _rai.channels_last_eligible(rpn_feature_maps, cf.dim):
            self.mrcnn_feature_maps = list(rpn_feature_maps)
        else:
            self.mrcnn_feature_maps = [m.contiguous() for m in rpn_feature_maps]
        if keep_rpn_maps is None:
            keep_rpn_maps = is_training
        self.rpn_feature_maps = rpn_feature_maps if keep_rpn_maps else None       # read by rpn_at_anchors, released after the RPN losses
        fused_rpn = rpn_levels_fused(self.rpn, rpn_feature_maps) if (RPN_HEADS_FUSED and not (rpn_graph and torch.is_grad_enabled())) else None
        if fused_rpn is not None:
            rpn_pred_logits, rpn_pred_probs, rpn_pred_deltas = fused_rpn
        else:
            with torch.set_grad_enabled(rpn_graph and torch.is_grad_enabled()):
                layer_outputs = [self.rpn(p) for p in rpn_feature_maps]
                rpn_pred_logits, rpn_pred_probs, rpn_pred_deltas = [torch.cat(list(o), dim=1) for o in zip(*layer_outputs)]
        proposal_count = cf.post_nms_rois_training if is_training else cf.post_nms_rois_inference
        batch_rpn_rois, batch_proposal_boxes = proposal_layer(rpn_pred_probs, rpn_pred_deltas, proposal_count, self.anchors, cf)
        batch_ixs = torch.arange(B, device=img.device, dtype=torch.float32).repeat_interleave(batch_rpn_rois.shape[1])
        rpn_rois = batch_rpn_rois.reshape(-1, batch_rpn_rois.shape[2])
        self.rpn_rois_batch_info = torch.cat((rpn_rois, batch_ixs.unsqueeze(1)), dim=1)
        class_logits_list, bboxes_list = [], []
        with torch.no_grad():
            for chunk in self.rpn_rois_batch_info.split(cf.roi_chunk_size):
                cl, bb = self.classifier(self.mrcnn_feature_maps, chunk)
                class_logits_list.append(cl)
                bboxes_list.append(bb)
        batch_mrcnn_class_logits = torch.cat(class_logits_list, 0)
        batch_mrcnn_bbox = torch.cat(bboxes_list, 0)
        self.batch_mrcnn_class_scores = F.softmax(batch_mrcnn_class_logits, dim=1)
        detections, det_valid = refine_detections(rpn_rois, self.batch_mrcnn_class_scores, batch_mrcnn_bbox, batch_ixs, cf, B)
        dim = cf.dim
        scale = mutils.const_tensor(list(cf.scale) + [1], torch.float32, img.device)
        detection_boxes = detections[:, :dim * 2 + 1] / scale
        detection_boxes = torch.cat([detection_boxes[:, :dim * 2],
                                     torch.where(det_valid, detection_boxes[:, dim * 2], torch.full_like(detection_boxes[:, dim * 2], -1.0)).unsqueeze(1)], 1)
        detection_masks = None
        if with_masks:
            with torch.no_grad():
                detection_masks = self.mask(self.mrcnn_feature_maps, detection_boxes)
        return [rpn_pred_logits, rpn_pred_deltas, batch_proposal_boxes, detections, det_valid, detection_masks]

    def loss_samples_forward(self, batch_gt_class_ids, batch_gt_boxes, batch_gt_masks, B, gt_dev=None):
        """mrcnn.py:1052-1082."""
        sample_ix, valid, is_pos, tcls, tdeltas, tmasks = detection_target_layer(
            self.rpn_rois_batch_info, self.batch_mrcnn_class_scores, batch_gt_class_ids, batch_gt_boxes, batch_gt_masks, self.cf, B,
            gt_dev=gt_dev)
        sample_proposals = self.rpn_rois_batch_info[sample_ix]
        dim = self.cf.dim
        sample_proposals = torch.cat([sample_proposals[:, :2 * dim],
                                      torch.where(valid, sample_proposals[:, 2 * dim], torch.full_like(sample_proposals[:, 2 * dim], -1.0)).unsqueeze(1)], 1)
        sample_logits, sample_boxes = self.classifier(self.mrcnn_feature_maps, sample_proposals)
        sample_mask = self.mask(self.mrcnn_feature_maps, sample_proposals)
        return [sample_logits, sample_boxes, sample_mask, tcls, tdeltas, tmasks, sample_proposals, valid, is_pos]

    # ------------------------------------------------------------------ the training step: host half / device half / read-out
    def prepare_batch(self, batch, gmax=None):
        """Host half of train_forward (mrcnn.py:864-869: the uploads).  Returns {'img': [B, C, *patch] fp32 device tensor, 'gt': GtOnDevice,
        'masks': stacked GT masks [sum_G, 1, *patch] uint8 device tensor | StagedUpload (resolved after the backbone was launched) | None}.
        All GT boxes / class ids go up in one pinned async copy BEFORE the backbone is launched: nothing in the step waits for the
        stream afterwards, so the host keeps running ahead of the GPU through the glue."""
        cf, dev = self.cf, self.device_
        ev = batch.get("ready_event")            # training.DevicePrefetcher uploaded on its side stream
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for k in ("data", "roi_masks_device"):
                if torch.is_tensor(batch.get(k)):
                    batch[k].record_stream(cur)
        img = mutils.upload(batch["data"], dev).float()
        if "roi_masks_device" in batch:          # already resident in HBM (utils.synthetic_data.to_device)
            gt_masks = batch["roi_masks_device"]
        else:
            # staged into pinned memory on a background thread while this thread launches the backbone
            gt_masks = mutils.StagedUpload([m for m in batch["roi_masks"] if len(m) > 0], dev)
        gt_dev = GtOnDevice(batch["bb_target"], batch["roi_labels"], cf.dim, dev, gmax=gmax)
        return {"img": img, "gt": gt_dev, "masks": gt_masks}

    def train_forward_device(self, img, gt_dev, gt_masks, with_masks=False):
        """Device half of train_forward (mrcnn.py:870-946): forward, target layer, heads, anchor matching, the five loss terms.  Touches no
        host data and no host-side count (GT counts are read on the device), launches only: the whole function is capturable in a
        hipGraph (training.GraphedTrainStep).  Returns a dict of DEVICE tensors: 'loss', 'terms', 'sample_counts' and what the
        monitoring read-out of exec.py needs ('mon')."""
        cf = self.cf
        B = img.shape[0]
        sparse_rpn = SPARSE_RPN_LOSS and rpn_sparse_supported(self.rpn)
        rpn_class_logits, rpn_pred_deltas, proposal_boxes, detections, det_valid, detection_masks = self.forward(
            img, with_masks=with_masks, rpn_graph=not sparse_rpn)
        if isinstance(gt_masks, mutils.StagedUpload):
            gt_masks = gt_masks.get()
        acc = None
        if SHARED_PYRAMID_GRAD and torch.is_grad_enabled() and sparse_rpn and self.rpn_feature_maps is not None and \
                all(a is b for a, b in zip(self.mrcnn_feature_maps, self.rpn_feature_maps)) and all(m.requires_grad for m in self.mrcnn_feature_maps):
            acc = _rai_mod.PyramidGradAccumulator(self.mrcnn_feature_maps)
        self._pyramid_grad_acc = acc
        _rai_mod.PyramidGradAccumulator.CURRENT = acc
        try:
            return self._train_forward_device_tail(img, gt_dev, gt_masks, B, sparse_rpn, rpn_class_logits, rpn_pred_deltas, proposal_boxes, detections, det_valid,
                                                   detection_masks)
        finally:
            _rai_mod.PyramidGradAccumulator.CURRENT = None

    def _train_forward_device_tail(self, img, gt_dev, gt_masks, B, sparse_rpn, rpn_class_logits, rpn_pred_deltas, proposal_boxes, detections, det_valid,
                                   detection_masks):
        cf = self.cf
        (mrcnn_class_logits, mrcnn_pred_deltas, mrcnn_pred_mask, target_class_ids, mrcnn_target_deltas, target_mask,
         sample_proposals, s_valid, s_pos) = self.loss_samples_forward(None, None, gt_masks, B, gt_dev=gt_dev)

        # anchor matching of the whole batch on the device, ONE launch pair (the reference: numpy on one host core per element,
        # mrcnn.py:894); the per-element GT counts are read by the kernel
        neg_thr = 0.1 if cf.dim == 2 else 0.01
        rpn_match, rpn_argmax = mutils.anchor_match_labels_batched(self.anchors_f64, gt_dev.px, gt_dev.n_gt, None, neg_thr, float(cf.anchor_matching_iou))
        sparse_eval = None
        if sparse_rpn:
            maps, n_apv = self.rpn_feature_maps, len(cf.rpn_anchor_ratios)

            def sparse_eval(idx):
                return rpn_at_anchors(self.rpn, maps, idx, n_apv)
        batch_rpn_class_loss, batch_rpn_bbox_loss, rpn_samples = compute_rpn_losses(
            rpn_match, rpn_argmax, rpn_class_logits, rpn_pred_deltas, self.anchors_f64, None, cf, gt_dev=gt_dev, sparse_eval=sparse_eval)
        self.rpn_feature_maps = None

        mrcnn_class_loss = compute_mrcnn_class_loss(target_class_ids, mrcnn_class_logits, s_valid)
        mrcnn_bbox_loss = compute_mrcnn_bbox_loss(mrcnn_target_deltas, mrcnn_pred_deltas, target_class_ids, s_pos)
        if not cf.frcnn_mode:
            mrcnn_mask_loss = compute_mrcnn_mask_loss(target_mask, mrcnn_pred_mask, target_class_ids, s_pos)
        else:
            mrcnn_mask_loss = torch.zeros((), device=img.device)
        loss = batch_rpn_class_loss + batch_rpn_bbox_loss + mrcnn_class_loss + mrcnn_bbox_loss + mrcnn_mask_loss
        terms = {"rpn_class": batch_rpn_class_loss.detach(), "rpn_bbox": batch_rpn_bbox_loss.detach(),
                 "mrcnn_class": mrcnn_class_loss.detach(), "mrcnn_bbox": mrcnn_bbox_loss.detach(), "mrcnn_mask": mrcnn_mask_loss.detach()}
        gc = getattr(self, "_grad_cond", None)
        if gc is not None:          # who had something to learn from in this step (FlatAdam skips the others like torch.optim.Adam does)
            n_rpn_pos, n_rpn_neg = rpn_samples[1].sum(), rpn_samples[3].sum()
            n_rois, n_pos = s_valid.sum(), s_pos.sum()
            gc.copy_(torch.stack([n_rpn_pos + n_rpn_neg, n_rpn_pos, n_rois, n_pos, n_rpn_pos + n_rpn_neg + n_rois]))
        mon = {"rpn_samples": rpn_samples, "proposal_boxes": proposal_boxes, "sample_proposals": sample_proposals.detach(),
               "target_class_ids": target_class_ids, "s_valid": s_valid, "detections": detections, "det_valid": det_valid,
               "detection_masks": detection_masks}
        return {"loss": loss, "terms": terms, "sample_counts": (s_valid.sum(), s_pos.sum()), "mon": mon}

    def monitor_pack(self, out):
        """Device side of the monitoring read-out (mrcnn.py:898-961 reads a dozen tensors back one by one): everything exec.py's
        consumers need -- the sampled anchors, the best n_plot_rpn_props proposals per element, the sampled RoIs with their target
        classes, the detections and the six loss values -- packed into ONE int32 buffer (fp32 values by bit pattern), so a step has one
        device->host copy.  Launches only (capturable).  Returns (buffer, layout)."""
        cf, mon, terms = self.cf, out["mon"], out["terms"]
        pidx, pvalid, nidx, nvalid = mon["rpn_samples"]
        props = mon["proposal_boxes"]
        n_plot = min(int(cf.n_plot_rpn_props), int(props.shape[1]))
        _, order = torch.sort(props[:, :, -1], dim=1, descending=True, stable=True)          # :919-921: best proposals first
        top = torch.gather(props[:, :, :-1], 1, order[:, :n_plot].unsqueeze(-1).expand(-1, -1, props.shape[2] - 1))
        vals = torch.stack([out["loss"].detach(), terms["rpn_class"], terms["rpn_bbox"], terms["mrcnn_class"], terms["mrcnn_bbox"], terms["mrcnn_mask"]])
        items = [("pidx", pidx), ("pvalid", pvalid), ("nidx", nidx), ("nvalid", nvalid), ("props", top), ("sample_proposals", mon["sample_proposals"]),
                 ("target_class_ids", mon["target_class_ids"]), ("s_valid", mon["s_valid"]), ("detections", mon["detections"]),
                 ("det_valid", mon["det_valid"]), ("vals", vals)]
        return mutils.pack_for_readout(items)

    def monitor_results(self, packed, batch, img_shape, is_validation=False, detection_masks=None):
        """Host side of the read-out: the reference's results_dict entries ('boxes', 'seg_preds', 'monitor_values', 'logger_string',
        mrcnn.py:898-961) from the packed buffer of monitor_pack()."""
        cf = self.cf
        r = mutils.unpack_readout(packed)
        B = img_shape[0]
        box_results_list = [[] for _ in range(B)]
        for b in range(B):
            for ix in range(len(batch["bb_target"][b])):
                box_results_list[b].append({"box_coords": batch["bb_target"][b][ix], "box_label": batch["roi_labels"][b][ix], "box_type": "gt"})
        if getattr(self, "_anchors_host", None) is None:
            self._anchors_host = self.anchors.cpu().numpy()          # constant table: read back once, not every step
        anchors_np = self._anchors_host
        pidx, pvalid, nidx, nvalid = r["pidx"], r["pvalid"].astype(bool), r["nidx"], r["nvalid"].astype(bool)
        for b in range(B):
            for a in anchors_np[pidx[b][pvalid[b]]]:
                box_results_list[b].append({"box_coords": a, "box_type": "pos_anchor"})
            for a in anchors_np[nidx[b][nvalid[b]]]:
                box_results_list[b].append({"box_coords": a, "box_type": "neg_anchor"})
            for row in r["props"][b]:
                box_results_list[b].append({"box_coords": row, "box_type": "prop"})
        sp, tc = r["sample_proposals"], r["target_class_ids"]
        for ix, row in enumerate(sp):
            if row[-1] >= 0:
                box_results_list[int(row[-1])].append({"box_coords": row[:-1] * cf.scale, "box_type": "pos_class" if tc[ix] > 0 else "neg_class"})
        return_masks = cf.return_masks_in_val if is_validation else False
        res = get_results(cf, img_shape, r["detections"], r["det_valid"].astype(bool), detection_masks, box_results_list, return_masks=return_masks)
        tcv = tc[r["s_valid"].astype(bool)]
        dcount = [int((tcv == c).sum()) for c in range(1, cf.head_classes)]
        vals = r["vals"]
        res["monitor_values"] = {"loss": float(vals[0]), "class_loss": float(vals[3])}
        res["logger_string"] = (
            "loss: {0:.2f}, rpn_class: {1:.2f}, rpn_bbox: {2:.2f}, mrcnn_class: {3:.2f}, mrcnn_bbox: {4:.2f}, "
            "mrcnn_mask: {5:.2f}, dcount {6}".format(vals[0], vals[1], vals[2], vals[3], vals[4], vals[5], dcount))
        return res

    def train_forward(self, batch, is_validation=False, monitor=True):
        """mrcnn.py:853-967.  batch: the reference's batch dict (numpy): 'data', 'roi_labels', 'bb_target', 'roi_masks'.
        monitor=False skips the read-out and the python box lists (the loss terms are unchanged).
        monitor="deferred": the read-out entries ('boxes', 'monitor_values', 'logger_string', ...) returned are those of the PREVIOUS
        call (`results["monitor_of_previous_step"] = True`; absent on the first call) -- the packed buffer travels with an asynchronous
        copy and nothing in the step waits for the GPU (utils.model_utils.DeferredReadout); `flush_deferred_monitor()` hands out the last one.
        The mask head over the DETECTIONS (mrcnn.py:1046-1048) only feeds the validation read-out (return_masks, :949); in a training
        step its result is dropped by the reference and it is not run here, unless cf.run_detection_mask_head_in_training asks for the
        reference's exact work (bench.py `exec_equivalent`)."""
        cf = self.cf
        d = self.prepare_batch(batch)
        with_masks = bool(is_validation and cf.return_masks_in_val) or bool(getattr(cf, "run_detection_mask_head_in_training", False))
        out = self.train_forward_device(d["img"], d["gt"], d["masks"], with_masks=with_masks)
        # the five terms of mrcnn.py:946 as device scalars (no read-out here): what the assembled-step parity test compares
        results_dict = {"torch_loss": out["loss"], "loss_terms": out["terms"], "sample_counts": out["sample_counts"]}
        if monitor == "deferred":
            if getattr(self, "_deferred", None) is None:
                self._deferred = mutils.DeferredReadout()
            prev = self._deferred.push(self.monitor_pack(out), (batch, tuple(d["img"].shape), is_validation))
            if prev is not None:
                results_dict.update(self._resolve_deferred(prev))
        elif monitor:
            results_dict.update(self.monitor_results(self.monitor_pack(out), batch, tuple(d["img"].shape), is_validation,
                                                     detection_masks=out["mon"]["detection_masks"]))
        return results_dict

    def _resolve_deferred(self, entry):
        packed, (batch, img_shape, is_validation) = mutils.DeferredReadout.resolve(entry)
        res = self.monitor_results(packed, batch, img_shape, is_validation, detection_masks=None)
        res["monitor_of_previous_step"] = True
        return res

    def flush_deferred_monitor(self):
        """read-out entries of the LAST train_forward(monitor="deferred") call, or None"""
        d = getattr(self, "_deferred", None)
        entry = d.flush() if d is not None else None
        return self._resolve_deferred(entry) if entry is not None else None

    def _test_is_training(self):
        """the `is_training` argument test_forward hands to forward(): the reference passes none (mrcnn.py:982 -> default True ->
        post_nms_rois_training proposals, :1014).  cf.test_forward_proposals = "inference" selects post_nms_rois_inference instead
        (what the config comment intends; NOT what the reference computes)."""
        return getattr(self.cf, "test_forward_proposals", "reference") != "inference"

    def test_forward(self, batch, return_masks=True):
        """mrcnn.py:969-985."""
        img = batch["data"]
        img = torch.from_numpy(np.ascontiguousarray(img)).to(self.device_).float() if not torch.is_tensor(img) else img.to(self.device_).float()
        with torch.no_grad():
            _, _, _, detections, det_valid, detection_masks = self.forward(img, is_training=self._test_is_training(), with_masks=return_masks, keep_rpn_maps=False)
        return get_results(self.cf, img.shape, detections, det_valid, detection_masks, return_masks=return_masks)

    def test_forward_detections(self, img):
        """test_forward without leaving the device (patch-tiled inference, predictor.collect_raw_boxes): img [B, C, *patch]
        device tensor -> (rows [B * M, 2 * dim + 3] float32 = box (y1, x1, y2, x2, (z1, z2)) truncated to integers like
        get_results (:467), batch_ix, class id, score; keep [B * M] bool = real detection with positive extent, :470-474).
        Rows are element-major in detection order -- the order get_results emits box dicts in.  No mask head, no read-out."""
        with torch.no_grad():
            img = img.to(self.device_).float()
            _, _, _, det, det_valid, _ = self.forward(img, is_training=self._test_is_training(), with_masks=False, keep_rpn_maps=False)
            dim = self.cf.dim
            boxes = det[:, :2 * dim].to(torch.int32).float()
            ext = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
            if dim == 3:
                ext = ext * (boxes[:, 5] - boxes[:, 4])
            return torch.cat([boxes, det[:, 2 * dim:]], 1), det_valid & (ext > 0)
