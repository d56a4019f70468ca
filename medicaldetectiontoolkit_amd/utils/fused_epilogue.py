"""Fused convolution epilogues (csrc/epilogue.hip): y = act(conv(x) + bias (+ residual)) with the bias add, the
residual / top-down add and the ReLU in ONE pass over the activation, and a backward that applies the ReLU mask and
reduces the bias gradient in one pass (deterministic).  The convolution itself stays on MIOpen (torch, bias=None).

The modules keep the reference's Sequential layout so state_dict keys are unchanged
(utils/model_utils.py:732-781: `conv` -> Sequential(conv[, norm][, relu]) or a bare conv when relu is None).

Also here, because they hang off the same modules (each with a module switch for A/B runs and a test against its torch op):
  * which convolution problem is posed: input gradients of unit-stride layers as forward convolutions (`_ConvStride1`), the stem in
    space-to-depth form (`_ConvStem221`), channels-last max pooling and x2 up-sampling (csrc/pool.hip, csrc/upsample.hip);
  * the layers MIOpen is 2-9x off its bound on, as fp32-MFMA kernels of this repo: 1x1x1 weight gradients
    (`conv1x1_weight_grad`, csrc/conv1x1_wgrad.hip), few-channel 3x3x3 layers forward / input gradient / weight gradient
    (`conv3x3x3_small*`, csrc/conv3x3x3_small.hip), the one-channel 7x7x7 stem forward with bias + ReLU epilogue and its weight
    gradient (`stem_forward`, `_ConvStemBiasReLU`, `stem_weight_grad`; csrc/conv_stem_fwd.hip, csrc/conv_stem_wgrad.hip)."""
import math
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib

ENABLED = True      # module switch (A/B measurements: bench.py --fused-epilogue 0)


def _layout(x):
    """(inner, memory_format) of a dense activation, or None when it is neither contiguous nor channels-last"""
    if x.is_contiguous():
        inner = 1
        for s in x.shape[2:]:
            inner *= int(s)
        return inner, torch.contiguous_format
    mf = torch.channels_last_3d if x.dim() == 5 else torch.channels_last if x.dim() == 4 else None
    if mf is not None and x.is_contiguous(memory_format=mf):
        return 1, mf
    return None


_WS = {}       # per-device workspace of the backward's partial sums, grown on demand (stream-ordered reuse)


def _workspace(nbytes, device):
    if _lib.CAPTURING:          # inside a hipGraph capture: a tensor of the graph's own pool (see _lib.CAPTURING)
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    ws = _WS.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[device] = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=device)
    return ws


def _on_current_device(t):
    return t.device.index is None or t.device.index == torch.cuda.current_device()


class _BiasAct(Function):
    """the per-call host work is kept minimal: these epilogues run ~150 times per training step"""

    @staticmethod
    def forward(ctx, x, bias, residual, relu, inner, mf):
        if residual is not None and residual.stride() != x.stride():
            residual = residual.contiguous(memory_format=mf)
        if not _on_current_device(x):
            raise RuntimeError("fused epilogue: tensor is not on the current device (one process per GPU)")
        rc = _lib.lib().mdt_bias_act_forward(x.data_ptr(), x.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
                                             x.numel(), x.shape[1], inner, 1 if relu else 0, _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_bias_act_forward")
        ctx.mark_dirty(x)
        ctx.relu, ctx.inner, ctx.mf, ctx.has_res = relu, inner, mf, residual is not None
        if relu:
            ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, gy):
        y = ctx.saved_tensors[0] if ctx.relu else None
        if not gy.is_contiguous(memory_format=ctx.mf):
            if not ctx.relu and ctx.inner == 1:
                r = bias_grad_to_channels_last(gy, ctx.mf)       # row-major gradient of a channels-last layer: converted and reduced in ONE pass
                if r is not None:
                    return r[0], r[1], (r[0] if ctx.has_res else None), None, None, None
            gy = gy.contiguous(memory_format=ctx.mf)
        L = _lib.lib()
        n, C = gy.numel(), gy.shape[1]
        # no activation: the input gradient IS gy -- returned as it is, the kernel only reduces (round 6; it used to store a copy)
        gx = torch.empty_like(gy) if (ctx.relu or not BIAS_BWD_NO_COPY) else None
        gbias = torch.empty(C, dtype=torch.float32, device=gy.device)
        wsb = (4096 * C * 4 + 256) if ctx.inner == 1 else L.mdt_bias_act_backward_workspace_bytes(n, C, ctx.inner)
        ws = _workspace(wsb, gy.device)
        if ctx.inner == 1 and BIAS_GRAD_IN_LAUNCH:          # one launch: the last block folds the partial rows (see _bias_act_bwd)
            tk = _TICKET.get(gy.device)
            if tk is None:
                tk = _TICKET[gy.device] = torch.zeros(1, dtype=torch.int32, device=gy.device)
            rc = L.mdt_bias_act_backward_ticket(gx.data_ptr() if gx is not None else None, gy.data_ptr(), y.data_ptr() if y is not None else None,
                                                gbias.data_ptr(), n, C, ctx.inner, 1 if ctx.relu else 0, ws.data_ptr(), ws.numel(), tk.data_ptr(), _lib.raw_stream())
        else:
            rc = L.mdt_bias_act_backward(gx.data_ptr() if gx is not None else None, gy.data_ptr(), y.data_ptr() if y is not None else None, gbias.data_ptr(),
                                         n, C, ctx.inner, 1 if ctx.relu else 0, ws.data_ptr(), ws.numel(), _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_bias_act_backward")
        if gx is None:
            gx = gy
        return gx, gbias, (gx if ctx.has_res else None), None, None, None


BIAS_BWD_NO_COPY = True        # module switch (A/B: bench.py --bias-bwd-no-copy 0): bias-only layers' backward returns gy itself as the input gradient
BIAS_GRAD_TRANSPOSE = True     # module switch (A/B: bench.py --bias-grad-transpose 0): row-major output gradients of channels-last bias-only layers in one pass
LATERAL_UPSAMPLE_FUSED = True  # module switch (A/B: bench.py --lateral-upsample-fused 0): the FPN's top-down add reads the coarser map directly


def bias_grad_to_channels_last(gy, mf):
    """(gx, gbias) for a bias-only layer on channels-last storage whose output gradient gy is a dense ROW-MAJOR tensor: gx = gy in channels-last
    storage, gbias = per-channel sum, one pass (csrc/epilogue.hip bias_grad_to_cl_kernel).  None when the case is not the kernel's."""
    if not (BIAS_GRAD_TRANSPOSE and gy.is_cuda and gy.dtype == torch.float32 and gy.is_contiguous() and gy.dim() >= 4 and _on_current_device(gy)):
        return None
    L = _lib.lib()
    B, C = int(gy.shape[0]), int(gy.shape[1])
    if not L.mdt_bias_grad_to_channels_last_supported(C):
        return None
    inner = int(gy.shape[2:].numel())
    gx = torch.empty_like(gy, memory_format=mf)
    gbias = torch.empty(C, dtype=torch.float32, device=gy.device)
    ws = _workspace(L.mdt_bias_grad_to_channels_last_workspace_bytes(B, C, inner), gy.device)
    rc = L.mdt_bias_grad_to_channels_last(gx.data_ptr(), gy.data_ptr(), gbias.data_ptr(), B, C, inner, ws.data_ptr(), ws.numel(), _lib.raw_stream())
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        return None
    if rc != 0:
        _lib.check(rc, "mdt_bias_grad_to_channels_last")
    return gx, gbias


class _BiasAddUpsampled(Function):
    """y = x + bias + nearest-up-sampled coarse map, in place on x (channels-last fp32): the FPN's top-down step (models/backbone.py:147-153) with the
    up-sampling folded into the lateral's epilogue -- the up-sampled map (151 MB on P2 at the benchmark patch) is never written.  Backward: the
    input gradient is gy itself, the bias gradient its per-channel sum, the coarse map's gradient the sum of gy over each voxel's replicas (the
    strided sum _UpsampleNearestCL.backward runs)."""

    @staticmethod
    def forward(ctx, x, bias, coarse, scale, mf):
        if not _on_current_device(x):
            raise RuntimeError("fused epilogue: tensor is not on the current device (one process per GPU)")
        nd = x.dim() - 2
        sp = [int(v) for v in x.shape[2:]]
        Y, X, Z = (sp[0], sp[1], sp[2]) if nd == 3 else (sp[0], sp[1], 1)
        sy, sx, sz = (scale[0], scale[1], scale[2]) if nd == 3 else (scale[0], scale[1], 1)
        rc = _lib.lib().mdt_bias_act_forward_upsampled(x.data_ptr(), x.data_ptr(), bias.data_ptr(), coarse.data_ptr(), int(x.shape[0]), Y, X, Z,
                                                       int(x.shape[1]), sy, sx, sz, _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_bias_act_forward_upsampled")
        ctx.mark_dirty(x)
        ctx.scale, ctx.mf, ctx.coarse_shape = scale, mf, tuple(coarse.shape)
        return x

    @staticmethod
    def backward(ctx, gy):
        gx = gbias = None
        if not gy.is_contiguous(memory_format=ctx.mf):
            r = bias_grad_to_channels_last(gy, ctx.mf)
            if r is not None:
                gx, gbias = r
            else:
                gy = gy.contiguous(memory_format=ctx.mf)
        if gx is None:
            L = _lib.lib()
            n, C = gy.numel(), gy.shape[1]
            gbias = torch.empty(C, dtype=torch.float32, device=gy.device)
            ws = _workspace(4096 * C * 4 + 256, gy.device)
            rc = L.mdt_bias_act_backward(None, gy.data_ptr(), None, gbias.data_ptr(), n, C, 1, 0, ws.data_ptr(), ws.numel(), _lib.raw_stream())
            if rc != 0:
                _lib.check(rc, "mdt_bias_act_backward")
            gx = gy
        nd = gx.dim() - 2
        perm = (0,) + tuple(range(2, 2 + nd)) + (1,)
        B, C = ctx.coarse_shape[0], ctx.coarse_shape[1]
        shp, red = [B], []
        for d in range(nd):
            shp += [ctx.coarse_shape[2 + d], ctx.scale[d]]
            red.append(2 + 2 * d)
        gs = gx.permute(*perm).reshape(*shp, C).sum(tuple(red))        # [B, *coarse spatial, C]
        inv = (0, nd + 1) + tuple(range(1, nd + 1))
        return gx, gbias, gs.permute(*inv), None, None


def conv_bias_add_upsampled(conv, x, coarse, scale_factor=2):
    """conv(x) + bias + F.interpolate(coarse, scale_factor) (nearest) -- the lateral of the FPN's top-down path; the add and the up-sampling ride
    the convolution's epilogue where that applies (channels-last fp32 on the GPU, channels % 4 == 0), torch ops otherwise"""
    h = _conv(conv, x)
    nd = h.dim() - 2
    sc = tuple(scale_factor) if isinstance(scale_factor, (tuple, list)) else (scale_factor,) * nd
    mf = torch.channels_last_3d if nd == 3 else torch.channels_last if nd == 2 else None
    if ENABLED and LATERAL_UPSAMPLE_FUSED and mf is not None and h.is_cuda and h.dtype == torch.float32 and conv.bias is not None and conv.bias.dtype == torch.float32 \
            and coarse.dtype == torch.float32 and all(float(v) == int(v) and int(v) >= 1 for v in sc) and not torch.is_autocast_enabled() \
            and h.is_contiguous(memory_format=mf) and not h.is_contiguous() and coarse.is_contiguous(memory_format=mf) and not coarse.is_contiguous() \
            and coarse.shape[:2] == h.shape[:2] and all(int(c) * int(v) == int(o) for c, v, o in zip(coarse.shape[2:], sc, h.shape[2:])) \
            and (h.data_ptr() | coarse.data_ptr()) % 16 == 0 and _lib.lib().mdt_bias_act_forward_upsampled_supported(int(h.shape[1]), h.numel()):
        return _BiasAddUpsampled.apply(h, conv.bias, coarse, tuple(int(v) for v in sc), mf)
    return bias_act(h, conv.bias, upsample_nearest(coarse, scale_factor), False)


def bias_act(x, bias, residual=None, relu=False):
    """x: fresh fp32 conv output on the GPU (modified in place); falls back to torch ops otherwise"""
    if ENABLED and x.is_cuda and x.dtype == torch.float32 and bias is not None and bias.dtype == torch.float32 \
            and (residual is None or (residual.dtype == torch.float32 and residual.shape == x.shape)):
        lay = _layout(x)
        if lay is not None:
            return _BiasAct.apply(x, bias, residual, relu, lay[0], lay[1])
    shape = [1, -1] + [1] * (x.dim() - 2)
    y = x + bias.view(shape).to(x.dtype) if bias is not None else x
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


BWD_DATA_AS_FWD = True   # module switch (A/B: bench.py --conv-bwd-as-fwd 0)
STEM_SPACE_TO_DEPTH = True   # module switch (A/B: bench.py --stem-s2d 0)


FLIP_BATCHED = True      # module switch (A/B): all flipped filters of a step from ONE launch (mdt_filter_flip_transpose_batched)
_WEIGHT_EPOCH = [0]      # bumped by weights_changed(): optimizers that update parameters without torch's version counter (training.FlatAdam)
_FLIP = {}               # device -> _FlipCache


def weights_changed():
    """an optimizer has rewritten parameters through a raw kernel (training.FlatAdam._update): cached flipped filters are stale.  In-place
    torch ops (torch.optim.*, load_state_dict, copy_) are seen through the tensors' own version counters."""
    _WEIGHT_EPOCH[0] += 1


class _FlipCache(object):
    """the flipped / transposed filters of every unit-stride convolution whose input gradient runs as a forward convolution: the weights
    change once per step (the optimizer), the ~60 flips of a step's backward are therefore ONE launch over a device table of
    {source, destination, shape} records -- issued by the first request of a backward, served from the persistent buffers for the rest.
    A filter is registered at its first request (single launch that time); dead weights (their nets were deleted) are dropped and the
    table rebuilt.  Eager steps only: inside a hipGraph capture flip_transpose_filter does not come here (a captured launch would keep the
    address of a table that a later rebuild frees)."""

    def __init__(self, device):
        self.device = device
        self.entries = {}          # (data_ptr, shape, cl) -> [weakref(w), out, version_seen]
        self.table = None          # uint8 device tensor holding the records
        self.order = []            # keys in table order
        self.total = 0
        self.dirty = True
        self.epoch = -1            # _WEIGHT_EPOCH value the buffers were made for

    def _rebuild(self):
        import struct
        self.entries = {k: e for k, e in self.entries.items() if e[0]() is not None and e[0]().data_ptr() == k[0]}
        self.order = list(self.entries)
        recs, first = [], 0
        for k in self.order:
            w = self.entries[k][0]()
            out = self.entries[k][1]
            cout, cin, taps = int(k[1][0]), int(k[1][1]), int(math.prod(k[1][2:]))
            recs.append(struct.pack("<QQiiiiq", w.data_ptr(), out.data_ptr(), cout, cin, taps, int(k[2]), first))
            first += cout * cin * taps
        self.total = first
        blob = b"".join(recs)
        self.table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.device) if blob else None
        self.dirty = False

    def get(self, w, mf):
        import weakref
        cl = 0 if mf == torch.contiguous_format else 1
        key = (w.data_ptr(), tuple(w.shape), cl)
        e = self.entries.get(key)
        if e is not None and e[0]() is not w:
            e = None                                        # the address was re-used by another tensor
        if e is None:
            if _lib.CAPTURING:
                return None
            out = torch.empty((int(w.shape[1]), int(w.shape[0])) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device, memory_format=mf)
            _flip_single(w, out, cl)
            self.entries[key] = [weakref.ref(w), out, w._version]
            self.dirty = True
            return out
        if self.epoch == _WEIGHT_EPOCH[0] and e[2] == w._version and not self.dirty:
            return e[1]
        # stale: one launch refreshes EVERY registered filter
        if self.dirty or any(en[0]() is None for en in self.entries.values()):
            if _lib.CAPTURING:
                return None
            self._rebuild()
            e = self.entries.get(key)
            if e is None:
                return self.get(w, mf)
        if self.table is not None:
            rc = _lib.lib().mdt_filter_flip_transpose_batched(self.table.data_ptr(), len(self.order), self.total, _lib.raw_stream())
            if rc != 0:
                _lib.check(rc, "mdt_filter_flip_transpose_batched")
        for en in self.entries.values():
            t = en[0]()
            en[2] = t._version if t is not None else -1
        self.epoch = _WEIGHT_EPOCH[0]
        return e[1]


def _flip_single(w, out, cl):
    rc = _lib.lib().mdt_filter_flip_transpose(w.data_ptr(), out.data_ptr(), int(w.shape[0]), int(w.shape[1]), int(w.shape[2:].numel()), cl, _lib.raw_stream())
    if rc != 0:
        _lib.check(rc, "mdt_filter_flip_transpose")


def flip_transpose_filter(w, mf):
    """w [cout, cin, *k] -> [cin, cout, *k] with every spatial axis reversed, dense in memory format `mf`: served from the per-step batched
    launch (_FlipCache) for parameters, one launch of mdt_filter_flip_transpose for other fp32 GPU tensors dense in `mf`, torch ops otherwise.
    The returned tensor of the cached path is a persistent buffer: valid until the weights change (read it within the step)."""
    nd = w.dim() - 2
    if w.is_cuda and w.dtype == torch.float32 and w.is_contiguous(memory_format=mf) and _on_current_device(w):
        # (never inside a hipGraph capture: the graph would bake the address of the record table, which is rebuilt -- and its old tensor freed -- as soon as
        # another net registers filters; replays then read a stale table.  Captured steps keep one flip launch per layer, written into the graph's pool.)
        if FLIP_BATCHED and not _lib.CAPTURING and isinstance(w, torch.nn.Parameter):
            cache = _FLIP.get(w.device)
            if cache is None:
                cache = _FLIP[w.device] = _FlipCache(w.device)
            out = cache.get(w, mf)
            if out is not None:
                return out
        cout, cin = int(w.shape[0]), int(w.shape[1])
        out = torch.empty((cin, cout) + tuple(w.shape[2:]), dtype=torch.float32, device=w.device, memory_format=mf)
        rc = _lib.lib().mdt_filter_flip_transpose(w.data_ptr(), out.data_ptr(), cout, cin, int(w.shape[2:].numel()),
                                                  0 if mf == torch.contiguous_format else 1, _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_filter_flip_transpose")
        return out
    wt = w.transpose(0, 1)
    if w.shape[2:].numel() > 1:
        wt = wt.flip(*range(2, 2 + nd))
    return wt.contiguous(memory_format=mf)


WGRAD_1X1 = True   # module switch (A/B: bench.py --wgrad-1x1 0)


def conv1x1_wgrad_pays(n_voxels, cout, cin):
    """where the kernel beats MIOpen's backward-weights (tools/wgrad_probe.py, profiles/r03_wgrad_probe.jsonl, B = 8): the
    few-channel layers on the large maps -- 18 -> 72: 431 -> 107 us, 72 -> 18: 310 -> 103, 18 -> 18: 549 -> 46, 128 -> 18: 309 -> 140,
    72 -> 36: 314 -> 195 on 32x32x128; 36 <-> 144 and 72 -> 36 on 16x16x64: 1.4-1.7x.  The small maps (launch-bound: 72 <-> 288 on
    8x8x32 74 vs 33 us) and many-tile layers stay on MIOpen."""
    tiles = ((cout + 31) // 32) * ((cin + 31) // 32)
    if cout > 4096 or cin > 4096:
        return False
    return (tiles <= 6 and n_voxels >= 65536) or (tiles <= 10 and 65536 <= n_voxels <= 262144)


def conv1x1_weight_grad(gy, x, w, force=False):
    """Weight gradient of a 1x1(x1) unit-stride convolution on channels-last activations with the fp32-MFMA kernel of
    csrc/conv1x1_wgrad.hip (MIOpen's backward-weights solvers take 310-434 us for the 18/72-channel layers on the 8 x 32x32x128
    maps against a 47 us HBM floor).  None when the layer is not of that form (the caller then asks MIOpen)."""
    if w.shape[2:].numel() != 1 or gy.dtype != torch.float32 or x.dtype != torch.float32 or not gy.is_cuda or not _on_current_device(gy):
        return None
    mf = torch.channels_last_3d if gy.dim() == 5 else torch.channels_last if gy.dim() == 4 else None
    if mf is None or not x.is_contiguous(memory_format=mf):
        return None
    if not gy.is_contiguous(memory_format=mf):
        gy = gy.contiguous(memory_format=mf)
    cout, cin = int(w.shape[0]), int(w.shape[1])
    V = gy.numel() // cout
    if not force and not conv1x1_wgrad_pays(V, cout, cin):
        return None
    L = _lib.lib()
    wsb = L.mdt_conv1x1_wgrad_workspace_bytes(V, cout, cin)
    ws = _workspace(wsb, gy.device)
    gw = torch.empty((cout, cin), dtype=torch.float32, device=gy.device)
    rc = L.mdt_conv1x1_wgrad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), V, cout, cin, ws.data_ptr(), ws.numel(),
                             _lib.raw_stream())
    if rc != 0:
        _lib.check(rc, "mdt_conv1x1_wgrad")
    return gw.view(w.shape)


CONV3_SMALL = True   # module switch (A/B: bench.py --conv3-small 0)


def _conv3_small_pays(cin, cout):
    """MIOpen has efficient kernels when the channel counts are multiples of 8 (16 -> 16 on 8 x 32x32x128: 254 us forward against
    504 us here) and poor ones otherwise (18 -> 18: 857 us against 393; 6 -> 6: 330 against 148; tools/conv3_probe.py)"""
    return cin % 8 != 0 or cout % 8 != 0


CONV_WIN = True     # module switch (A/B: bench.py --conv-win 0): the few-channel 3x3x3 layers on the unit-stride window kernel (mdt_conv_win_forward) where it is faster


def conv3x3x3_small(x, w, bias=None, relu=False):
    """3x3x3 / stride 1 / pad 1 convolution of a channels_last_3d fp32 activation with a few-channel filter (C_in even <= 32,
    C_out <= 32, Z % 32 == 0) on the fp32-MFMA kernel of csrc/conv3x3x3_small.hip (MIOpen: 862 us for 18 -> 18 on 8 x 32x32x128),
    optionally with the bias add and the ReLU in the kernel's epilogue.
    w: [C_out, C_in, 3, 3, 3] in any memory format.  None when the shape is not of that form (the caller then asks MIOpen)."""
    if not (CONV3_SMALL and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 5
            and tuple(w.shape[2:]) == (3, 3, 3) and _on_current_device(x)):
        return None
    B, cin, Y, X, Z = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    L = _lib.lib()
    if int(w.shape[1]) != cin or not L.mdt_conv3x3x3_small_supported(Y, X, Z, cin, cout) or B * Y * X * Z < 65536 or not _conv3_small_pays(cin, cout):
        return None
    if not x.is_contiguous(memory_format=torch.channels_last_3d):
        return None
    wt = w.permute(2, 3, 4, 1, 0).contiguous()                  # [27][C_in][C_out]
    y = torch.empty((B, cout, Y, X, Z), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d)
    if CONV_WIN and L.mdt_conv_win_forward_supported(Y, X, Z, cin, cout, 3):
        # round 6: the unit-stride instantiation of the window kernel of csrc/conv_s221.hip (z as the M dimension, Toeplitz A operand out of an LDS image of
        # the input columns): 368 us against 404 us for 18 -> 18 on 8 x 32 x 32 x 128, 5.2 ms against 6.0 ms on 8 x 128^3 (tools/conv_win_probe.py)
        rc = L.mdt_conv_win_forward(x.data_ptr(), wt.data_ptr(), bias.data_ptr() if bias is not None else None, 1 if relu else 0, y.data_ptr(),
                                    B, Y, X, Z, cin, cout, 3, _lib.raw_stream())
        if rc == 0:
            return y
        if rc != _lib.MDT_ERR_UNSUPPORTED:
            _lib.check(rc, "mdt_conv_win_forward")
    if bias is None and not relu:
        rc = L.mdt_conv3x3x3_small_forward(x.data_ptr(), wt.data_ptr(), y.data_ptr(), B, Y, X, Z, cin, cout, _lib.raw_stream())
    else:
        rc = L.mdt_conv3x3x3_small_forward_bias_act(x.data_ptr(), wt.data_ptr(), bias.data_ptr() if bias is not None else None, 1 if relu else 0,
                                                    y.data_ptr(), B, Y, X, Z, cin, cout, _lib.raw_stream())
    if rc != 0:
        _lib.check(rc, "mdt_conv3x3x3_small_forward")
    return y


CONV_WIN_WGRAD = True     # module switch (A/B: bench.py --conv-win-wgrad 0)


def conv_win_weight_grad(gy, x, w):
    """weight gradient of a size-preserving k x k x k layer with k * C_in <= 128, C_out <= 32 on the window kernel of csrc/conv_s221.hip at unit stride
    (mdt_conv_win_wgrad): M = the (kz, ci) window rows straight out of the channels-last input column, N = co, K = the voxels; None outside its budgets"""
    if not (CONV_WIN_WGRAD and gy.is_cuda and gy.dtype == torch.float32 and x.dtype == torch.float32 and x.dim() == 5 and w.dim() == 5 and _on_current_device(gy)
            and x.is_contiguous(memory_format=torch.channels_last_3d)):
        return None
    k = int(w.shape[2])
    if tuple(int(v) for v in w.shape[2:]) != (k, k, k):
        return None
    B, Ci, Y, X, Z = (int(v) for v in x.shape)
    Co = int(w.shape[0])
    L = _lib.lib()
    if int(w.shape[1]) != Ci or tuple(int(v) for v in gy.shape) != (B, Co, Y, X, Z) or not L.mdt_conv_win_wgrad_supported(B, Y, X, Z, Ci, Co, k):
        return None
    if not gy.is_contiguous(memory_format=torch.channels_last_3d):
        gy = gy.contiguous(memory_format=torch.channels_last_3d)
    ws = _workspace(L.mdt_conv_win_wgrad_workspace_bytes(B, Y, X, Z, Ci, Co, k), gy.device)
    gw = torch.empty((Co, k, k, k, Ci), dtype=torch.float32, device=gy.device)
    rc = L.mdt_conv_win_wgrad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), B, Y, X, Z, Ci, Co, k, ws.data_ptr(), ws.numel(), _lib.raw_stream())
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        return None
    if rc != 0:
        _lib.check(rc, "mdt_conv_win_wgrad")
    return gw.permute(0, 4, 1, 2, 3)


def conv3x3x3_small_weight_grad(gy, x, w):
    """weight gradient of the few-channel 3x3x3 layer on the same MFMA machinery (csrc/conv3x3x3_small.hip; MIOpen: 1078 us for
    18 -> 18 on 8 x 32x32x128).  None when the shape is not of that form."""
    if not (CONV3_SMALL and gy.is_cuda and gy.dtype == torch.float32 and x.dtype == torch.float32 and x.dim() == 5
            and tuple(w.shape[2:]) == (3, 3, 3) and _on_current_device(gy)):
        return None
    B, cin, Y, X, Z = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    L = _lib.lib()
    if B * Y * X * Z >= 65536 and _conv3_small_pays(cin, cout):
        # round 6: the window kernel at unit stride (mdt_conv_win_wgrad): 460 us against 598 us for 18 -> 18 on 8 x 32 x 32 x 128, 5.0 against 9.4 ms on 8 x 128^3,
        # and 36 -> 18 / 36 -> 32 which the older kernel does not cover (2.6 ms against MIOpen's 3.9; tools/conv_win_probe.py)
        gw = conv_win_weight_grad(gy, x, w)
        if gw is not None:
            return gw
    if not L.mdt_conv3x3x3_small_supported(Y, X, Z, cin, cout) or B * Y * X * Z < 65536 or not _conv3_small_pays(cin, cout) \
            or not x.is_contiguous(memory_format=torch.channels_last_3d):
        return None
    if not gy.is_contiguous(memory_format=torch.channels_last_3d):
        gy = gy.contiguous(memory_format=torch.channels_last_3d)
    wsb = L.mdt_conv3x3x3_small_wgrad_workspace_bytes(B, Y, X, cin, cout)
    ws = _workspace(wsb, gy.device)
    gw = torch.empty((3, 3, 3, cin, cout), dtype=torch.float32, device=gy.device)
    rc = L.mdt_conv3x3x3_small_wgrad(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), B, Y, X, Z, cin, cout, ws.data_ptr(), ws.numel(),
                                     _lib.raw_stream())
    if rc != 0:
        _lib.check(rc, "mdt_conv3x3x3_small_wgrad")
    return gw.permute(4, 3, 0, 1, 2)


class _ConvStride1(Function):
    """Unit-stride convolution whose input gradient is computed as a FORWARD convolution of the output gradient with the
    flipped, transposed filter (the textbook identity; same arithmetic up to fp32 summation order).  MIOpen's forward
    solvers for this backbone's 18..144-channel 3D layers are 1.3-2.4x faster than its backward-data solvers on gfx950
    (tools/conv_bwd_probe.py: 36->128 3x3x3 on 8x32x32x128: 5.1 ms -> 3.8 ms; 18->18: 1.25 -> 0.86 ms; 144->144 on
    4x4x16: 214 -> 88 us).  Used for the size-preserving layers (2*pad + 1 == kernel); the weight gradient stays on MIOpen's
    backward-weights path."""

    @staticmethod
    def forward(ctx, x, w, padding):
        ctx.save_for_backward(x, w)
        ctx.padding = padding
        if w.dim() == 5 and tuple(padding) == (1, 1, 1):
            y = conv3x3x3_small(x, w)
            if y is not None:
                return y
        return (F.conv3d if w.dim() == 5 else F.conv2d)(x, w, None, 1, padding)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx, gw = _stride1_grads(x, w, ctx.padding, gy, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, None


def _stride1_grads(x, w, padding, gy, need_x, need_w):
    """input and weight gradient of a unit-stride, size-preserving convolution (shared by _ConvStride1 and _Conv3SmallBiasReLU): the input
    gradient as a FORWARD convolution with the flipped / transposed filter, the weight gradient on this repo's kernels where they pay"""
    nd = w.dim() - 2
    gx = gw = None
    if need_x:
        mf = torch.contiguous_format if gy.is_contiguous() else (torch.channels_last_3d if nd == 3 else torch.channels_last)
        pad_t = tuple(int(k) - 1 - int(p) for k, p in zip(w.shape[2:], padding))
        wf = flip_transpose_filter(w, mf)
        gx = None
        if nd == 3 and pad_t == (1, 1, 1) and x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous():
            # the few-channel MFMA kernel runs on channels-last activations only: ask it only when the forward saw a channels-last x
            # (its gx then has x's layout), and convert gy only when the kernel WILL run -- an NCDHW gy of a 36-channel P2 layer is a
            # 151 MB copy that used to be made and thrown away when the shape was not the kernel's
            cin_t, cout_t = int(wf.shape[1]), int(wf.shape[0])
            Bq, _, Yq, Xq, Zq = (int(v) for v in gy.shape)
            if CONV3_SMALL and gy.is_cuda and gy.dtype == torch.float32 and Bq * Yq * Xq * Zq >= 65536 and _conv3_small_pays(cin_t, cout_t) \
                    and _lib.lib().mdt_conv3x3x3_small_supported(Yq, Xq, Zq, cin_t, cout_t):
                gx = conv3x3x3_small(gy if gy.is_contiguous(memory_format=torch.channels_last_3d) else gy.contiguous(memory_format=torch.channels_last_3d), wf)
        if gx is None:
            gx = (F.conv3d if nd == 3 else F.conv2d)(gy, wf, None, 1, pad_t)
    if need_w:
        gw = conv1x1_weight_grad(gy, x, w) if WGRAD_1X1 else None
        if gw is None and nd == 3 and tuple(padding) == (1, 1, 1):
            gw = conv3x3x3_small_weight_grad(gy, x, w)
        if gw is None:
            gw = torch.ops.aten.convolution_backward(gy, x, w, None, [1] * nd, list(padding), [1] * nd, False, [0] * nd, 1,
                                                     [False, True, False])[1]
    return gx, gw


BIAS_GRAD_IN_LAUNCH = False     # module switch (A/B): the bias gradient's second stage inside the backward epilogue's launch (mdt_bias_act_backward_ticket).
                                # MEASURED NEGATIVE (profiles/r06/r06_bias_grad_in_launch_probe.txt): the last block reads the partial rows at the cross-XCD rate
_TICKET = {}                    # device -> int32[1], zero between launches


def _bias_act_bwd(gy, y, relu, mf):
    """g = gy * (y > 0) (when relu) and the bias gradient (per-channel sum of g) in one pass (csrc/epilogue.hip), channels-last or contiguous"""
    if not gy.is_contiguous(memory_format=mf):
        gy = gy.contiguous(memory_format=mf)
    L = _lib.lib()
    n, C = gy.numel(), int(gy.shape[1])
    inner = 1 if mf != torch.contiguous_format else int(gy.shape[2:].numel())
    g = torch.empty_like(gy) if (relu or not BIAS_BWD_NO_COPY) else None       # no activation: the kernel only reduces, g is gy
    gbias = torch.empty(C, dtype=torch.float32, device=gy.device)
    wsb = (4096 * C * 4 + 256) if inner == 1 else L.mdt_bias_act_backward_workspace_bytes(n, C, inner)
    if inner == 1 and BIAS_GRAD_IN_LAUNCH:
        # one launch: the per-block partial rows are folded by the last block of the same launch (a ticket this module owns; every call of a
        # device's backward runs on one stream, so launches sharing the ticket and the workspace are ordered)
        tk = _TICKET.get(gy.device)
        if tk is None:
            tk = _TICKET[gy.device] = torch.zeros(1, dtype=torch.int32, device=gy.device)
        ws = _workspace(wsb, gy.device)
        rc = L.mdt_bias_act_backward_ticket(g.data_ptr() if g is not None else None, gy.data_ptr(), y.data_ptr() if relu else None, gbias.data_ptr(), n, C, inner,
                                            1 if relu else 0, ws.data_ptr(), ws.numel(), tk.data_ptr(), _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_bias_act_backward_ticket")
        return (g if g is not None else gy), gbias
    ws = _workspace(wsb, gy.device)
    rc = L.mdt_bias_act_backward(g.data_ptr() if g is not None else None, gy.data_ptr(), y.data_ptr() if relu else None, gbias.data_ptr(), n, C, inner,
                                 1 if relu else 0, ws.data_ptr(), ws.numel(), _lib.raw_stream())
    if rc != 0:
        _lib.check(rc, "mdt_bias_act_backward")
    return (g if g is not None else gy), gbias


CONV3_SMALL_EPILOGUE = True   # module switch: bias + ReLU of the few-channel 3x3x3 layers inside the convolution kernel


def _conv3_small_fused_applies(conv, x):
    """ConvBiasReLU layer that csrc/conv3x3x3_small.hip runs with its bias + ReLU epilogue: 3x3x3, unit stride, pad 1, few odd-sized channel
    counts, channels-last fp32 activation of >= 65 536 voxels, in training (the fused Function owns the backward)"""
    if not (ENABLED and BWD_DATA_AS_FWD and CONV3_SMALL and CONV3_SMALL_EPILOGUE and isinstance(conv, nn.Conv3d) and conv.bias is not None
            and conv.groups == 1 and tuple(conv.kernel_size) == (3, 3, 3) and _unit(conv.stride) and _unit(conv.dilation)
            and not isinstance(conv.padding, str) and tuple(int(p) for p in conv.padding) == (1, 1, 1)):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and _on_current_device(x) and torch.is_grad_enabled()
            and (x.requires_grad or conv.weight.requires_grad) and not torch.is_autocast_enabled()
            and x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous()):
        return False
    B, cin, Y, X, Z = (int(v) for v in x.shape)
    cout = int(conv.out_channels)
    return cin == int(conv.in_channels) and B * Y * X * Z >= 65536 and _conv3_small_pays(cin, cout) \
        and bool(_lib.lib().mdt_conv3x3x3_small_supported(Y, X, Z, cin, cout))


class _Conv3SmallBiasReLU(Function):
    """relu(conv3x3x3(x) + bias) of a few-channel layer with the bias add and the ReLU in the convolution kernel's epilogue (no separate
    pass over the output); backward = the fused ReLU-mask / bias-gradient pass + the gradients of `_ConvStride1`"""

    @staticmethod
    def forward(ctx, x, w, bias):
        y = conv3x3x3_small(x, w, bias.detach(), True)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        g, gbias = _bias_act_bwd(gy, y, True, torch.channels_last_3d)
        gx, gw = _stride1_grads(x, w, (1, 1, 1), g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, gbias


STEM_WGRAD = True   # module switch (A/B: bench.py --stem-wgrad 0)


def stem_weight_grad(gy, x, w, stride, xp=None):
    """Weight gradient of a one-channel k x k x k convolution with stride (sy, sx, 1) and padding k // 2 on the fp32-MFMA kernel of
    csrc/conv_stem_wgrad.hip (MIOpen: 3.7 ms for the 7x7x7 stem on 8 x 128^3).  None when the layer is not of that form."""
    if not (STEM_WGRAD and gy.is_cuda and gy.dtype == torch.float32 and x.dtype == torch.float32 and x.dim() == 5 and int(x.shape[1]) == 1
            and int(w.shape[1]) == 1 and int(stride[2]) == 1 and _on_current_device(gy)):
        return None
    k = int(w.shape[2])
    cout = int(w.shape[0])
    B, _, OY, OX, OZ = (int(v) for v in gy.shape)
    if tuple(int(v) for v in w.shape[2:]) != (k, k, k) or k % 2 == 0 or k ** 3 > 384 or cout > 32 or OZ % 8 != 0:
        return None
    pad = k // 2
    if (int(x.shape[2]) + 2 * pad - k) // int(stride[0]) + 1 != OY or (int(x.shape[3]) + 2 * pad - k) // int(stride[1]) + 1 != OX or int(x.shape[4]) != OZ:
        return None
    if not gy.is_contiguous(memory_format=torch.channels_last_3d):
        gy = gy.contiguous(memory_format=torch.channels_last_3d)
    if xp is None:      # (the forward kernel's padded copy is passed on when it ran)
        xp = F.pad(x.reshape(B, int(x.shape[2]), int(x.shape[3]), int(x.shape[4])), (pad, pad, pad, pad, pad, pad)).contiguous()
    L = _lib.lib()
    wsb = L.mdt_conv_stem_wgrad_workspace_bytes(cout, k)
    ws = _workspace(wsb, gy.device)
    gw = torch.empty((cout, k * k * k), dtype=torch.float32, device=gy.device)
    rc = L.mdt_conv_stem_wgrad(gy.data_ptr(), xp.data_ptr(), gw.data_ptr(), B, OY, OX, OZ, cout, k, int(stride[0]), int(stride[1]),
                               int(xp.shape[1]), int(xp.shape[2]), int(xp.shape[3]), ws.data_ptr(), ws.numel(), _lib.raw_stream())
    if rc != 0:
        _lib.check(rc, "mdt_conv_stem_wgrad")
    return gw.view(w.shape)


STEM_FWD = True     # module switch (A/B: bench.py --stem-fwd 0)


def stem_forward_supported(x, w):
    """the one-channel 7 x 7 x 7 stride-(2, 2, 1) pad-3 stem in a shape csrc/conv_stem_fwd.hip handles"""
    if not (STEM_FWD and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 5 and int(x.shape[1]) == 1
            and int(w.shape[1]) == 1 and tuple(int(v) for v in w.shape[2:]) == (7, 7, 7) and _on_current_device(x)
            and not torch.is_autocast_enabled()):
        return False
    Y, X, Z = (int(v) for v in x.shape[2:])
    return Y % 2 == 0 and X % 2 == 0 and bool(_lib.lib().mdt_conv_stem_forward_supported(Y // 2, X // 2, Z, int(w.shape[0]), 7, 2, 2))


def stem_forward(x, w, bias=None, relu=False):
    """The one-channel 7 x 7 x 7 stride-(2, 2, 1) stem on the fp32-MFMA kernel of csrc/conv_stem_fwd.hip (810 us + a 35 us padding
    copy on 8 x 128^3; MIOpen: 1976 us in space-to-depth form), optionally with the bias add and the ReLU in its epilogue.
    Returns (out channels-last, padded input) or None when the layer is not of that form."""
    if not stem_forward_supported(x, w):
        return None
    B, _, Y, X, Z = (int(v) for v in x.shape)
    cout = int(w.shape[0])
    OY, OX = Y // 2, X // 2
    L = _lib.lib()
    xp = F.pad(x.reshape(B, Y, X, Z), (3, 3, 3, 3, 3, 3)).contiguous()
    out = torch.empty((B, cout, OY, OX, Z), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d)
    wc = w.detach().reshape(cout, 343).contiguous()
    rc = L.mdt_conv_stem_forward(xp.data_ptr(), wc.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), B, OY, OX, Z,
                                 cout, 7, 2, 2, Y + 6, X + 6, Z + 6, 1 if relu else 0, _lib.raw_stream())
    if rc != 0:
        _lib.check(rc, "mdt_conv_stem_forward")
    return out, xp


# ---- space-to-depth form of a stride-(2, 2, 1) convolution (any channel count) --------------------------------------------------------------
S2D_GENERAL = True      # module switch (A/B: bench.py --conv-s2d 0): stride-(2, 2, 1) layers with many input channels in space-to-depth form


def _cl_rows(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.is_contiguous(memory_format=torch.channels_last_3d) and _on_current_device(x)


def s2d_input(x, k):
    """[B, C, Y, X, Z] -> the 2 x 2 (y, x) phases of the input padded by k // 2 as channels: [B, 4C, (Y + 2p) / 2, (X + 2p) / 2, Z + 2p]
    in channels-last storage, channel = (c, py, px).  On the GPU one pass of csrc/conv_s221.hip (torch: a padded copy + a permuted copy)."""
    p = k // 2
    B, C = int(x.shape[0]), int(x.shape[1])
    if _cl_rows(x) and int(x.shape[2]) % 2 == 0 and int(x.shape[3]) % 2 == 0:
        Y, X, Z = (int(v) for v in x.shape[2:])
        xs = torch.empty((B, (Y + 2 * p) // 2, (X + 2 * p) // 2, Z + 2 * p, 4 * C), dtype=torch.float32, device=x.device)
        rc = _lib.lib().mdt_s2d221_input(x.data_ptr(), xs.data_ptr(), B, C, Y, X, Z, k, _lib.raw_stream())
        if rc == 0:
            return xs.permute(0, 4, 1, 2, 3)
        if rc != _lib.MDT_ERR_UNSUPPORTED:
            _lib.check(rc, "mdt_s2d221_input")
    xp = F.pad(x, (p, p, p, p, p, p))
    Y, X, Z = (int(v) for v in xp.shape[2:])
    return xp.view(B, C, Y // 2, 2, X // 2, 2, Z).permute(0, 2, 4, 6, 1, 3, 5).reshape(B, Y // 2, X // 2, Z, 4 * C).permute(0, 4, 1, 2, 3)


def s2d_filter(w):
    """[O, C, k, k, k] -> [O, 4C, (k+1)/2, (k+1)/2, k] (the taps outside the k x k window are zero), channels-last"""
    O, C, k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
    h = (k + 1) // 2
    ws = F.pad(w, (0, 0, 0, 1, 0, 1)).view(O, C, h, 2, h, 2, k).permute(0, 1, 3, 5, 2, 4, 6).reshape(O, 4 * C, h, h, k)
    return ws.contiguous(memory_format=torch.channels_last_3d)


def s2d_input_grad_conv(gy, ws):
    """gradient w.r.t. the space-to-depth input, as a FORWARD convolution of the zero-padded output gradient with the flipped filter"""
    h, k = int(ws.shape[2]), int(ws.shape[4])
    wf = ws.flip(2, 3, 4).transpose(0, 1).contiguous(memory_format=torch.channels_last_3d)
    return F.conv3d(F.pad(gy, (k - 1, k - 1, h - 1, h - 1, h - 1, h - 1)), wf, None, 1, 0)


def s2d_input_grad_fold(gxs, x_shape, k):
    """[B, 4C, Y'/2, X'/2, Z'] (gradient of s2d_input's output) -> gradient of x, channels-last, in one pass (csrc/conv_s221.hip on the GPU;
    torch's strided copy takes 4.5 ms for the 2.6 GB of the Retina U-Net's C1 layer, the kernel moves them at HBM speed)"""
    p = k // 2
    B, C, Y, X, Z = (int(v) for v in x_shape)
    Y2, X2, Zp = (int(v) for v in gxs.shape[2:])
    if _cl_rows(gxs) and Y % 2 == 0 and X % 2 == 0 and (Y2, X2, Zp) == ((Y + 2 * p) // 2, (X + 2 * p) // 2, Z + 2 * p):
        gx = torch.empty((B, Y, X, Z, C), dtype=torch.float32, device=gxs.device)
        rc = _lib.lib().mdt_s2d221_fold_input_grad(gxs.data_ptr(), gx.data_ptr(), B, C, Y, X, Z, k, _lib.raw_stream())
        if rc == 0:
            return gx.permute(0, 4, 1, 2, 3)
        if rc != _lib.MDT_ERR_UNSUPPORTED:
            _lib.check(rc, "mdt_s2d221_fold_input_grad")
    g = gxs.view(B, C, 2, 2, Y2, X2, Zp).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, C, 2 * Y2, 2 * X2, Zp)
    return g[:, :, p:p + Y, p:p + X, p:p + Z].contiguous(memory_format=torch.channels_last_3d)


def s2d_filter_grad_fold(gws, w_shape):
    """gradient of s2d_filter's output -> gradient of w"""
    O, C, k = int(w_shape[0]), int(w_shape[1]), int(w_shape[2])
    h = (k + 1) // 2
    g = gws.reshape(O, C, 2, 2, h, h, k).permute(0, 1, 4, 2, 5, 3, 6).reshape(O, C, 2 * h, 2 * h, k)
    return g[:, :, :k, :k, :].contiguous(memory_format=torch.channels_last_3d)


S221_WGRAD = True      # module switch: weight gradient of those layers on the fp32-MFMA kernel of csrc/conv_s221.hip


def s221_weight_grad(gy, x, w):
    """Weight gradient of a k x k x k, stride (2, 2, 1), pad k // 2 convolution with k * C_in <= 128 and C_out <= 32 (the Retina U-Net's C1:
    18 -> 18, k = 7) on the fp32-MFMA kernel of csrc/conv_s221.hip (MIOpen: 45.9 ms on 8 x 128^3).  None when the layer is not of that form."""
    if not (S221_WGRAD and _cl_rows(x) and gy.is_cuda and gy.dtype == torch.float32 and w.dim() == 5):
        return None
    k = int(w.shape[2])
    if tuple(int(v) for v in w.shape[2:]) != (k, k, k):
        return None
    B, Ci, Y, X, Z = (int(v) for v in x.shape)
    Co = int(w.shape[0])
    L = _lib.lib()
    if int(w.shape[1]) != Ci or tuple(int(v) for v in gy.shape) != (B, Co, Y // 2, X // 2, Z) or not L.mdt_conv_s221_wgrad_supported(B, Y, X, Z, Ci, Co, k):
        return None
    if not gy.is_contiguous(memory_format=torch.channels_last_3d):
        gy = gy.contiguous(memory_format=torch.channels_last_3d)
    ws = _workspace(L.mdt_conv_s221_wgrad_workspace_bytes(B, Y, X, Z, Ci, Co, k), gy.device)
    gw = torch.empty((Co, k, k, k, Ci), dtype=torch.float32, device=gy.device)
    rc = L.mdt_conv_s221_wgrad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), B, Y, X, Z, Ci, Co, k, ws.data_ptr(), ws.numel(), _lib.raw_stream())
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        return None
    if rc != 0:
        _lib.check(rc, "mdt_conv_s221_wgrad")
    return gw.permute(0, 4, 1, 2, 3)


S221_FWD = True     # module switch (A/B: bench.py --conv-s221-fwd 0): the forward of the stride-(2, 2, 1) many-channel layer on this repo's fp32-MFMA kernel


def s221_forward(x, w, bias=None, relu=False):
    """mdt_conv_s221_forward (csrc/conv_s221.hip): the layer's forward straight from the channels-last input -- an LDS image of the input columns read as a
    Toeplitz A operand, the (ky, kx) filter slices as B -- no space-to-depth copy; None when the shape is outside the kernel's budgets or x is not fp32
    channels-last on the current device"""
    if not (S221_FWD and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 5 and _on_current_device(x)
            and x.is_contiguous(memory_format=torch.channels_last_3d) and not torch.is_autocast_enabled()):
        return None
    B, Ci, Y, X, Z = (int(v) for v in x.shape)
    Co, k = int(w.shape[0]), int(w.shape[2])
    L = _lib.lib()
    if not L.mdt_conv_s221_forward_supported(Y, X, Z, Ci, Co, k):
        return None
    wt = w.detach().permute(2, 3, 4, 1, 0).contiguous()
    y = torch.empty((B, Co, Y // 2, X // 2, Z), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d)
    rc = L.mdt_conv_s221_forward(x.data_ptr(), wt.data_ptr(), bias.data_ptr() if bias is not None else None, 1 if relu else 0, y.data_ptr(), B, Y, X, Z, Ci, Co, k,
                                 _lib.raw_stream())
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        return None
    if rc != 0:
        _lib.check(rc, "mdt_conv_s221_forward")
    return y


S221_DGRAD = True   # module switch (A/B: bench.py --conv-s221-dgrad 0): the layer's input gradient on this repo's fp32-MFMA kernel


def s221_input_grad(gy, w, x_shape):
    """mdt_conv_s221_input_grad: the input gradient straight from the channels-last output gradient (no padded copy, no space-to-depth problem, no fold);
    None outside the kernel's budgets"""
    if not (S221_DGRAD and gy.is_cuda and gy.dtype == torch.float32 and w.dtype == torch.float32 and gy.dim() == 5 and _on_current_device(gy)
            and gy.is_contiguous(memory_format=torch.channels_last_3d) and not torch.is_autocast_enabled()):
        return None
    B, Ci, Y, X, Z = (int(v) for v in x_shape)
    Co, k = int(w.shape[0]), int(w.shape[2])
    L = _lib.lib()
    if tuple(gy.shape) != (B, Co, Y // 2, X // 2, Z) or not L.mdt_conv_s221_input_grad_supported(Y, X, Z, Ci, Co, k):
        return None
    wd = w.detach().flip(4).permute(2, 3, 4, 0, 1).contiguous()
    gx = torch.empty((B, Ci, Y, X, Z), dtype=torch.float32, device=gy.device, memory_format=torch.channels_last_3d)
    rc = L.mdt_conv_s221_input_grad(gy.data_ptr(), wd.data_ptr(), gx.data_ptr(), B, Y, X, Z, Ci, Co, k, _lib.raw_stream())
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        return None
    if rc != 0:
        _lib.check(rc, "mdt_conv_s221_input_grad")
    return gx


class _ConvS2D221(Function):
    """k x k x k, stride (2, 2, 1), pad k // 2 convolution with MANY input channels (backbone.py:84: the Retina U-Net's C1 = 18 -> 18, k = 7
    on the full-resolution C0 output; 40 % of the config-2 step on MIOpen's direct problem: 39.6 ms forward, 58.1 ms input gradient, 45.9 ms
    weight gradient at 8 x 128^3).  Posed in space-to-depth form (4 C_in channels, ((k+1)/2, (k+1)/2, k) filter, unit stride) the forward takes
    33.2 ms and the input gradient becomes a FORWARD convolution of the padded output gradient with the flipped filter: 26.2 ms
    (tools/c1_probe.py).  The weight gradient runs on this repo's fp32-MFMA kernel (`s221_weight_grad`), else on MIOpen's direct problem."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        y = s221_forward(x, w)
        if y is not None:
            return y
        return F.conv3d(s2d_input(x, int(w.shape[2])), s2d_filter(w), None, 1, 0)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        k = int(w.shape[2])
        p = k // 2
        if not gy.is_contiguous(memory_format=torch.channels_last_3d):
            gy = gy.contiguous(memory_format=torch.channels_last_3d)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = s221_input_grad(gy, w, x.shape)
            if gx is None:
                gx = s2d_input_grad_fold(s2d_input_grad_conv(gy, s2d_filter(w)), x.shape, k)
        if ctx.needs_input_grad[1]:
            gw = s221_weight_grad(gy, x, w)
            if gw is None:
                gw = torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2, 1], [p, p, p], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False])[1]
        return gx, gw


def _is_s221_general(conv, x):
    k = conv.kernel_size
    return x.dim() == 5 and tuple(conv.stride) == (2, 2, 1) and k[0] == k[1] == k[2] and k[0] % 2 == 1 and k[0] >= 3 \
        and tuple(conv.padding) == (k[0] // 2,) * 3 and 4 < conv.in_channels <= 64 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0


class _ConvStem221(Function):
    """The stem: few input channels, odd k x k x k filter, stride (2, 2, 1), pad k // 2 (backbone.py:66-68: 1 -> 18, 7x7x7).
    Forward in space-to-depth form: the 2 x 2 (y, x) phases of the padded input become 4x the input channels and the filter
    becomes ((k+1)/2, (k+1)/2, k) with stride 1 (the taps outside the k x k window are zero) -- the same sums, but MIOpen's
    channels-last kernels get 16-byte loads instead of scalar ones (tools/stem_probe.py: 3.14 ms -> 1.29 ms at 8 x 128^3).
    The weight gradient is taken on the ORIGINAL problem (MIOpen's backward-weights is faster there than on the
    space-to-depth problem)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.xp = None
        r = stem_forward(x, w)
        if r is not None:       # own kernel (the padded copy serves the weight gradient too)
            ctx.save_for_backward(x, w)
            ctx.xp = r[1]
            return r[0]
        ctx.save_for_backward(x, w)
        k = int(w.shape[2])
        p = k // 2
        B, C = x.shape[0], x.shape[1]
        xp = F.pad(x, (p, p, p, p, p, p))
        Y, X, Z = xp.shape[2:]
        # [B, C, Y/2, 2, X/2, 2, Z] -> channels-last memory [B, Y/2, X/2, Z, (C, py, px)] in ONE copy
        xs = xp.view(B, C, Y // 2, 2, X // 2, 2, Z).permute(0, 2, 4, 6, 1, 3, 5).reshape(B, Y // 2, X // 2, Z, 4 * C).permute(0, 4, 1, 2, 3)
        O, h = w.shape[0], (k + 1) // 2
        ws = F.pad(w, (0, 0, 0, 1, 0, 1)).view(O, C, h, 2, h, 2, k).permute(0, 1, 3, 5, 2, 4, 6).reshape(O, 4 * C, h, h, k)
        return F.conv3d(xs, ws.contiguous(memory_format=torch.channels_last_3d), None, 1, 0)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        p = int(w.shape[2]) // 2
        gw = stem_weight_grad(gy, x, w, (2, 2, 1), xp=ctx.xp) if ctx.needs_input_grad[1] else None
        ctx.xp = None
        need_w = bool(ctx.needs_input_grad[1]) and gw is None
        gx = None
        if ctx.needs_input_grad[0] or need_w:
            gx, gw2, _ = torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2, 1], [p, p, p], [1, 1, 1], False, [0, 0, 0], 1,
                                                             [bool(ctx.needs_input_grad[0]), need_w, False])
            if need_w:
                gw = gw2
        return (gx if ctx.needs_input_grad[0] else None), (gw if ctx.needs_input_grad[1] else None)


class _ConvStemBiasReLU(Function):
    """relu(stem(x) + bias) with the bias add and the ReLU in the convolution kernel's epilogue (no separate pass over the
    302 MB stem output); backward = the fused ReLU-mask / bias-gradient pass of csrc/epilogue.hip + the stem weight gradient"""

    @staticmethod
    def forward(ctx, x, w, bias):
        y, xp = stem_forward(x, w, bias.detach(), True)
        ctx.save_for_backward(x, w, y)
        ctx.xp = xp
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        mf = torch.channels_last_3d
        if not gy.is_contiguous(memory_format=mf):
            gy = gy.contiguous(memory_format=mf)
        L = _lib.lib()
        n, C = gy.numel(), int(gy.shape[1])
        g = torch.empty_like(gy)
        gbias = torch.empty(C, dtype=torch.float32, device=gy.device)
        ws = _workspace(4096 * C * 4 + 256, gy.device)
        rc = L.mdt_bias_act_backward(g.data_ptr(), gy.data_ptr(), y.data_ptr(), gbias.data_ptr(), n, C, 1, 1, ws.data_ptr(), ws.numel(),
                                     _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_bias_act_backward")
        xp, ctx.xp = ctx.xp, None
        gw = stem_weight_grad(g, x, w, (2, 2, 1), xp=xp) if ctx.needs_input_grad[1] else None
        need_w = bool(ctx.needs_input_grad[1]) and gw is None
        gx = None
        if ctx.needs_input_grad[0] or need_w:
            gx, gw2, _ = torch.ops.aten.convolution_backward(g, x, w, None, [2, 2, 1], [3, 3, 3], [1, 1, 1], False, [0, 0, 0], 1,
                                                             [bool(ctx.needs_input_grad[0]), need_w, False])
            if need_w:
                gw = gw2
        return (gx if ctx.needs_input_grad[0] else None), (gw if ctx.needs_input_grad[1] else None), gbias


CONV_C0 = True   # module switch (A/B: bench.py --conv-c0 0): the one-channel 3x3x3 first layer of the stride-1 backbone on this repo's kernels (csrc/conv_c0.hip)


class _ConvC0BiasReLU(Function):
    """relu(conv3x3x3(x, w) + bias) for a ONE-channel volume, output in channels-last storage (models/backbone.py:60-63: C0[0] of the Retina U-Net).  The
    library runs this layer row-major (a one-channel tensor is contiguous in both layouts) between layout transposes, and the next layer needs a
    channels-last copy of the 2.4 GB result; the backward pays the same conversions again.  Forward: mdt_conv_c0_forward; backward: weight and bias
    gradient from ONE pass over gy and y (mdt_conv_c0_backward).  The input has no gradient."""

    @staticmethod
    def forward(ctx, x, w, bias, relu):
        B, _, Y, X, Z = x.shape
        co = int(w.shape[0])
        xc = x.detach().contiguous()
        y = torch.empty((B, co, Y, X, Z), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d)
        wd = w.detach().reshape(co, 27).t().contiguous()          # [27][18]: a tap's filter values are one contiguous (wave-uniform) read
        rc = _lib.lib().mdt_conv_c0_forward(xc.data_ptr(), wd.data_ptr(), bias.detach().data_ptr() if bias is not None else None, 1 if relu else 0, y.data_ptr(),
                                            B, Y, X, Z, co, _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_conv_c0_forward")
        ctx.relu, ctx.has_bias = relu, bias is not None
        ctx.w_shape = tuple(w.shape)
        ctx.save_for_backward(xc, y) if relu else ctx.save_for_backward(xc)
        return y

    @staticmethod
    def backward(ctx, gy):
        xc = ctx.saved_tensors[0]
        y = ctx.saved_tensors[1] if ctx.relu else None
        mf = torch.channels_last_3d
        if not gy.is_contiguous(memory_format=mf):
            gy = gy.contiguous(memory_format=mf)
        B, co, Y, X, Z = gy.shape
        L = _lib.lib()
        gw = torch.empty((co, 27), dtype=torch.float32, device=gy.device)
        gb = torch.empty(co, dtype=torch.float32, device=gy.device)
        ws = _workspace(L.mdt_conv_c0_wgrad_workspace_bytes(B, Y, X, Z), gy.device)
        rc = L.mdt_conv_c0_backward(gy.data_ptr(), y.data_ptr() if y is not None else None, xc.data_ptr(), 1 if ctx.relu else 0, gw.data_ptr(), gb.data_ptr(), B, Y, X, Z, co,
                                    ws.data_ptr(), ws.numel(), _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_conv_c0_backward")
        return None, gw.view(ctx.w_shape), (gb if ctx.has_bias else None), None


def conv_c0_applies(seq, x):
    """seq: the ConvBiasReLU Sequential of a one-channel 3x3x3 unit-stride layer; x: its fp32 GPU input [B, 1, Y, X, Z] that needs no gradient"""
    if not (ENABLED and CONV_C0 and isinstance(seq, ConvBiasReLU) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and not x.requires_grad
            and not torch.is_autocast_enabled() and _on_current_device(x)):
        return False
    conv = seq[0]
    if not (isinstance(conv, nn.Conv3d) and conv.groups == 1 and _unit(conv.stride) and _unit(conv.dilation) and not isinstance(conv.padding, str)
            and tuple(int(k) for k in conv.kernel_size) == (3, 3, 3) and tuple(int(v) for v in conv.padding) == (1, 1, 1) and conv.weight.dtype == torch.float32):
        return False
    return bool(_lib.lib().mdt_conv_c0_supported(int(conv.in_channels), int(conv.out_channels), 3, int(x.shape[4])))


def conv_c0_bias_relu(seq, x):
    conv = seq[0]
    return _ConvC0BiasReLU.apply(x, conv.weight, conv.bias, True)


SEG_HEAD_COMPOSED = True   # module switch (A/B: bench.py --seg-head-composed 0): final_conv o P0_conv2 of the Retina U-Net as one 36 -> 2 3x3x3 layer (csrc/conv_seg.hip)


class _ConvSeg(Function):
    """y = conv3x3x3(x, w) + bias for a 36-channel channels-last map and a two-channel result (csrc/conv_seg.hip): forward, input gradient and weight / bias
    gradient on this repo's kernels (VALU forward / input gradient, fp32-MFMA weight gradient).  w: [2, 36, 3, 3, 3] (a composed filter: a non-leaf)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        B, C, Y, X, Z = x.shape
        S = int(w.shape[0])
        wd = w.detach()
        wt = wd.permute(2, 3, 4, 1, 0).contiguous()                      # [27][36][2]
        y = torch.empty((B, S, Y, X, Z), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d)
        bd = bias.detach().contiguous()
        rc = _lib.lib().mdt_conv_seg_forward(x.data_ptr(), wt.data_ptr(), bd.data_ptr(), y.data_ptr(), B, Y, X, Z, C, S, _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_conv_seg_forward")
        ctx.save_for_backward(x, wd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wd = ctx.saved_tensors
        B, C, Y, X, Z = x.shape
        S = int(wd.shape[0])
        mf = torch.channels_last_3d
        if not gy.is_contiguous(memory_format=mf):
            gy = gy.contiguous(memory_format=mf)
        L = _lib.lib()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wflip = wd.flip(2, 3, 4).permute(2, 3, 4, 0, 1).contiguous()   # [27][2][36], taps mirrored
            gx = torch.empty_like(x)
            rc = L.mdt_conv_seg_input_grad(gy.data_ptr(), wflip.data_ptr(), gx.data_ptr(), B, Y, X, Z, C, S, _lib.raw_stream())
            if rc != 0:
                _lib.check(rc, "mdt_conv_seg_input_grad")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gw = torch.empty((S, C, 3, 3, 3), dtype=torch.float32, device=x.device)
            gb = torch.empty(S, dtype=torch.float32, device=x.device)
            ws = _workspace(L.mdt_conv_seg_wgrad_workspace_bytes(B, Y, X, Z), x.device)
            rc = L.mdt_conv_seg_weight_grad(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, Y, X, Z, C, S, ws.data_ptr(), ws.numel(), _lib.raw_stream())
            if rc != 0:
                _lib.check(rc, "mdt_conv_seg_weight_grad")
        return gx, gw, gb


def seg_head_composed_applies(conv2, final_conv, x):
    """conv2: the 3x3x3 bias-only layer (P0_conv2), final_conv: the 1x1x1 bias-only layer behind it, x: conv2's input -- channels-last fp32 on this GPU, shapes
    of csrc/conv_seg.hip"""
    if not (ENABLED and SEG_HEAD_COMPOSED and isinstance(conv2, ConvBias) and isinstance(final_conv, ConvBias) and isinstance(conv2, nn.Conv3d)
            and isinstance(final_conv, nn.Conv3d) and conv2.bias is not None and final_conv.bias is not None and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 5 and not torch.is_autocast_enabled() and _on_current_device(x)):
        return False
    if not (tuple(int(k) for k in conv2.kernel_size) == (3, 3, 3) and tuple(int(v) for v in conv2.padding) == (1, 1, 1) and _unit(conv2.stride) and _unit(conv2.dilation)
            and conv2.groups == 1 and all(int(k) == 1 for k in final_conv.kernel_size) and _unit(final_conv.stride) and final_conv.groups == 1
            and not any(int(v) for v in final_conv.padding) and int(final_conv.in_channels) == int(conv2.out_channels)):
        return False
    if not (x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous() and x.data_ptr() % 16 == 0):
        return False
    return bool(_lib.lib().mdt_conv_seg_supported(int(conv2.in_channels), int(final_conv.out_channels), int(x.shape[2]), int(x.shape[3]), int(x.shape[4])))


def seg_head_composed(conv2, final_conv, x):
    """final_conv(conv2(x)) as ONE convolution: the two layers are linear with nothing between them, so W' = Wf . W2 and b' = Wf b2 + bf (built from the modules'
    parameters with differentiable ops: autograd splits the composed filter's gradient between them)"""
    S, Cm = int(final_conv.out_channels), int(conv2.out_channels)
    wf = final_conv.weight.reshape(S, Cm)
    w = (wf @ conv2.weight.reshape(Cm, -1)).reshape(S, int(conv2.in_channels), 3, 3, 3)
    b = wf @ conv2.bias + final_conv.bias
    return _ConvSeg.apply(x, w, b)


STEM_POOL_FUSED = True   # module switch (A/B: bench.py --stem-pool-fused 0)


class _ConvStemBiasReLUPool(Function):
    """maxpool_{3, (2, 2, 1), 1}(relu(stem(x) + bias)) as ONE autograd node (models/backbone.py:128-131 when the stem output has no other consumer).
    Forward: the two kernels _ConvStemBiasReLU and _MaxPoolK3S221 run.  Backward: the ReLU mask and the bias gradient are applied at the POOLED
    resolution BEFORE the pooling backward -- a pooled value is the stem output at its arg-max tap, so (y[tap] > 0) == (p > 0), and the bias gradient
    sum_taps g[tap] == sum_windows gp * (p > 0).  The mask / bias pass then touches the 75 MB pooled tensors instead of the 302 MB stem output
    (226 MB of traffic instead of 906 MB at the benchmark patch), and the stem output itself is not kept for the backward (302 MB of activations)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        y, xp = stem_forward(x, w, bias.detach(), True)
        B, C, Y, X, Z = y.shape
        OY, OX = (Y - 1) // 2 + 1, (X - 1) // 2 + 1
        p = torch.empty((B, C, OY, OX, Z), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d)
        arg = torch.empty((B, OY, OX, Z, C), dtype=torch.uint8, device=x.device)
        rc = _lib.lib().mdt_maxpool3d_k3s221_cl_forward(y.data_ptr(), p.data_ptr(), arg.data_ptr(), B, Y, X, Z, C, _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_maxpool3d_k3s221_cl_forward")
        ctx.save_for_backward(x, w, p, arg)
        ctx.xp = xp
        ctx.y_shape = (B, C, Y, X, Z)
        return p

    @staticmethod
    def backward(ctx, gp):
        x, w, p, arg = ctx.saved_tensors
        gm, gbias = _bias_act_bwd(gp, p, True, torch.channels_last_3d)          # gp * (p > 0) and its per-channel sum, at the pooled size
        B, C, Y, X, Z = ctx.y_shape
        g = torch.empty((B, C, Y, X, Z), dtype=torch.float32, device=gp.device, memory_format=torch.channels_last_3d)
        rc = _lib.lib().mdt_maxpool3d_k3s221_cl_backward(gm.data_ptr(), arg.data_ptr(), g.data_ptr(), B, Y, X, Z, C, _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_maxpool3d_k3s221_cl_backward")
        xp, ctx.xp = ctx.xp, None
        gw = stem_weight_grad(g, x, w, (2, 2, 1), xp=xp) if ctx.needs_input_grad[1] else None
        need_w = bool(ctx.needs_input_grad[1]) and gw is None
        gx = None
        if ctx.needs_input_grad[0] or need_w:
            gx, gw2, _ = torch.ops.aten.convolution_backward(g, x, w, None, [2, 2, 1], [3, 3, 3], [1, 1, 1], False, [0, 0, 0], 1,
                                                             [bool(ctx.needs_input_grad[0]), need_w, False])
            if need_w:
                gw = gw2
        return (gx if ctx.needs_input_grad[0] else None), (gw if ctx.needs_input_grad[1] else None), gbias


def stem_pool_fused_applies(stem, pool, x):
    """stem: the ConvBiasReLU Sequential of the one-channel 7x7x7 stride-(2, 2, 1) stem, pool: the MaxPool3dStem behind it, x: the stem's input"""
    if not (ENABLED and STEM_POOL_FUSED and STEM_SPACE_TO_DEPTH and POOL_CHANNELS_LAST and isinstance(stem, ConvBiasReLU) and isinstance(pool, MaxPool3dStem)):
        return False
    conv = stem[0]
    if not (x.is_cuda and x.dtype == torch.float32 and conv.bias is not None and isinstance(conv, nn.Conv3d) and conv.groups == 1 and _unit(conv.dilation)
            and not isinstance(conv.padding, str) and _is_stem221(conv, x) and stem_forward_supported(x, conv.weight) and torch.is_grad_enabled()
            and not torch.is_autocast_enabled() and _on_current_device(x)):
        return False
    return _t3(pool.kernel_size) == (3, 3, 3) and _t3(pool.stride) == (2, 2, 1) and _t3(pool.padding) == (1, 1, 1) and _t3(pool.dilation) == (1, 1, 1) \
        and not pool.ceil_mode and not pool.return_indices and int(conv.out_channels) > 1


def conv_stem_bias_relu_pool(stem, x):
    conv = stem[0]
    return _ConvStemBiasReLUPool.apply(x, conv.weight, conv.bias)


def _unit(t):
    return all(int(v) == 1 for v in t)


def _is_stem221(conv, x):
    k = conv.kernel_size
    return x.dim() == 5 and tuple(conv.stride) == (2, 2, 1) and k[0] == k[1] == k[2] and k[0] % 2 == 1 and k[0] >= 3 \
        and tuple(conv.padding) == (k[0] // 2,) * 3 and conv.in_channels <= 4 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0


def _conv(conv, x):
    if BWD_DATA_AS_FWD and x.is_cuda and x.dtype == torch.float32 and conv.groups == 1 and _unit(conv.stride) and _unit(conv.dilation) \
            and not isinstance(conv.padding, str) and all(2 * int(p) + 1 == int(k) for p, k in zip(conv.padding, conv.kernel_size)) \
            and torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        return _ConvStride1.apply(x, conv.weight, tuple(int(p) for p in conv.padding))
    if STEM_SPACE_TO_DEPTH and x.is_cuda and x.dtype == torch.float32 and conv.groups == 1 and _unit(conv.dilation) \
            and not isinstance(conv.padding, str) and _is_stem221(conv, x):
        return _ConvStem221.apply(x, conv.weight)
    if S2D_GENERAL and x.is_cuda and x.dtype == torch.float32 and conv.groups == 1 and _unit(conv.dilation) and not isinstance(conv.padding, str) \
            and _is_s221_general(conv, x) and not torch.is_autocast_enabled():
        return _ConvS2D221.apply(x if x.is_contiguous(memory_format=torch.channels_last_3d) else x.contiguous(memory_format=torch.channels_last_3d), conv.weight)
    fn = F.conv3d if isinstance(conv, nn.Conv3d) else F.conv2d
    return fn(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


def conv_unit_stride(x, w, padding):
    """unit-stride, size-preserving convolution of x with an explicit filter tensor (e.g. the concatenation of two layers' filters):
    the same dispatch as `_conv` -- `_ConvStride1` (input gradient as a forward convolution, own weight-gradient kernels) when
    gradients are wanted, a plain MIOpen forward otherwise"""
    nd = w.dim() - 2
    padding = tuple(int(p) for p in (padding if isinstance(padding, (tuple, list)) else (padding,) * nd))
    if BWD_DATA_AS_FWD and x.is_cuda and x.dtype == torch.float32 and all(2 * p + 1 == int(k) for p, k in zip(padding, w.shape[2:])) \
            and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return _ConvStride1.apply(x, w, padding)
    return (F.conv3d if nd == 3 else F.conv2d)(x, w, None, 1, padding)


CONV1X1_FWD = True   # module switch (A/B: bench.py --conv1x1-fwd 0): the bottleneck 1x1 layers of the C2 / C3 stages with their epilogue in one pass (mdt_conv1x1_forward)


def conv1x1_forward_applies(conv, x):
    """the layer is one of mdt_conv1x1_forward's: 1x1(x1), unit stride, bias, channels-last fp32 on this process's GPU, a large map, a supported channel pair"""
    if not (ENABLED and CONV1X1_FWD and x.is_cuda and x.dtype == torch.float32 and conv.bias is not None and conv.groups == 1 and x.dim() in (4, 5)
            and all(int(k) == 1 for k in conv.kernel_size) and _unit(conv.stride) and _unit(conv.dilation) and not isinstance(conv.padding, str)
            and not any(int(p) for p in conv.padding) and not torch.is_autocast_enabled()):
        return False
    mf = torch.channels_last_3d if x.dim() == 5 else torch.channels_last
    if not (x.is_contiguous(memory_format=mf) and not x.is_contiguous() and x.data_ptr() % 16 == 0 and _on_current_device(x)):
        return False
    return x.numel() >= (1 << 20) and bool(_lib.lib().mdt_conv1x1_forward_supported(int(conv.in_channels), int(conv.out_channels)))


def _conv1x1_forward(x, w, bias, res, relu):
    """act(conv1x1(x, w) + bias (+ res)) on channels-last rows, one launch; None when the kernel declines (alignment)"""
    cout, cin = int(w.shape[0]), int(w.shape[1])
    out = torch.empty((x.shape[0], cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device,
                      memory_format=torch.channels_last_3d if x.dim() == 5 else torch.channels_last)
    wd = w.detach()
    rc = _lib.lib().mdt_conv1x1_forward(x.data_ptr(), wd.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None, out.data_ptr(),
                                        x.numel() // cin, cin, cout, 1 if relu else 0, _lib.raw_stream())
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        return None
    if rc != 0:
        _lib.check(rc, "mdt_conv1x1_forward")
    return out


CONV1X1_BWD = True   # module switch (A/B: bench.py --conv1x1-bwd 0): ReLU mask + bias gradient + input gradient of conv3 (C2 blocks) in one pass (mdt_conv1x1_backward)


def _conv1x1_backward(gy, y, w, mf):
    """(g, gx, gbias) of act(conv1x1(x, w) + bias (+ res)) for the output gradient gy (y: the saved output when the activation is a ReLU, else None), one pass;
    None when the layer is not mdt_conv1x1_backward's (shape, layout, alignment)"""
    cout, cin = int(w.shape[0]), int(w.shape[1])
    L = _lib.lib()
    if not (L.mdt_conv1x1_backward_supported(cin, cout) and gy.is_cuda and gy.dtype == torch.float32 and gy.is_contiguous(memory_format=mf) and not gy.is_contiguous()
            and gy.data_ptr() % 16 == 0 and (y is None or y.data_ptr() % 16 == 0) and _on_current_device(gy)):
        return None
    V = gy.numel() // cout
    g = torch.empty_like(gy) if y is not None else None
    gx = torch.empty((gy.shape[0], cin) + tuple(gy.shape[2:]), dtype=torch.float32, device=gy.device, memory_format=mf)
    gbias = torch.empty(cout, dtype=torch.float32, device=gy.device)
    ws = _workspace(L.mdt_conv1x1_backward_workspace_bytes(V, cout), gy.device)
    rc = L.mdt_conv1x1_backward(gy.data_ptr(), y.data_ptr() if y is not None else None, w.detach().data_ptr(), g.data_ptr() if g is not None else None, gx.data_ptr(),
                                gbias.data_ptr(), V, cin, cout, ws.data_ptr(), ws.numel(), _lib.raw_stream())
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        return None
    if rc != 0:
        _lib.check(rc, "mdt_conv1x1_backward")
    return (g if g is not None else gy), gx, gbias


class _Conv1x1BiasAct(Function):
    """y = act(conv1x1(x) + bias (+ residual)) in ONE pass over the operands (csrc/conv1x1_fwd.hip) -- conv3 + residual + ReLU of a ResBlock
    (models/backbone.py:203-205).  Backward: the ReLU mask and the bias gradient as in _BiasAct, input / weight gradients as in _ConvStride1."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, relu):
        mf = torch.channels_last_3d if x.dim() == 5 else torch.channels_last
        if residual is not None and not residual.is_contiguous(memory_format=mf):
            residual = residual.contiguous(memory_format=mf)
        y = _conv1x1_forward(x, w, bias, residual, relu)
        if y is None:
            y = (F.conv3d if x.dim() == 5 else F.conv2d)(x, w)
            rc = _lib.lib().mdt_bias_act_forward(y.data_ptr(), y.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
                                                 y.numel(), y.shape[1], 1, 1 if relu else 0, _lib.raw_stream())
            if rc != 0:
                _lib.check(rc, "mdt_bias_act_forward")
        ctx.relu, ctx.mf, ctx.has_res = relu, mf, residual is not None
        if relu:
            ctx.save_for_backward(x, w, y)
        else:
            ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors[0], ctx.saved_tensors[1]
        y = ctx.saved_tensors[2] if ctx.relu else None
        nd = w.dim() - 2
        fused = _conv1x1_backward(gy, y, w, ctx.mf) if (CONV1X1_BWD and ctx.needs_input_grad[0]) else None
        if fused is not None:               # ReLU mask, bias gradient and input gradient from ONE pass over gy (mdt_conv1x1_backward)
            g, gx, gbias = fused
            gw = _stride1_grads(x, w, (0,) * nd, g, False, True)[1] if ctx.needs_input_grad[1] else None
            return gx, gw, (gbias if ctx.needs_input_grad[2] else None), (g if ctx.has_res and ctx.needs_input_grad[3] else None), None
        if not ctx.relu and not gy.is_contiguous(memory_format=ctx.mf):
            r = bias_grad_to_channels_last(gy, ctx.mf)
            g, gbias = r if r is not None else _bias_act_bwd(gy, None, False, ctx.mf)
        else:
            g, gbias = _bias_act_bwd(gy, y, ctx.relu, ctx.mf)
        gx, gw = _stride1_grads(x, w, (0,) * nd, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, (gbias if ctx.needs_input_grad[2] else None), (g if ctx.has_res and ctx.needs_input_grad[3] else None), None


RES_TAP = True   # module switch (A/B: bench.py --res-tap 0)


def conv1x1_dgrad_add(gy, w, res):
    """res + (input gradient of the 1x1(x1) convolution with weight w for the output gradient gy) in ONE pass over channels-last fp32 rows
    (csrc/epilogue.hip conv1x1_dgrad_add_kernel).  None when the shapes / layouts are not the kernel's (the caller then uses torch ops)."""
    nd = gy.dim() - 2
    mf = torch.channels_last_3d if nd == 3 else torch.channels_last if nd == 2 else None
    if mf is None or not (gy.is_cuda and gy.dtype == torch.float32 and res.dtype == torch.float32 and _on_current_device(gy)):
        return None
    cout, cin = int(w.shape[0]), int(w.shape[1])
    L = _lib.lib()
    if not L.mdt_conv1x1_dgrad_add_supported(cout, cin) or not gy.is_contiguous(memory_format=mf) or not res.is_contiguous(memory_format=mf):
        return None
    out = torch.empty_like(res)
    wc = w.detach().reshape(cout, cin).contiguous()
    rc = L.mdt_conv1x1_dgrad_add(gy.data_ptr(), wc.data_ptr(), res.data_ptr(), out.data_ptr(), gy.numel() // cout, cout, cin, _lib.raw_stream())
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        return None
    if rc != 0:
        _lib.check(rc, "mdt_conv1x1_dgrad_add")
    return out


class _Conv1x1ResTap(Function):
    """h = conv1x1(x, w) AND an alias of x for the residual path of a ResBlock (models/backbone.py:197-205: x feeds conv1 and the residual
    add).  As two consumers of x, autograd produces conv1's input gradient and the residual gradient as two 302 MB tensors (C2 maps) and
    adds them in a third pass; owning both paths, this node's backward computes  gx = g_residual + W^T g_h  in one pass
    (mdt_conv1x1_dgrad_add).  The alias is returned as-is (autograd wraps it as a view whose gradient arrives here)."""

    @staticmethod
    def forward(ctx, x, w, bias=None):
        nd = w.dim() - 2
        ctx.fused = False
        if bias is not None:         # round 6: conv1 + bias + ReLU in one pass (mdt_conv1x1_forward); the backward masks gh with the saved output
            h = _conv1x1_forward(x, w, bias, None, True)
            if h is None:
                h = (F.conv3d if nd == 3 else F.conv2d)(x, w)
                rc = _lib.lib().mdt_bias_act_forward(h.data_ptr(), h.data_ptr(), bias.data_ptr(), None, h.numel(), h.shape[1], 1, 1, _lib.raw_stream())
                if rc != 0:
                    _lib.check(rc, "mdt_bias_act_forward")
            ctx.fused = True
            ctx.save_for_backward(x, w, h)
            return h, x
        ctx.save_for_backward(x, w)
        return (F.conv3d if nd == 3 else F.conv2d)(x, w), x

    @staticmethod
    def backward(ctx, gh, gres):
        x, w = ctx.saved_tensors[0], ctx.saved_tensors[1]
        nd = w.dim() - 2
        gbias = None
        if ctx.fused and gh is not None:
            gh, gbias = _bias_act_bwd(gh, ctx.saved_tensors[2], True, torch.channels_last_3d if nd == 3 else torch.channels_last)
        gx = None
        if ctx.needs_input_grad[0]:
            if gres is not None and gh is not None:
                gx = conv1x1_dgrad_add(gh, w, gres)
            if gx is None:
                if gh is not None:
                    gx = _stride1_grads(x, w, (0,) * nd, gh, True, False)[0]
                    if gres is not None:
                        gx = gx + gres
                else:
                    gx = gres
        gw = _stride1_grads(x, w, (0,) * nd, gh, False, True)[1] if (ctx.needs_input_grad[1] and gh is not None) else None
        if ctx.fused:
            return gx, gw, gbias
        return gx, gw


def res_tap_applies(seq, x):
    """ConvBiasReLU 1x1(x1) unit-stride layer on a large channels-last fp32 activation that needs a gradient: the ResBlock input tap"""
    if not (ENABLED and BWD_DATA_AS_FWD and RES_TAP and isinstance(seq, ConvBiasReLU)):
        return False
    conv = seq[0]
    if not (conv.bias is not None and conv.groups == 1 and all(int(k) == 1 for k in conv.kernel_size) and _unit(conv.stride) and _unit(conv.dilation)
            and not isinstance(conv.padding, str) and not any(int(p) for p in conv.padding)):
        return False
    mf = torch.channels_last_3d if x.dim() == 5 else torch.channels_last if x.dim() == 4 else None
    if mf is None or not (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and x.requires_grad and not torch.is_autocast_enabled()
                          and x.is_contiguous(memory_format=mf) and not x.is_contiguous() and _on_current_device(x)):
        return False
    return x.numel() >= (1 << 22) and bool(_lib.lib().mdt_conv1x1_dgrad_add_supported(int(conv.out_channels), int(conv.in_channels)))


def conv_bias_relu_with_res_tap(seq, x):
    """(relu(conv1(x) + bias), alias of x for the residual add) through _Conv1x1ResTap"""
    conv = seq[0]
    if conv1x1_forward_applies(conv, x):
        return _Conv1x1ResTap.apply(x, conv.weight, conv.bias)
    h, x_res = _Conv1x1ResTap.apply(x, conv.weight)
    return bias_act(h, conv.bias, None, True), x_res


STRIDE_TAP = True   # module switch (A/B: bench.py --stride-tap 0)
UPSAMPLE_NEAREST_CL = True   # module switch (A/B: bench.py --upsample-nearest-cl 0)


class _UpsampleNearestCL(Function):
    """Nearest-neighbour up-sampling by an integer factor per axis ON channels-last storage (the FPN's top-down path,
    models/backbone.py:147-153: F.interpolate(p, scale_factor=2)).  torch's nearest kernels for 5-D tensors are row-major only: a
    channels-last input is converted, the row-major result is converted back in front of the lateral's fused add (151 MB copies on P2 at
    8 x 128^3), and the backward makes the same two conversions.  In channels-last memory the operation is a broadcast of whole C-rows:
    forward = ONE expanding copy, backward = ONE strided sum over the s_y x s_x x s_z replicas; no layout change either way."""

    @staticmethod
    def forward(ctx, x, scale):
        nd = x.dim() - 2
        ctx.scale = scale
        ctx.in_shape = tuple(x.shape)
        perm = (0,) + tuple(range(2, 2 + nd)) + (1,)
        xv = x.permute(*perm)                                        # [B, *spatial, C], contiguous for channels-last x
        B, C = x.shape[0], x.shape[1]
        idx, exp, outsp = [slice(None)], [B], []
        for d in range(nd):
            idx += [slice(None), None]
            exp += [x.shape[2 + d], scale[d]]
            outsp.append(x.shape[2 + d] * scale[d])
        idx.append(slice(None))
        exp.append(C)
        out = xv[tuple(idx)].expand(*exp).reshape(B, *outsp, C)       # the one copy
        inv = (0, nd + 1) + tuple(range(1, nd + 1))
        return out.permute(*inv)

    @staticmethod
    def backward(ctx, g):
        nd = len(ctx.in_shape) - 2
        mf = torch.channels_last_3d if nd == 3 else torch.channels_last
        if not g.is_contiguous(memory_format=mf):
            g = g.contiguous(memory_format=mf)
        perm = (0,) + tuple(range(2, 2 + nd)) + (1,)
        B, C = ctx.in_shape[0], ctx.in_shape[1]
        shp, red = [B], []
        for d in range(nd):
            shp += [ctx.in_shape[2 + d], ctx.scale[d]]
            red.append(2 + 2 * d)
        gs = g.permute(*perm).reshape(*shp, C).sum(tuple(red))        # [B, *spatial_in, C]
        inv = (0, nd + 1) + tuple(range(1, nd + 1))
        return gs.permute(*inv), None


def upsample_nearest(x, scale_factor):
    """F.interpolate(x, scale_factor=s) (mode 'nearest') -- on channels-last storage without layout changes where that applies"""
    nd = x.dim() - 2
    sc = tuple(scale_factor) if isinstance(scale_factor, (tuple, list)) else (scale_factor,) * nd
    mf = torch.channels_last_3d if nd == 3 else torch.channels_last if nd == 2 else None
    if ENABLED and UPSAMPLE_NEAREST_CL and mf is not None and all(float(v) == int(v) and int(v) >= 1 for v in sc) and x.is_contiguous(memory_format=mf) \
            and not x.is_contiguous() and not torch.is_autocast_enabled():
        return _UpsampleNearestCL.apply(x, tuple(int(v) for v in sc))
    return F.interpolate(x, scale_factor=scale_factor)


class _StrideTap(Function):
    """x -> (alias of x for the FPN lateral, x_s = x[..., ::s, ::s, ::s] for the strided 1x1 layers of the next stage's first ResBlock).

    A stage output c_k has three consumers (models/backbone.py:128-153): conv1 and downsample of the next stage's first block -- both 1x1
    convolutions with stride 2, i.e. unit-stride layers on the SAME sub-sampled tensor -- and the lateral P_k_conv1.  As three autograd
    consumers, each strided layer's input gradient is a full-size tensor that is zero at 7 of 8 voxels (2 x 302 MB on c2 at 8 x 128^3,
    written by MIOpen's backward-data kernels) and two more full-size passes add them to the lateral's gradient.  Sub-sampled ONCE here,
    the two layers run at unit stride (their gradients meet at 1/8 size), and this node's backward adds that sum into the lateral's
    gradient at the even voxels, in place: one strided pass over 1/8 of the rows."""

    @staticmethod
    def forward(ctx, x, stride):
        ctx.set_materialize_grads(False)
        ctx.stride = tuple(int(s) for s in stride)
        ctx.mf = torch.channels_last_3d if x.dim() == 5 else torch.channels_last
        ctx.x_shape = tuple(x.shape)
        sl = (slice(None), slice(None)) + tuple(slice(None, None, s) for s in ctx.stride)
        return x, x[sl].contiguous(memory_format=ctx.mf)

    @staticmethod
    def backward(ctx, g_lat, g_s):
        sl = (slice(None), slice(None)) + tuple(slice(None, None, s) for s in ctx.stride)
        if g_lat is None:
            if g_s is None:
                return None, None
            g = torch.zeros(ctx.x_shape, dtype=g_s.dtype, device=g_s.device).contiguous(memory_format=ctx.mf)
            g[sl].copy_(g_s)
            return g, None
        if g_s is None:
            return g_lat, None
        # g_lat is the lateral convolution's input gradient: a fresh dense tensor this node is the only consumer of (the alias has ONE
        # user by construction, backbone.FPN._stage).  Autograd does not promise that: a tensor hook on the alias that KEEPS the gradient
        # it is shown, or a producer that hands one tensor to two inputs, shares it -- the in-place add would corrupt the other holder's
        # copy.  So the add runs in place only when ownership is provable: dense, not a view, and no more Python references
        # than the engine itself holds when it calls this function (calibrated once per process by `_tap_owned_refs`); otherwise on a clone.
        refs = sys.getrefcount(g_lat)
        if _TAP_REFS[0] == -1:           # calibration call
            _TAP_REFS[0] = refs
        owned = refs <= _tap_owned_refs() and g_lat._base is None and (g_lat.is_contiguous(memory_format=ctx.mf) or g_lat.is_contiguous())
        g = g_lat if owned else g_lat.clone(memory_format=ctx.mf)
        _TAP_REFS[1 if owned else 2] += 1
        g[sl].add_(g_s)
        return g, None


_TAP_REFS = [None, 0, 0]     # [references the engine holds on a gradient it passes to _StrideTap.backward, in-place adds, cloned adds]


def _tap_owned_refs():
    """sys.getrefcount of a gradient nobody else holds, at the line of _StrideTap.backward that asks: measured, not assumed (it depends on the
    interpreter and on how torch's engine wraps gradients for Python functions) -- one tiny CPU backward the first time it is needed"""
    if _TAP_REFS[0] is None:
        _TAP_REFS[0] = -1
        with torch.enable_grad():
            x = torch.zeros((1, 1, 2, 2, 2), requires_grad=True)
            lat, xs = _StrideTap.apply(x * 1.0, (2, 2, 2))
            (F.conv3d(lat, torch.ones((1, 1, 1, 1, 1))).sum() + xs.sum()).backward()
        if _TAP_REFS[0] == -1:          # the calibration backward did not come by (never expected): nothing is provably owned
            _TAP_REFS[0] = 0
        _TAP_REFS[1] = _TAP_REFS[2] = 0
    return _TAP_REFS[0] if _TAP_REFS[0] != -1 else (1 << 30)


def stride_tap_applies(block, x):
    """first ResBlock of a stage: 1x1 conv1 (+bias+ReLU) and 1x1 downsample (+bias) with the SAME stride > 1, fused-epilogue modules, fp32 CUDA
    activation that needs a gradient"""
    if not (ENABLED and STRIDE_TAP and BWD_DATA_AS_FWD) or block.downsample is None or not isinstance(block.conv1, ConvBiasReLU) \
            or not isinstance(block.downsample, ConvBias):
        return False
    c1, ds = block.conv1[0], block.downsample

    def one_by_one(c):
        return c.bias is not None and c.groups == 1 and all(int(k) == 1 for k in c.kernel_size) and _unit(c.dilation) \
            and not isinstance(c.padding, str) and not any(int(p) for p in c.padding)
    if not (one_by_one(c1) and one_by_one(ds) and tuple(c1.stride) == tuple(ds.stride) and any(int(v) > 1 for v in c1.stride)):
        return False
    return x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and x.requires_grad and not torch.is_autocast_enabled() \
        and x.dim() in (4, 5) and _on_current_device(x)


def stride_tap(x, stride):
    """(alias of x for the lateral, x sub-sampled by `stride`)"""
    _tap_owned_refs()          # calibrated here, in forward context, not inside the first backward
    return _StrideTap.apply(x, tuple(int(s) for s in stride))


def conv1x1_unit_stride_bias_act(conv, x_s, residual=None, relu=False):
    """a 1x1 convolution module applied at UNIT stride to an already sub-sampled input (same parameters, same result as the strided
    layer on the full tensor) + its bias / ReLU epilogue"""
    return bias_act(conv_unit_stride(x_s, conv.weight, 0), conv.bias, residual, relu)


class ConvBias(object):
    """mixin for the bare-conv form (relu=None in the reference's generator): forward(x, residual=None, relu=False)"""

    def forward(self, x, residual=None, relu=False):
        if conv1x1_forward_applies(self, x) and (residual is None or (residual.dtype == torch.float32 and residual.is_cuda
                and tuple(residual.shape) == (x.shape[0], self.out_channels) + tuple(x.shape[2:]))):
            return _Conv1x1BiasAct.apply(x, self.weight, self.bias, residual, relu)
        return bias_act(_conv(self, x), self.bias, residual, relu)


class ConvBias2d(ConvBias, nn.Conv2d):
    pass


class ConvBias3d(ConvBias, nn.Conv3d):
    pass


class ConvBiasReLU(nn.Sequential):
    """Sequential(conv, ReLU) of the reference's generator with the fused epilogue; keys '0.weight' / '0.bias'"""

    def forward(self, x, residual=None):
        conv = self[0]
        if residual is None and ENABLED and STEM_SPACE_TO_DEPTH and conv.bias is not None and isinstance(conv, nn.Conv3d) and conv.groups == 1 \
                and _unit(conv.dilation) and not isinstance(conv.padding, str) and _is_stem221(conv, x) and stem_forward_supported(x, conv.weight):
            return _ConvStemBiasReLU.apply(x, conv.weight, conv.bias)
        if residual is None and _conv3_small_fused_applies(conv, x):
            return _Conv3SmallBiasReLU.apply(x, conv.weight, conv.bias)
        return bias_act(_conv(conv, x), conv.bias, residual, True)


class _MaxPoolK3S221(Function):
    """MaxPool3d(3, stride (2, 2, 1), padding 1) on channels_last_3d storage (csrc/pool.hip); arg-max taps as uint8"""

    @staticmethod
    def forward(ctx, x):
        B, C, Y, X, Z = x.shape
        OY, OX = (Y - 1) // 2 + 1, (X - 1) // 2 + 1
        y = torch.empty((B, C, OY, OX, Z), dtype=torch.float32, device=x.device, memory_format=torch.channels_last_3d)
        arg = torch.empty((B, OY, OX, Z, C), dtype=torch.uint8, device=x.device)
        rc = _lib.lib().mdt_maxpool3d_k3s221_cl_forward(x.data_ptr(), y.data_ptr(), arg.data_ptr(), B, Y, X, Z, C,
                                                        _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_maxpool3d_k3s221_cl_forward")
        ctx.save_for_backward(arg)
        ctx.in_shape = (B, C, Y, X, Z)
        return y

    @staticmethod
    def backward(ctx, gy):
        arg, = ctx.saved_tensors
        B, C, Y, X, Z = ctx.in_shape
        if not gy.is_contiguous(memory_format=torch.channels_last_3d):
            gy = gy.contiguous(memory_format=torch.channels_last_3d)
        gx = torch.empty((B, C, Y, X, Z), dtype=torch.float32, device=gy.device, memory_format=torch.channels_last_3d)
        rc = _lib.lib().mdt_maxpool3d_k3s221_cl_backward(gy.data_ptr(), arg.data_ptr(), gx.data_ptr(), B, Y, X, Z, C,
                                                         _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_maxpool3d_k3s221_cl_backward")
        return gx


POOL_CHANNELS_LAST = True   # module switch (A/B: bench.py --pool-cl 0)


class MaxPool3dStem(nn.MaxPool3d):
    """nn.MaxPool3d that runs the channels-last kernel when it is the stem pooling (kernel 3, stride (2, 2, 1), padding 1,
    models/backbone.py:77-79) on an fp32 channels_last_3d GPU activation with C > 1; torch's kernel otherwise (torch has no
    channels-last 3D max pooling: it copies to NCDHW and the next convolution transposes back)."""

    def forward(self, x):
        if POOL_CHANNELS_LAST and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] > 1 \
                and _t3(self.kernel_size) == (3, 3, 3) and _t3(self.stride) == (2, 2, 1) and _t3(self.padding) == (1, 1, 1) \
                and _t3(self.dilation) == (1, 1, 1) and not self.ceil_mode and not self.return_indices \
                and x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous() and _on_current_device(x):
            return _MaxPoolK3S221.apply(x)
        return super(MaxPool3dStem, self).forward(x)


def _t3(v):
    return tuple(int(a) for a in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


UPSAMPLE_CL = True   # module switch (A/B: bench.py --upsample-cl 0)


class _Upsample2xYX(Function):
    """x2 linear up-sampling of (y, x) on channels-last storage (csrc/upsample.hip)"""

    @staticmethod
    def forward(ctx, x):
        nd = x.dim() - 2
        B, C, Y, X = int(x.shape[0]), int(x.shape[1]), int(x.shape[2]), int(x.shape[3])
        Z = int(x.shape[4]) if nd == 3 else 1
        mf = torch.channels_last_3d if nd == 3 else torch.channels_last
        shape = (B, C, 2 * Y, 2 * X) + ((Z,) if nd == 3 else ())
        y = torch.empty(shape, dtype=torch.float32, device=x.device, memory_format=mf)
        rc = _lib.lib().mdt_upsample2x_yx_cl_forward(x.data_ptr(), y.data_ptr(), B, Y, X, Z * C, _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_upsample2x_yx_cl_forward")
        ctx.dims = (B, C, Y, X, Z, nd)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, C, Y, X, Z, nd = ctx.dims
        mf = torch.channels_last_3d if nd == 3 else torch.channels_last
        if not gy.is_contiguous(memory_format=mf):
            gy = gy.contiguous(memory_format=mf)
        gx = torch.empty((B, C, Y, X) + ((Z,) if nd == 3 else ()), dtype=torch.float32, device=gy.device, memory_format=mf)
        rc = _lib.lib().mdt_upsample2x_yx_cl_backward(gy.data_ptr(), gx.data_ptr(), B, Y, X, Z * C, _lib.raw_stream())
        if rc != 0:
            _lib.check(rc, "mdt_upsample2x_yx_cl_backward")
        return gx


def upsample2x_yx(x, scale_factor, mode):
    """F.interpolate(x, scale_factor, mode, align_corners=False) through the channels-last kernel when it is the decoder's
    form -- 'trilinear' with scale (2, 2, 1) on a channels_last_3d fp32 GPU tensor, or 'bilinear' with scale 2 on a
    channels_last one (C > 1) -- else None (the caller uses torch)."""
    if not (UPSAMPLE_CL and x.is_cuda and x.dtype == torch.float32 and x.shape[1] > 1 and _on_current_device(x)):
        return None
    sf = tuple(float(v) for v in scale_factor) if isinstance(scale_factor, (tuple, list)) else (float(scale_factor),) * (x.dim() - 2)
    if x.dim() == 5 and mode == "trilinear" and sf == (2.0, 2.0, 1.0) and x.is_contiguous(memory_format=torch.channels_last_3d) and not x.is_contiguous():
        return _Upsample2xYX.apply(x)
    if x.dim() == 4 and mode == "bilinear" and sf == (2.0, 2.0) and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        return _Upsample2xYX.apply(x)
    return None
