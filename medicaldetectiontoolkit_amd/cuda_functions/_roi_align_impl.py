"""Shared implementation behind cuda_functions.roi_align_{2D,3D}.roi_align.crop_and_resize.

The reference uses legacy instance-style autograd Functions (crop_and_resize.py:10-51),
which torch >= 1.3 rejects; here `CropAndResizeFunction(ch, cw[, cd], extrap)` stays a
callable object with the same constructor / call signature and dispatches to a
new-style static Function.  Gradient flows to `image` only (crop_and_resize.py:51).
"""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib


# bench.py hook: when PROFILE is a list, every backward call appends (start_event, end_event, meta); the events are
# recorded on the stream the kernels are launched on (torch's current stream).  None = off (no overhead).
PROFILE = None


def _prep(image, boxes, box_ind, dim):
    _lib.require_cuda(image, "image")
    # quirk 6: mask targets arrive as a (n,1,Y,X,Z,1) view -- tolerate trailing singletons
    while image.dim() > dim + 2 and image.size(-1) == 1:
        image = image.squeeze(-1)
    if image.dim() != dim + 2:
        raise ValueError("image must be [B, C, %s], got %s" % ("Y, X, Z" if dim == 3 else "Y, X", tuple(image.shape)))
    if boxes.dim() != 2 or boxes.size(1) != 2 * dim:
        raise ValueError("boxes must be [N, %d], got %s" % (2 * dim, tuple(boxes.shape)))
    if box_ind.dim() != 1 or box_ind.size(0) != boxes.size(0):
        raise ValueError("box_ind must be [N]")
    # the reference's C glue never checks dtype / contiguity (crop_and_resize_gpu.c); do it here
    image = image.contiguous()
    # bf16 feature maps (autocast, BASELINE config 5) ALWAYS go to the bf16-input kernel as they are: it widens each bf16
    # value exactly to fp32 and interpolates in fp32 (bit-equal to the fp32 kernel on the widened map), in training too --
    # the backward produces an fp32 gradient map that is cast back to the map's dtype.  Other dtypes are converted to fp32.
    # uint8 images (the GT masks of a batch) are read as uint8 by their own kernel instantiation when no gradient is wanted.
    if image.dtype not in (torch.float32, torch.bfloat16) and not (image.dtype == torch.uint8 and not image.requires_grad):
        image = image.float()
    boxes = boxes.detach().to(device=image.device, dtype=torch.float32).contiguous()
    box_ind = box_ind.detach().to(device=image.device, dtype=torch.int32).contiguous()
    return image, boxes, box_ind


def crop_forward(image, boxes, box_ind, crop, extrapolation_value=0.0):
    dim = len(crop)
    L = _lib.lib()
    n = boxes.size(0)
    B, C = image.size(0), image.size(1)
    crops = torch.empty((n, C) + tuple(crop), dtype=torch.float32, device=image.device)
    if n == 0 or C == 0:
        return crops
    with torch.cuda.device(image.device):
        s = _lib.current_stream_ptr()
        if image.dtype == torch.uint8:
            if dim == 3:
                rc = L.mdt_crop_and_resize_3d_forward_u8(
                    _lib.ptr(image), _lib.ptr(boxes), _lib.ptr(box_ind), n, B, image.size(2), image.size(3), image.size(4),
                    crop[0], crop[1], crop[2], C, _lib.ptr(crops), s)
            else:
                rc = L.mdt_crop_and_resize_2d_forward_u8(
                    _lib.ptr(image), _lib.ptr(boxes), _lib.ptr(box_ind), n, B, image.size(2), image.size(3),
                    crop[0], crop[1], C, _lib.ptr(crops), s)
        elif image.dtype == torch.bfloat16:
            if dim == 3:
                rc = L.mdt_crop_and_resize_3d_forward_bf16(
                    _lib.ptr(image), _lib.ptr(boxes), _lib.ptr(box_ind), n, B, image.size(2), image.size(3), image.size(4),
                    crop[0], crop[1], crop[2], C, _lib.ptr(crops), s)
            else:
                rc = L.mdt_crop_and_resize_2d_forward_bf16(
                    _lib.ptr(image), _lib.ptr(boxes), _lib.ptr(box_ind), n, B, image.size(2), image.size(3),
                    crop[0], crop[1], C, _lib.ptr(crops), s)
        elif dim == 3:
            rc = L.mdt_crop_and_resize_3d_forward(
                _lib.ptr(image), _lib.ptr(boxes), _lib.ptr(box_ind), n, B, image.size(2), image.size(3), image.size(4),
                crop[0], crop[1], crop[2], C, ctypes.c_float(extrapolation_value), _lib.ptr(crops), s)
        else:
            rc = L.mdt_crop_and_resize_2d_forward(
                _lib.ptr(image), _lib.ptr(boxes), _lib.ptr(box_ind), n, B, image.size(2), image.size(3),
                crop[0], crop[1], C, ctypes.c_float(extrapolation_value), _lib.ptr(crops), s)
    _lib.check(rc, "mdt_crop_and_resize_%dd_forward" % dim)
    return crops


_SMALL_WS = {}


def _workspace(nbytes, device):
    """the default kernel needs no workspace (the query answers 256 bytes): reuse one tiny buffer per device"""
    if nbytes <= 256 and not _lib.CAPTURING:
        ws = _SMALL_WS.get(device)
        if ws is None:
            ws = _SMALL_WS[device] = torch.empty(256, dtype=torch.uint8, device=device)
        return ws
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def crop_backward(grads, boxes, box_ind, im_size, mode="fast", atomic=False, out=None):
    """grads [N, C, *crop] -> grad_image of shape im_size; fully written by the kernel(s) (`out`: write into this
    contiguous fp32 tensor instead of a fresh one -- cache-cold timing rotates several).
    mode: "fast" (default: the single-launch gather-form kernel, csrc/roi_align_bwd_v3.hip; beyond its budgets -- more than 128 RoIs --
    the library runs the exact-order kernel), "ordered" (bit-exact vs the sequential oracle).
    A/B history, served by libmdt_hip_ab.so and asked for by tests / tools only: "twophase" (round-1 separable two-kernel form),
    "territory" (round-2 single-launch kernel), "atomic" (the reference's algorithm)."""
    if atomic:
        mode = "atomic"
    dim = len(im_size) - 2
    L = _lib.lib()
    grads = grads.contiguous()
    if grads.dtype != torch.float32:
        grads = grads.float()
    n = grads.size(0)
    if out is not None:
        if tuple(out.shape) != tuple(im_size) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous fp32 tensor of shape im_size")
        grad_image = out
    else:
        grad_image = torch.empty(tuple(im_size), dtype=torch.float32, device=grads.device)
    if grad_image.numel() == 0:
        return grad_image
    crop = tuple(grads.shape[2:])
    prof = PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_valid = ((box_ind >= 0) & (box_ind < im_size[0])).sum()      # device scalar, read after the timed region
        ev0.record()
    with torch.cuda.device(grads.device):
        s = _lib.current_stream_ptr()
        head = [_lib.ptr(grads), _lib.ptr(boxes), _lib.ptr(box_ind), n, im_size[0]] + list(im_size[2:]) + \
            list(crop) + [im_size[1], _lib.ptr(grad_image)]
        if mode == "fast":
            fn = L.mdt_crop_and_resize_3d_backward if dim == 3 else L.mdt_crop_and_resize_2d_backward
            rc = fn(*(head + [None, 0, s]))
        elif mode == "twophase":
            A = _lib.ab_lib()
            wsb = A.mdt_crop_and_resize_backward_twophase_workspace_bytes(dim, n, im_size[1], im_size[2], im_size[3], im_size[4] if dim == 3 else 1,
                                                                          crop[0], crop[1], crop[2] if dim == 3 else 1)
            ws = _workspace(wsb, grads.device)
            fn = A.mdt_crop_and_resize_3d_backward_twophase if dim == 3 else A.mdt_crop_and_resize_2d_backward_twophase
            rc = fn(*(head + [_lib.ptr(ws), wsb, s]))
        elif mode == "territory":
            rc = _lib.ab_lib().mdt_ab_crop_and_resize_backward_territory(
                dim, _lib.ptr(grads), _lib.ptr(boxes), _lib.ptr(box_ind), n, im_size[0], im_size[2], im_size[3], im_size[4] if dim == 3 else 1,
                crop[0], crop[1], crop[2] if dim == 3 else 1, im_size[1], _lib.ptr(grad_image), s)
        elif mode == "ordered":
            fn = L.mdt_crop_and_resize_3d_backward_ordered if dim == 3 else L.mdt_crop_and_resize_2d_backward_ordered
            rc = fn(*(head + [s]))
        elif mode == "atomic" and dim == 3:
            rc = _lib.ab_lib().mdt_crop_and_resize_3d_backward_atomic(*(head + [s]))
        else:
            raise ValueError("unknown backward mode %r for dim %d" % (mode, dim))
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, {"im_size": tuple(im_size), "crop": crop, "n_rows": n, "n_valid": n_valid, "mode": mode}))
    _lib.check(rc, "mdt_crop_and_resize_%dd_backward(%s)" % (dim, mode))
    return grad_image


class _CropAndResize(Function):
    @staticmethod
    def forward(ctx, image, boxes, box_ind, crop, extrapolation_value):
        orig_shape = image.shape
        image_c, boxes_c, box_ind_c = _prep(image, boxes, box_ind, len(crop))
        crops = crop_forward(image_c, boxes_c, box_ind_c, crop, extrapolation_value)
        ctx.im_size = tuple(image_c.shape)
        ctx.orig_shape = tuple(orig_shape)
        ctx.in_dtype = image.dtype
        ctx.save_for_backward(boxes_c, box_ind_c)
        return crops

    @staticmethod
    def backward(ctx, grad_outputs):
        boxes, box_ind = ctx.saved_tensors
        grad_image = crop_backward(grad_outputs, boxes, box_ind, ctx.im_size)
        grad_image = grad_image.reshape(ctx.orig_shape)
        if grad_image.dtype != ctx.in_dtype:
            grad_image = grad_image.to(ctx.in_dtype)
        return grad_image, None, None, None, None


class CropAndResizeFunctionBase(object):
    """Callable with the reference's constructor / call convention:
    CropAndResizeFunction(ch, cw[, cd], extrapolation_value=0)(image, boxes, box_ind)."""
    _dim = None

    def __init__(self, *crop_and_extrap, **kw):
        args = list(crop_and_extrap)
        extrap = kw.pop("extrapolation_value", None)
        if kw:
            raise TypeError("unexpected arguments %s" % sorted(kw))
        if len(args) == self._dim + 1:
            extrap = args.pop()
        if len(args) != self._dim:
            raise TypeError("expected %d crop extents (+ optional extrapolation_value)" % self._dim)
        self.crop = tuple(int(a) for a in args)
        self.crop_height, self.crop_width = self.crop[0], self.crop[1]
        if self._dim == 3:
            self.crop_zdepth = self.crop[2]
        self.extrapolation_value = 0.0 if extrap is None else float(extrap)

    def __call__(self, image, boxes, box_ind):
        return _CropAndResize.apply(image, boxes, box_ind, self.crop, self.extrapolation_value)

    # legacy spelling used by some callers of the reference
    def forward(self, image, boxes, box_ind):
        return self(image, boxes, box_ind)


# ---------------------------------------------------------------------------------------------- all levels, one launch
def _int_array(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[_lib.ptr(t) for t in tensors])


def _level_dims(shapes, dim):
    H = _int_array([s[2] for s in shapes])
    W = _int_array([s[3] for s in shapes])
    D = _int_array([s[4] if dim == 3 else 1 for s in shapes])
    return H, W, D


ROI_ALIGN_CHANNELS_LAST = True     # module switch (A/B): channels-last pyramid maps are pooled as they are (mdt_pyramid_roi_align_forward_cl)


def channels_last_eligible(maps, dim):
    """the maps can go to the channels-last forward as they are: 3D, every map dense in channels_last_3d storage (and not also row-major),
    C % 4 == 0, all fp32 or all bf16"""
    if not (ROI_ALIGN_CHANNELS_LAST and dim == 3 and len(maps) > 0):
        return False
    C = maps[0].size(1)
    if C % 4 != 0 or C < 4:
        return False
    dt = maps[0].dtype
    if dt not in (torch.float32, torch.bfloat16):
        return False
    return all(m.dim() == 5 and m.dtype == dt and m.size(1) == C and m.is_contiguous(memory_format=torch.channels_last_3d) and not m.is_contiguous()
               for m in maps)


def pyramid_forward(maps, boxes, batch_ix, level, crop, channels_last=False):
    """maps: list of [B, C, *spatial_l] (all fp32 or all bf16; contiguous, or -- channels_last=True -- dense channels_last_3d); one launch for
    all levels"""
    dim = len(crop)
    L = _lib.lib()
    n, B, C = boxes.size(0), maps[0].size(0), maps[0].size(1)
    crops = torch.empty((n, C) + tuple(crop), dtype=torch.float32, device=maps[0].device)
    if n == 0 or C == 0:
        return crops
    H, W, D = _level_dims([m.shape for m in maps], dim)
    if channels_last:
        with torch.cuda.device(maps[0].device):
            rc = L.mdt_pyramid_roi_align_forward_cl(len(maps), _ptr_array(maps), int(maps[0].dtype == torch.bfloat16), H, W, D,
                                                    _lib.ptr(boxes), _lib.ptr(batch_ix), _lib.ptr(level), n, B, C,
                                                    crop[0], crop[1], crop[2], _lib.ptr(crops), _lib.current_stream_ptr())
        if rc != _lib.MDT_ERR_UNSUPPORTED:
            _lib.check(rc, "mdt_pyramid_roi_align_forward_cl")
            return crops
        # outside the channels-last kernel's budgets (ch + cw + cd > 192, C / 4 > 256, tile budget: launch_fwd_cl): the row-major entry serves
        # every shape (it falls back to the generic kernel itself, roi_align.hip) -- one row-major copy of the maps, as before round 5
        maps = [m.contiguous() for m in maps]
    with torch.cuda.device(maps[0].device):
        rc = L.mdt_pyramid_roi_align_forward(dim, len(maps), _ptr_array(maps), int(maps[0].dtype == torch.bfloat16), H, W, D,
                                             _lib.ptr(boxes), _lib.ptr(batch_ix), _lib.ptr(level), n, B, C,
                                             crop[0], crop[1], crop[2] if dim == 3 else 1, _lib.ptr(crops),
                                             _lib.current_stream_ptr())
    _lib.check(rc, "mdt_pyramid_roi_align_forward")
    return crops


def pyramid_backward(grads, boxes, batch_ix, level, shapes, outs=None):
    """grads [N, C, *crop] -> list of grad maps (one per level, fully written).  One launch when every level fits the
    single-launch kernel, else one default backward per level (box_ind = -1 off-level).  `outs`: write into these maps."""
    dim = len(shapes[0]) - 2
    L = _lib.lib()
    grads = grads.contiguous()
    if grads.dtype != torch.float32:
        grads = grads.float()
    dev = grads.device
    n = grads.size(0)
    if n == 0:
        return [torch.zeros(tuple(s), dtype=torch.float32, device=dev) for s in shapes]
    if outs is None:
        outs = [torch.empty(tuple(s), dtype=torch.float32, device=dev) for s in shapes]
    else:
        outs = list(outs)
    crop = tuple(grads.shape[2:])
    H, W, D = _level_dims(shapes, dim)
    prof = PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_valid = ((batch_ix >= 0) & (batch_ix < shapes[0][0]) & (level >= 0) & (level < len(shapes))).sum()
        ev0.record()
    with torch.cuda.device(dev):
        rc = L.mdt_pyramid_roi_align_backward(dim, len(shapes), _lib.ptr(grads), _lib.ptr(boxes), _lib.ptr(batch_ix),
                                              _lib.ptr(level), n, shapes[0][0], shapes[0][1], H, W, D,
                                              crop[0], crop[1], crop[2] if dim == 3 else 1, _ptr_array(outs),
                                              _lib.current_stream_ptr())
    if prof is not None:
        ev1.record()
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        for l, s in enumerate(shapes):
            ind = torch.where(level == l, batch_ix, torch.full_like(batch_ix, -1))
            outs[l] = crop_backward(grads, boxes, ind, tuple(s))
        return outs
    if prof is not None:
        prof.append((ev0, ev1, {"im_size": tuple(shapes[0]), "crop": crop, "n_rows": n, "n_valid": n_valid, "mode": "pyramid",
                                "levels": [tuple(s) for s in shapes]}))
    _lib.check(rc, "mdt_pyramid_roi_align_backward")
    return outs


def pyramid_backward_accumulate(grads, boxes, batch_ix, level, shapes, outs):
    """adds the RoIs' gradient to maps that already hold one (mdt_pyramid_roi_align_backward_accumulate); outside the kernel's budgets: a
    separate backward + add"""
    dim = len(shapes[0]) - 2
    L = _lib.lib()
    grads = grads.contiguous()
    if grads.dtype != torch.float32:
        grads = grads.float()
    n = grads.size(0)
    if n == 0:
        return outs
    crop = tuple(grads.shape[2:])
    H, W, D = _level_dims(shapes, dim)
    with torch.cuda.device(grads.device):
        rc = L.mdt_pyramid_roi_align_backward_accumulate(dim, len(shapes), _lib.ptr(grads), _lib.ptr(boxes), _lib.ptr(batch_ix), _lib.ptr(level), n,
                                                         shapes[0][0], shapes[0][1], H, W, D, crop[0], crop[1], crop[2] if dim == 3 else 1,
                                                         _ptr_array(outs), _lib.current_stream_ptr())
    if rc == _lib.MDT_ERR_UNSUPPORTED:
        for o, extra in zip(outs, pyramid_backward(grads, boxes, batch_ix, level, shapes)):
            o.add_(extra)
        return outs
    _lib.check(rc, "mdt_pyramid_roi_align_backward_accumulate")
    return outs


class PyramidGradAccumulator(object):
    """ONE gradient buffer per pyramid map for all its consumers of a training step (round 6).

    The FPN outputs are read by the classifier head's RoIAlign, the mask head's RoIAlign and the sampled-anchor RPN evaluation.  Left to
    autograd, each hands back a dense gradient per level and the engine adds them: the P2 map (151 MB at 8 x 128^3) is written three times
    and summed twice (one of the adds between a row-major and a channels-last tensor: 125 us).  With an accumulator -- installed by
    models/mrcnn.train_forward_device around the consumers, found by their autograd Functions through `CURRENT` -- the first RoIAlign
    backward writes every byte of the buffers once (zeros included, as always), the second one runs in the kernel's accumulate mode
    (read-modify-write of the quads its RoIs touch), the RPN patches are scatter-added as a few thousand rows, and whoever arrives LAST
    returns the buffers to autograd; the others return None.  Sparse contributions that arrive before a dense one are kept and applied
    after it.  The result is row-major [B, C, *spatial] like every RoIAlign gradient of this package.

    Every registered consumer's backward MUST run in the backward pass (true for the step: all three branches reach the loss);
    `check()` after the backward raises if one did not (its gradients would be missing silently)."""

    CURRENT = None

    def __init__(self, maps):
        self.maps = list(maps)
        self.expected = 0
        self.arrived = 0
        self.bufs = None
        self.pending = []

    def matches(self, maps):
        return len(maps) == len(self.maps) and all(a is b for a, b in zip(maps, self.maps))

    def register(self):
        self.expected += 1

    def _alloc(self):
        self.bufs = [torch.empty(tuple(m.shape), dtype=torch.float32, device=m.device) for m in self.maps]

    def _done(self):
        self.arrived += 1
        if self.arrived < self.expected:
            return None
        if self.bufs is None:                       # only sparse consumers: zero maps + their rows
            self._alloc()
            for b in self.bufs:
                b.zero_()
        for fn in self.pending:
            fn(self.bufs)
        self.pending = []
        out, self.bufs = self.bufs, None
        return out

    def dense(self, launch):
        """launch(bufs, accumulate) writes (accumulate False: every byte) or adds (True) a dense consumer's gradient"""
        if self.bufs is None:
            self._alloc()
            launch(self.bufs, False)
            for fn in self.pending:
                fn(self.bufs)
            self.pending = []
        else:
            launch(self.bufs, True)
        return self._done()

    def sparse(self, add):
        """add(bufs) adds a few rows; deferred until a dense consumer has written the buffers"""
        if self.bufs is None:
            self.pending.append(add)
        else:
            add(self.bufs)
        return self._done()

    def check(self):
        if self.arrived != self.expected:
            raise RuntimeError("PyramidGradAccumulator: %d of %d consumers of the pyramid maps ran their backward -- the gradient of the FPN outputs is "
                               "incomplete (a branch did not reach the loss); run the step without the accumulator" % (self.arrived, self.expected))


class _PyramidRoIAlign(Function):
    @staticmethod
    def forward(ctx, boxes, batch_ix, level, crop, *maps):
        dim = len(crop)
        for m in maps:
            _lib.require_cuda(m, "feature map")
        # inside Function.forward grad mode is always off, so there is nothing to test here: bf16 maps are always read by
        # the bf16-input kernel (exact widening, fp32 interpolation); backward returns fp32 maps cast to ctx.dtypes
        bf16 = all(m.dtype == torch.bfloat16 for m in maps)
        cl = channels_last_eligible(maps, dim)          # the conv path's own layout: pooled as it is, no row-major copy of the pyramid
        maps_c = list(maps) if cl else [m.contiguous() if (m.dtype == torch.float32 or bf16) else m.float().contiguous() for m in maps]
        boxes = boxes.detach().to(device=maps_c[0].device, dtype=torch.float32).contiguous()
        batch_ix = batch_ix.detach().to(device=maps_c[0].device, dtype=torch.int32).contiguous()
        level = level.detach().to(device=maps_c[0].device, dtype=torch.int32).contiguous()
        if boxes.dim() != 2 or boxes.size(1) != 2 * dim:
            raise ValueError("boxes must be [N, %d], got %s" % (2 * dim, tuple(boxes.shape)))
        crops = pyramid_forward(maps_c, boxes, batch_ix, level, crop, channels_last=cl)
        ctx.shapes = [tuple(m.shape) for m in maps_c]
        ctx.dtypes = [m.dtype for m in maps]
        ctx.save_for_backward(boxes, batch_ix, level)
        acc = PyramidGradAccumulator.CURRENT
        ctx.acc = None
        if acc is not None and acc.matches(maps) and any(ctx.needs_input_grad[4:]) and all(dt == torch.float32 for dt in ctx.dtypes):
            acc.register()
            ctx.acc = acc
        return crops

    @staticmethod
    def backward(ctx, grad_outputs):
        boxes, batch_ix, level = ctx.saved_tensors
        if ctx.acc is not None:
            shapes = ctx.shapes

            def launch(bufs, accumulate):
                if not accumulate:
                    res = pyramid_backward(grad_outputs, boxes, batch_ix, level, shapes, outs=bufs)
                    for l, r in enumerate(res):          # (outside the one-launch kernel's budgets the per-level fallback returns fresh tensors)
                        if r is not bufs[l]:
                            bufs[l].copy_(r)
                else:
                    pyramid_backward_accumulate(grad_outputs, boxes, batch_ix, level, shapes, bufs)
            outs = ctx.acc.dense(launch)
            return (None, None, None, None) + (tuple(outs) if outs is not None else (None,) * len(shapes))
        outs = pyramid_backward(grad_outputs, boxes, batch_ix, level, ctx.shapes)
        outs = [o if o.dtype == dt else o.to(dt) for o, dt in zip(outs, ctx.dtypes)]
        return (None, None, None, None) + tuple(outs)


def pyramid_crop_and_resize(maps, boxes, batch_ix, level, crop):
    """RoIAlign of every RoI on ITS pyramid level (level[n] indexes `maps`), rows in input order; gradients to the maps."""
    return _PyramidRoIAlign.apply(boxes, batch_ix, level, tuple(int(c) for c in crop), *maps)
