"""Device-resident mirror of the detection-op half of the reference's utils/model_utils.py.

Same names, argument meaning and return conventions; the numpy / many-tiny-torch-op bodies are
replaced by the gfx950 kernels behind include/mdt_hip.h.  Tensors live on the GPU; functions that
the reference runs on the host with numpy (anchors, matching) return DEVICE tensors here (use
.cpu().numpy() for the reference's exact return type).
"""
import ctypes

import numpy as np
import torch

from .. import _lib


def _dev(device):
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


# --------------------------------------------------------------------------- anchors
def generate_anchors_level(dim, scales_xy, scales_z, ratios, shape, feature_stride_xy, feature_stride_z,
                           anchor_stride, device=None):
    """One pyramid level: generate_anchors (utils/model_utils.py:190-226) / generate_anchors_3D (:230-272).
    Returns (float64 [n, 2*dim], float32 [n, 2*dim]) device tensors."""
    L = _lib.lib()
    device = _dev(device)
    sxy = np.ascontiguousarray(scales_xy, dtype=np.float64)
    sz = np.ascontiguousarray(scales_z if scales_z is not None else np.zeros_like(sxy), dtype=np.float64)
    if dim == 3 and len(sz) != len(sxy):
        raise ValueError("scales_xy and scales_z must have the same length (np.tile at :249 assumes it)")
    rt = np.ascontiguousarray(ratios, dtype=np.float64)
    shp = np.ascontiguousarray(shape, dtype=np.int32)
    pos = 1
    for s in shp[:dim]:
        pos *= (int(s) + anchor_stride - 1) // anchor_stride
    n = pos * len(sxy) * len(rt)
    out = torch.empty((n, 2 * dim), dtype=torch.float64, device=device)
    out32 = torch.empty((n, 2 * dim), dtype=torch.float32, device=device)
    if n == 0:
        return out, out32
    with torch.cuda.device(device):
        rc = L.mdt_generate_anchors(dim, sxy.ctypes.data, sz.ctypes.data, len(sxy), rt.ctypes.data, len(rt),
                                    shp.ctypes.data, ctypes.c_double(float(feature_stride_xy)),
                                    ctypes.c_double(float(feature_stride_z)), int(anchor_stride),
                                    _lib.ptr(out), _lib.ptr(out32), _lib.current_stream_ptr())
    _lib.check(rc, "mdt_generate_anchors")
    return out, out32


def generate_pyramid_anchors(logger, cf, device=None, return_f32=False):
    """generate_pyramid_anchors (utils/model_utils.py:275-314): levels concatenated in cf.pyramid_levels order."""
    outs, outs32 = [], []
    for level in cf.pyramid_levels:
        shape = cf.backbone_shapes[level]
        dim = len(shape)
        a, a32 = generate_anchors_level(
            dim, cf.rpn_anchor_scales["xy"][level], cf.rpn_anchor_scales["z"][level] if dim == 3 else None,
            cf.rpn_anchor_ratios, shape, cf.backbone_strides["xy"][level],
            cf.backbone_strides["z"][level] if dim == 3 else 1, cf.rpn_anchor_stride, device)
        outs.append(a)
        outs32.append(a32)
        if logger is not None:
            logger.info("level {}: built anchors {}".format(level, tuple(a.shape)))
    anchors = torch.cat(outs, 0)
    if return_f32:
        return anchors, torch.cat(outs32, 0)
    return anchors


_STAGE_POOL = None


def _stage_pool():
    global _STAGE_POOL
    if _STAGE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _STAGE_POOL = ThreadPoolExecutor(max_workers=6, thread_name_prefix="mdt-stage")
    return _STAGE_POOL


RING_WAIT_S = [0.0]      # wall time spent waiting for a ring slot's previous copy (back-pressure: the host is a ring-depth ahead of the GPU)


class PinnedStager(object):
    """A ring of persistent pinned host buffers for ONE upload channel (the image of a batch, its GT masks, its seg map).
    torch's caching host allocator hands a block out again only once the copy that used it has EXECUTED; a host that runs a step
    or more ahead of the GPU therefore gets a fresh hipHostMalloc (page-locking 67 MB: 10-20 ms, plus the implicit device
    synchronisation of hipHostFree) on every step -- measured 100-400 ms of host time per step with numpy batches
    (profiles/r03_host_issue_probe.json).  Here a slot is reused after waiting for ITS OWN last copy (`depth` steps back), so the
    host never runs more than `depth` uploads ahead and never allocates in steady state."""

    def __init__(self, depth=3):
        self.slots = [None] * depth
        self.events = [None] * depth
        self.at = 0

    def acquire(self, nbytes):
        k = self.at
        self.at = (k + 1) % len(self.slots)
        if self.events[k] is not None:
            if not self.events[k].query():
                import time
                t0 = time.perf_counter()
                self.events[k].synchronize()
                RING_WAIT_S[0] += time.perf_counter() - t0
            self.events[k] = None
        buf = self.slots[k]
        if buf is None or buf.numel() < nbytes:
            buf = self.slots[k] = torch.empty(max(int(nbytes * 1.25), 4096), dtype=torch.uint8, pin_memory=True)
        return k, buf[:nbytes]

    def release(self, k):
        """call after the asynchronous copy out of slot k has been enqueued (records on the current stream)"""
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev


_STAGERS = {}


def _stager(device, channel):
    key = (str(torch.device(device)), channel)
    st = _STAGERS.get(key)
    if st is None:
        st = _STAGERS[key] = PinnedStager()
    return st


def _as_cpu_tensor(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))


def _np_copy(dst, src):
    """CPU tensor -> CPU tensor by numpy: a plain memcpy that releases the GIL.  (torch's copy_ runs on the OpenMP pool; called
    from a second thread it spins up a second team of one thread per core next to the main thread's -- measured: the host side of
    a step went from 70 to 190 ms, profiles/r03_host_issue_probe.json.)"""
    np.copyto(dst.numpy(), src.numpy())


def _stack_into(raw, parts, pool=None, split=2):
    """the CPU tensors `parts` stacked along dim 0 into the pinned byte buffer `raw` (no intermediate torch.cat copy); with a
    pool, parts of 32 MB and more are copied as `split` slices in parallel (one core moves 8-16 GB/s)"""
    n0 = sum(int(t.shape[0]) for t in parts)
    pinned = raw.view(parts[0].dtype).view((n0,) + tuple(parts[0].shape[1:]))
    at, jobs = 0, []
    for t in parts:
        dst = pinned[at:at + int(t.shape[0])]
        at += int(t.shape[0])
        if pool is not None and t.numel() * t.element_size() >= (32 << 20):
            fd, fs = dst.view(-1), t.view(-1)
            step = (fs.numel() + split - 1) // split
            jobs += [pool.submit(_np_copy, fd[i:i + step], fs[i:i + step]) for i in range(0, fs.numel(), step)]
        else:
            _np_copy(dst, t)
    for j in jobs:
        j.result()
    return pinned


_SMALL_UPLOAD = 1 << 20


def upload(array_or_tensor, device, channel="data"):
    """host numpy array / CPU tensor -> device through PINNED staging memory with an asynchronous copy: a pageable
    upload waits for the stream (the host loses its run-ahead) and moves at a fraction of the PCIe rate.  Bulky arrays go
    through the persistent ring of their `channel` (PinnedStager); small ones through torch's caching host allocator."""
    t = _as_cpu_tensor(array_or_tensor)
    if t.device.type != "cpu" or torch.device(device).type != "cuda":
        return t.to(device, non_blocking=True)
    nbytes = t.numel() * t.element_size()
    if nbytes < _SMALL_UPLOAD or t.dim() == 0:
        return t.pin_memory().to(device, non_blocking=True)
    st = _stager(device, channel)
    k, raw = st.acquire(nbytes)
    d = _stack_into(raw, [t.contiguous()], pool=_stage_pool()).to(device, non_blocking=True)
    st.release(k)
    return d


_PENDING_SLOT = {}


def stage_pinned(array_or_parts, device, channel):
    """host numpy array / CPU tensor (or a list of them, stacked along dim 0) -> a view of the channel's persistent pinned ring slot
    holding a copy of it.  The caller enqueues ITS OWN asynchronous copy out of the slot (e.g. straight into a static graph input,
    training.GraphedTrainStep) and then calls stage_release(device, channel), which guards the slot with an event."""
    parts = array_or_parts if isinstance(array_or_parts, (list, tuple)) else [array_or_parts]
    parts = [_as_cpu_tensor(p).contiguous() for p in parts]
    st = _stager(device, channel)
    nbytes = sum(t.numel() * t.element_size() for t in parts)
    k, raw = st.acquire(nbytes)
    _PENDING_SLOT[(str(torch.device(device)), channel)] = k
    return _stack_into(raw, parts, pool=_stage_pool())


def stage_release(device, channel):
    """call after the asynchronous copy out of the slot stage_pinned() returned has been enqueued on the current stream"""
    _stager(device, channel).release(_PENDING_SLOT.pop((str(torch.device(device)), channel)))


class StagedUpload(object):
    """Upload whose host half (stacking `parts` along dim 0 into pinned memory) runs on a background thread while the caller
    keeps launching kernels; `.get()` -- called on the caller's thread, i.e. on ITS current stream -- waits for the staging
    and enqueues the asynchronous copy.  Used for the GT masks of a training batch, which the step needs only after the
    backbone, the RPN and the proposal layer have been launched (models/mrcnn.py train_forward)."""

    def __init__(self, parts, device, channel="masks"):
        self.device = torch.device(device)
        self.parts = [_as_cpu_tensor(p).contiguous() for p in parts]
        self.future = self.stager = None
        if self.parts and self.device.type == "cuda":
            self.stager = _stager(self.device, channel)
            nbytes = sum(t.numel() * t.element_size() for t in self.parts)
            self.future = _stage_pool().submit(self._stage, nbytes)

    def _stage(self, nbytes):
        with torch.cuda.device(self.device):       # the current device is per thread: a pool thread starts on device 0
            k, raw = self.stager.acquire(nbytes)
            return k, _stack_into(raw, self.parts)

    def get(self):
        if not self.parts:
            return None
        if self.future is None:
            return torch.cat(self.parts, 0).to(self.device)
        k, pinned = self.future.result()
        d = pinned.to(self.device, non_blocking=True)
        self.stager.release(k)
        return d


# --------------------------------------------------------------------------- one device->host copy for a set of small tensors
def pack_for_readout(items):
    """[(name, device tensor)] -> (one int32 device buffer, layout): fp32 tensors travel as their bit pattern, bool / integer ones as
    int32.  Device side of a read-out that costs ONE device->host copy (and one sync) instead of one per tensor; launches only."""
    flat, layout, at = [], [], 0
    for name, t in items:
        t = t.detach()
        if t.dtype == torch.float64:
            t = t.float()
        if t.dtype == torch.float32:
            f, kind = t.contiguous().view(-1).view(torch.int32), "f32"
        else:
            f, kind = t.to(torch.int32).reshape(-1), "i32"
        flat.append(f)
        layout.append((name, kind, tuple(t.shape), at, int(f.numel())))
        at += int(f.numel())
    return torch.cat(flat), layout


class DeferredReadout(object):
    """The monitoring read-out one step LATE: the packed device buffer of step i travels to a pinned host buffer with an asynchronous copy
    (no sync in step i), and is turned into the reference's results entries when step i + 1 asks -- by then the copy has long finished, so
    the host never waits for the GPU and the GPU never idles while the host builds box lists.  exec.py:76-79 logs / collects the same
    values, one batch later.  Two pinned buffers alternate (step i + 1's copy must not land in the buffer step i's values are read from)."""

    def __init__(self):
        self.bufs = [None, None]
        self.extra_bufs = [None, None]
        self.k = 0
        self.pending = None          # (event, host tensor, layout, context)

    def push(self, packed, context, extra=None):
        """enqueue the copy of `packed` = (device int32 buffer, layout); returns what was pending before (or None).
        extra: one more device tensor (any dtype, e.g. the Retina U-Net's uint8 label map) that travels the same way; `resolve` hands its
        pinned host copy back inside the context tuple's place: context becomes (context, host_extra)."""
        buf, layout = packed
        prev = self.pending
        host = self.bufs[self.k]
        if host is None or host.numel() < buf.numel():
            host = self.bufs[self.k] = torch.empty(int(buf.numel()), dtype=torch.int32).pin_memory()
        view = host[:buf.numel()]
        view.copy_(buf.detach(), non_blocking=True)
        if extra is not None:
            e = extra.detach().contiguous()
            eh = self.extra_bufs[self.k]
            if eh is None or eh.numel() < e.numel() or eh.dtype != e.dtype:
                eh = self.extra_bufs[self.k] = torch.empty(int(e.numel()), dtype=e.dtype).pin_memory()
            ev_view = eh[:e.numel()].view(e.shape)
            ev_view.copy_(e, non_blocking=True)
            context = (context, ev_view)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = (ev, view, layout, context)
        self.k ^= 1
        return prev

    @staticmethod
    def resolve(entry):
        """(numpy view of the arrived buffer, layout), context -- waits for the copy's event (a no-op one step later)"""
        ev, view, layout, context = entry
        ev.synchronize()
        return (view.numpy(), layout), context

    def flush(self):
        """the last step's entry (end of an epoch)"""
        prev, self.pending = self.pending, None
        return prev


def unpack_readout(packed):
    """host side of pack_for_readout: {name: numpy array}; `packed[0]` may be the device buffer (copied here: the one sync of the
    read-out) or a host tensor / numpy array that was already copied"""
    buf, layout = packed
    host = buf.detach().cpu().numpy() if torch.is_tensor(buf) else np.asarray(buf)
    out = {}
    for name, kind, shape, at, n in layout:
        a = host[at:at + n]
        out[name] = (a.view(np.float32) if kind == "f32" else a).reshape(shape)
    return out


# --------------------------------------------------------------------------- anchor <-> GT matching
_CONST = {}


def const_tensor(values, dtype, device):
    """Small constants (scales, std devs, clip windows) live on the device, uploaded ONCE per (values, dtype, device).
    A host->device copy from pageable memory waits for the stream, so re-creating them inside a step drains the launch
    queue every time (torch.tensor(..., device=) / torch.as_tensor(numpy, device=) are such copies).  Read-only."""
    flat = np.asarray(values, dtype=np.float64).reshape(-1)
    key = (tuple(flat.tolist()), np.asarray(values).shape, dtype, str(device))
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.as_tensor(np.asarray(values, dtype=np.float64), dtype=dtype, device=device)
    return t


def anchor_match_labels(anchors, gt_boxes, gt_class_ids, neg_thresh, pos_thresh):
    """Steps 1-3 of gt_anchor_matching (utils/model_utils.py:505-563) on the device.
    anchors [A, 2*dim] f64, gt_boxes [G, 2*dim] f64, gt_class_ids [G] i32 or None.
    Returns matches [A] i32, iou_argmax [A] i32, iou_max [A] f64, gt_best_anchor [G] i32."""
    L = _lib.lib()
    dev = anchors.device
    A, dim = anchors.size(0), anchors.size(1) // 2
    anchors = anchors.contiguous()
    G = 0 if gt_boxes is None else gt_boxes.size(0)
    matches = torch.empty(A, dtype=torch.int32, device=dev)
    argmax = torch.empty(A, dtype=torch.int32, device=dev)
    iou_max = torch.empty(A, dtype=torch.float64, device=dev)
    gt_best = torch.empty(max(G, 1), dtype=torch.int32, device=dev)
    wsb = L.mdt_anchor_match_workspace_bytes(A, max(G, 1))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    gtb = gt_boxes.to(device=dev, dtype=torch.float64).contiguous() if G else None
    cls = gt_class_ids.to(device=dev, dtype=torch.int32).contiguous() if (G and gt_class_ids is not None) else None
    with torch.cuda.device(dev):
        rc = L.mdt_anchor_match(_lib.ptr(anchors), A, dim, _lib.ptr(gtb), _lib.ptr(cls), G,
                                ctypes.c_double(neg_thresh), ctypes.c_double(pos_thresh),
                                _lib.ptr(matches), _lib.ptr(argmax), _lib.ptr(iou_max), _lib.ptr(gt_best),
                                _lib.ptr(ws), wsb, _lib.current_stream_ptr())
    _lib.check(rc, "mdt_anchor_match")
    return matches, argmax, iou_max, gt_best[:G]


def anchor_match_labels_batched(anchors, gt_boxes, n_gt, gt_class_ids=None, neg_thresh=0.01, pos_thresh=0.5):
    """anchor_match_labels for a whole batch in ONE launch pair (mdt_anchor_match_batched), the per-element GT counts read on the
    device: replaces the loop over batch elements of models/mrcnn.py:894 / retina_unet.py:408 -- and, because no host-side count
    enters the launch, keeps the training step free of host-dependent launch parameters (capturable in a hipGraph).
    anchors [A, 2*dim] f64; gt_boxes [B, Gmax, 2*dim] f64 (rows >= n_gt[b] ignored); n_gt [B] i32 DEVICE tensor;
    gt_class_ids [B, Gmax] i32 or None.  Returns matches [B, A] i32, iou_argmax [B, A] i32."""
    L = _lib.lib()
    dev = anchors.device
    A, dim = anchors.size(0), anchors.size(1) // 2
    B, gmax = int(gt_boxes.shape[0]), int(gt_boxes.shape[1])
    anchors = anchors.contiguous()
    gt_boxes = gt_boxes.to(dtype=torch.float64).contiguous()
    n_gt = n_gt.to(dtype=torch.int32).contiguous()
    cls = gt_class_ids.to(dtype=torch.int32).contiguous() if gt_class_ids is not None else None
    matches = torch.empty((B, A), dtype=torch.int32, device=dev)
    argmax = torch.empty((B, A), dtype=torch.int32, device=dev)
    gt_best = torch.empty((B, gmax), dtype=torch.int32, device=dev)
    wsb = L.mdt_anchor_match_batched_workspace_bytes(A, B, gmax)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.mdt_anchor_match_batched(_lib.ptr(anchors), A, dim, B, _lib.ptr(gt_boxes), _lib.ptr(cls), _lib.ptr(n_gt), gmax,
                                        ctypes.c_double(neg_thresh), ctypes.c_double(pos_thresh), _lib.ptr(matches), _lib.ptr(argmax),
                                        None, _lib.ptr(gt_best), _lib.ptr(ws), wsb, _lib.current_stream_ptr())
    _lib.check(rc, "mdt_anchor_match_batched")
    return matches, argmax


def anchor_delta_targets(anchors, gt_boxes, std_dev):
    """Delta targets of matched (anchor, gt) rows, utils/model_utils.py:575-617 (float64)."""
    dim = anchors.size(1) // 2
    a, g = anchors.double(), gt_boxes.double()
    a_h, a_w = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
    g_h, g_w = g[:, 2] - g[:, 0], g[:, 3] - g[:, 1]
    a_cy, a_cx = a[:, 0] + 0.5 * a_h, a[:, 1] + 0.5 * a_w
    g_cy, g_cx = g[:, 0] + 0.5 * g_h, g[:, 1] + 0.5 * g_w
    cols = [(g_cy - a_cy) / a_h, (g_cx - a_cx) / a_w]
    if dim == 3:
        a_d, g_d = a[:, 5] - a[:, 4], g[:, 5] - g[:, 4]
        a_cz, g_cz = a[:, 4] + 0.5 * a_d, g[:, 4] + 0.5 * g_d
        cols += [(g_cz - a_cz) / a_d, torch.log(g_h / a_h), torch.log(g_w / a_w), torch.log(g_d / a_d)]
    else:
        cols += [torch.log(g_h / a_h), torch.log(g_w / a_w)]
    std = const_tensor(std_dev, torch.float64, a.device)
    return torch.stack(cols, 1) / std


def gt_anchor_matching(cf, anchors, gt_boxes, gt_class_ids=None, generator=None):
    """gt_anchor_matching (utils/model_utils.py:505-619).  anchors: [A, 2*dim] f64 DEVICE tensor.
    Returns (anchor_class_matches [A] int32 device, anchor_delta_targets [rpn_train_anchors_per_image, 2*dim]
    float64 device).  The random subsampling of surplus positives (:566-571, np.random.choice in the
    reference) uses torch's device RNG."""
    dev = anchors.device
    dim = anchors.size(1) // 2
    n_t = cf.rpn_train_anchors_per_image
    targets = torch.zeros((n_t, 2 * dim), dtype=torch.float64, device=dev)
    if gt_boxes is None or len(gt_boxes) == 0:
        return torch.full((anchors.size(0),), -1, dtype=torch.int32, device=dev), targets
    gt_t = torch.as_tensor(np.asarray(gt_boxes, dtype=np.float64) if not torch.is_tensor(gt_boxes) else gt_boxes,
                           dtype=torch.float64, device=dev)
    cls_t = None
    if gt_class_ids is not None:
        cls_t = torch.as_tensor(np.asarray(gt_class_ids) if not torch.is_tensor(gt_class_ids) else gt_class_ids,
                                device=dev).to(torch.int32)
    neg = 0.1 if dim == 2 else 0.01
    matches, argmax, _, _ = anchor_match_labels(anchors, gt_t, cls_t, neg, float(cf.anchor_matching_iou))
    ids = torch.nonzero(matches > 0)[:, 0]
    extra = ids.numel() - (n_t // 2)
    if extra > 0:
        perm = torch.randperm(ids.numel(), device=dev, generator=generator)[:extra]
        matches[ids[perm]] = 0
        ids = torch.nonzero(matches > 0)[:, 0]
    k = min(ids.numel(), n_t)
    if k > 0:
        ids = ids[:k]
        targets[:k] = anchor_delta_targets(anchors[ids], gt_t[argmax[ids].long()], cf.rpn_bbox_std_dev)
    return matches, targets


# --------------------------------------------------------------------------- box decode + clip
def decode_clip_boxes(boxes, deltas, std_dev, window, order=None, scores=None):
    """Fused apply_box_deltas_{2D,3D}(boxes[order], deltas[order] * std_dev) -> clip_boxes_{2D,3D}(., window)
    (utils/model_utils.py:318-398 as chained in models/mrcnn.py:320-345).  With `scores` (already gathered,
    one per output row) the result is [n, 2*dim+1], directly the input of nms_*."""
    L = _lib.lib()
    dev = boxes.device
    dim = boxes.size(1) // 2
    boxes = boxes.contiguous().float()
    deltas = deltas.contiguous().float()
    n = boxes.size(0) if order is None else order.numel()
    stride = 2 * dim + (1 if scores is not None else 0)
    out = torch.empty((n, stride), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    std = np.ascontiguousarray(std_dev, dtype=np.float32)
    win = np.ascontiguousarray(window, dtype=np.float32)
    order_c = order.contiguous().long() if order is not None else None
    scores_c = scores.contiguous().float() if scores is not None else None
    with torch.cuda.device(dev):
        rc = L.mdt_decode_clip_boxes(_lib.ptr(boxes), _lib.ptr(deltas), _lib.ptr(order_c), _lib.ptr(scores_c), n, dim,
                                     std.ctypes.data, win.ctypes.data, _lib.ptr(out), stride, _lib.current_stream_ptr())
    _lib.check(rc, "mdt_decode_clip_boxes")
    return out


def apply_box_deltas_3D(boxes, deltas):
    """apply_box_deltas_3D (utils/model_utils.py:343-370) without clipping."""
    big = [-3.0e38, -3.0e38, 3.0e38, 3.0e38, -3.0e38, 3.0e38]
    return decode_clip_boxes(boxes, deltas, [1.0] * 6, big)


def apply_box_deltas_2D(boxes, deltas):
    big = [-3.0e38, -3.0e38, 3.0e38, 3.0e38]
    return decode_clip_boxes(boxes, deltas, [1.0] * 4, big)


def clip_boxes(boxes, window):
    """clip_boxes_{2D,3D} / clip_to_window (utils/model_utils.py:374-398, 623-637)."""
    dim = boxes.size(1) // 2
    w = [float(v) for v in window]
    lo = const_tensor([w[0], w[1], w[0], w[1]] + ([w[4], w[4]] if dim == 3 else []), torch.float32, boxes.device)
    hi = const_tensor([w[2], w[3], w[2], w[3]] + ([w[5], w[5]] if dim == 3 else []), torch.float32, boxes.device)
    return torch.min(torch.max(boxes, lo), hi)


def bbox_overlaps(boxes1, boxes2):
    """bbox_overlaps_{2D,3D} (utils/model_utils.py:429-501): IoU [n1, n2], no +1 convention, fp32,
    same operation order, one broadcasted expression instead of repeat/tile."""
    dim = boxes1.size(-1) // 2
    # [..., n1, 2 dim] x [..., n2, 2 dim] -> [..., n1, n2]: leading (batch) axes broadcast, so the per-element loop of the
    # callers is ONE set of ~25 elementwise launches instead of B sets (same operations per entry, same bits)
    b1, b2 = boxes1[..., :, None, :], boxes2[..., None, :, :]
    y1 = torch.max(b1[..., 0], b2[..., 0])
    x1 = torch.max(b1[..., 1], b2[..., 1])
    y2 = torch.min(b1[..., 2], b2[..., 2])
    x2 = torch.min(b1[..., 3], b2[..., 3])
    inter = torch.clamp(x2 - x1, min=0) * torch.clamp(y2 - y1, min=0)
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    if dim == 3:
        z1 = torch.max(b1[..., 4], b2[..., 4])
        z2 = torch.min(b1[..., 5], b2[..., 5])
        inter = inter * torch.clamp(z2 - z1, min=0)
        a1 = a1 * (b1[..., 5] - b1[..., 4])
        a2 = a2 * (b2[..., 5] - b2[..., 4])
    return inter / (a1 + a2 - inter)


bbox_overlaps_2D = bbox_overlaps
bbox_overlaps_3D = bbox_overlaps


def box_refinement(box, gt_box):
    """box_refinement (utils/model_utils.py:114-143)."""
    height, width = box[:, 2] - box[:, 0], box[:, 3] - box[:, 1]
    center_y, center_x = box[:, 0] + 0.5 * height, box[:, 1] + 0.5 * width
    gt_height, gt_width = gt_box[:, 2] - gt_box[:, 0], gt_box[:, 3] - gt_box[:, 1]
    gt_center_y, gt_center_x = gt_box[:, 0] + 0.5 * gt_height, gt_box[:, 1] + 0.5 * gt_width
    dy, dx = (gt_center_y - center_y) / height, (gt_center_x - center_x) / width
    dh, dw = torch.log(gt_height / height), torch.log(gt_width / width)
    if box.shape[1] > 4:
        depth = box[:, 5] - box[:, 4]
        center_z = box[:, 4] + 0.5 * depth
        gt_depth = gt_box[:, 5] - gt_box[:, 4]
        gt_center_z = gt_box[:, 4] + 0.5 * gt_depth
        dz = (gt_center_z - center_z) / depth
        dd = torch.log(gt_depth / depth)
        return torch.stack([dy, dx, dz, dh, dw, dd], dim=1)
    return torch.stack([dy, dx, dh, dw], dim=1)


def shem(roi_probs_neg, negative_count, ohem_poolsize, generator=None):
    """Stochastic hard example mining (utils/model_utils.py:674-691), device-only (no .cpu() round trip)."""
    probs, order = roi_probs_neg[:, 1:].max(1)[0].sort(descending=True)
    select = min(ohem_poolsize * int(negative_count), order.size(0))
    pool_indices = order[:select]
    rand_idx = torch.randperm(pool_indices.size(0), device=pool_indices.device, generator=generator)
    return pool_indices[rand_idx[:negative_count]]


def log2(x):
    """utils/model_utils.py:658-663: log(x) / log(2) in the tensor's dtype."""
    key = ("log2", x.dtype, str(x.device))
    d = _CONST.get(key)
    if d is None:        # log(2) as the device computes it, once (see const_tensor)
        d = _CONST[key] = torch.log(torch.tensor(2.0, dtype=x.dtype, device=x.device))
    return torch.log(x) / d


class NDConvGenerator(object):
    """2D/3D conv (+norm) (+relu) factory with the reference's Sequential layout, so state_dict keys match
    (utils/model_utils.py:732-781).  The convolutions themselves run on MIOpen through torch."""

    def __init__(self, dim):
        self.dim = dim

    def __call__(self, c_in, c_out, ks, pad=0, stride=1, norm=None, relu="relu"):
        import torch.nn as nn
        from . import fused_epilogue as fe
        if norm is None and relu in (None, "relu"):
            # no normalisation layer between conv and activation: bias add (+ residual) + ReLU run as one fused
            # epilogue pass (csrc/epilogue.hip); same Sequential layout / state-dict keys as the reference
            conv = (fe.ConvBias2d if self.dim == 2 else fe.ConvBias3d)(c_in, c_out, kernel_size=ks, padding=pad, stride=stride)
            return conv if relu is None else fe.ConvBiasReLU(conv, nn.ReLU(inplace=True))
        Conv = nn.Conv2d if self.dim == 2 else nn.Conv3d
        conv = Conv(c_in, c_out, kernel_size=ks, padding=pad, stride=stride)
        if norm is not None:
            if norm == "instance_norm":
                norm_layer = (nn.InstanceNorm2d if self.dim == 2 else nn.InstanceNorm3d)(c_out)
            elif norm == "batch_norm":
                norm_layer = (nn.BatchNorm2d if self.dim == 2 else nn.BatchNorm3d)(c_out)
            else:
                raise ValueError("norm type as specified in configs is not implemented... {}".format(norm))
            conv = nn.Sequential(conv, norm_layer)
        if relu is not None:
            if relu == "relu":
                relu_layer = nn.ReLU(inplace=True)
            elif relu == "leaky_relu":
                relu_layer = nn.LeakyReLU(inplace=True)
            else:
                raise ValueError("relu type as specified in configs is not implemented...")
            conv = nn.Sequential(conv, relu_layer)
        return conv
