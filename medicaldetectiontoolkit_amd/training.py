"""Training step driver: the reference's exec.py:67-79 hot loop (forward, zero_grad, backward, Adam step),
plus patch-level data parallelism -- one process per GPU, gradients averaged with a single flat-bucket
all-reduce over RCCL (the whole model is 4.94 M fp32 parameters = 19.75 MB)."""
import os

import weakref

import torch
import torch.distributed as dist


def build_optimizer(net, cf, fused=False, flat=False, grad_sync=None):
    """exec.py:39: Adam(lr=cf.learning_rate[0], weight_decay=cf.weight_decay); fused=True uses torch's single-kernel
    multi-tensor implementation (same update rule); flat=True: FlatAdam (one launch of csrc/adam.hip over flat buffers)."""
    if flat:
        opt = FlatAdam(net.parameters(), lr=cf.learning_rate[0], weight_decay=cf.weight_decay, grad_sync=grad_sync,
                       absent_grad=getattr(cf, "adam_absent_grad", "skip"))
        if hasattr(net, "grad_condition_spec"):
            opt.attach_conditions(net)
        return opt
    kw = {"fused": True} if fused else {}
    return torch.optim.Adam(net.parameters(), lr=cf.learning_rate[0], weight_decay=cf.weight_decay, **kw)


def _view_like(flat, off, p):
    """the n = p.numel() slots of `flat` from `off` on, seen with p's own shape AND strides (channels_last weights are dense,
    permuted views: element k of the slice is the same storage slot k of p)"""
    chunk = flat[off:off + p.numel()]
    return chunk.as_strided(p.size(), p.stride()) if p.is_contiguous() is False and _dense(p) else chunk.view_as(p)


class FlatAdam(torch.optim.Adam):
    """torch.optim.Adam (exec.py:39) with the whole model in four flat fp32 buffers -- parameters, gradients, exp_avg, exp_avg_sq
    (4.94 M values = 19.75 MB each) -- and the update as ONE C call (mdt_adam_flat_segments, csrc/adam.hip: a one-block kernel that
    derives every parameter's own bias corrections + one pass over the buffers) instead of ~20 multi-tensor launches and 3.4 ms of
    host time per step (profiles/r03_host_profile.txt).

    * every `p.data`, `p.grad`, `state[p]['exp_avg']`, `state[p]['exp_avg_sq']` is a VIEW (with p's strides) into the flat buffers,
      in reverse parameter order (= FlatGradAllReduce's order, whose gradient buffer is adopted when one is passed), so
      `state_dict()` / `load_state_dict()` keep torch.optim.Adam's format (checkpoints of the reference's Adam load unchanged);
    * gradients: with a FlatGradAllReduce (N > 1) they LIVE in its flat buffer (`p.grad` views; `zero_grad()` = one fill).  Without
      one, `zero_grad()` sets them to None so that autograd hands its gradient tensors over without an accumulate-add per
      parameter (130 tiny launches per step), and `step()` gathers them into the flat buffer with one `torch._foreach_copy_`;
    * **per-parameter semantics (round 5)**: every parameter is a SEGMENT of the flat buffers with its own step counter (device
      int32, torch's state[p]['step']) and a parameter WITHOUT a gradient in a step is skipped exactly like torch.optim.Adam skips
      `p.grad is None` -- no moment decay, no weight decay, no update, counter unchanged.  "Without a gradient" has two sources:
      (a) the host knows it (`p.grad is None`; with a FlatGradAllReduce: no accumulate hook fired for p since `zero()`), and
      (b) the step says so ON THE DEVICE (`attach_conditions`): the reference's loss helpers return constants when a step samples
      no positive RoI / anchor (mrcnn.py:233-234, 266-268, 287-288), so its mask head, `linear_bbox` and `conv_bbox` get no gradient
      there, while this repo's fixed-size masked step hands them an exact ZERO gradient -- the model writes the counts into a small
      device tensor (`net.grad_cond`) and the kernel treats a segment whose condition is 0 as "no gradient".  With N > 1 that
      tensor is the tail of the all-reduced gradient buffer: a head is updated iff ANY rank had a positive sample.
      `absent_grad="zero_after_first"` selects the behaviour of the reference's pinned torch 0.4.1 instead, whose `zero_grad()`
      leaves zero tensors behind: after its first gradient a parameter is updated in every step (g = 0 when it has none).
    * a loaded state with DIFFERENT step counts per parameter (a reference checkpoint whose mask head lagged) is adopted as is.
    * `arith=1` (default): the fma pattern of torch's foreach kernels -- bit-equal to torch.optim.Adam on this stack.
    * every step checks that each `p.data` is still the view into the flat parameter buffer (`net.to(memory_format=...)`, `.half()`,
      `load_state_dict(assign=True)` or a second FlatAdam over the same net re-point it -- the kernel would then update an orphaned
      buffer and the model would silently stop training): on mismatch the flat buffers are rebuilt from the live parameters.
    * amsgrad / maximize / capturable / differentiable are not supported (the reference never sets them)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, grad_sync=None, absent_grad="skip", arith=1):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam: one parameter group (the reference's exec.py:39 passes net.parameters())")
        if absent_grad not in ("skip", "zero_after_first"):
            raise ValueError("FlatAdam: absent_grad is 'skip' (torch >= 2) or 'zero_after_first' (torch 0.4.1)")
        self._grad_sync = grad_sync
        self._policy = 0 if absent_grad == "skip" else 1
        self._arith = int(arith)
        self._flat = None           # (param, grad, exp_avg, exp_avg_sq)
        self._cond = None           # (net, {id(param): condition index}, n_conditions)
        self._steps_host = None     # per-segment counters as last read from the device (state_dict())

    def attach_conditions(self, net):
        """the model tells which parameters get a gradient only under a condition the step evaluates on the device
        (net.grad_condition_spec() -> [(name, [parameters])], net.grad_cond: float tensor [len(spec)], > 0 = gradient exists)"""
        spec = net.grad_condition_spec()
        cmap = {}
        for i, (_, plist) in enumerate(spec):
            for p in plist:
                cmap[id(p)] = i
        self._cond = (net, cmap, len(spec))
        # the tensor exists from now on (the first step's forward runs BEFORE the lazy `_build`): ones = "everything has a gradient"
        self._cond_buf = torch.ones(len(spec), dtype=torch.float32, device=next(net.parameters()).device)
        net.set_grad_cond_buffer(self._cond_buf)
        if self._flat is not None:      # called after steps: the per-parameter counters live on the device -- keep them for the re-build
            for p, t in zip(self._rparams, self._device_steps()):
                self.state[p]["step"] = torch.tensor(float(t))
        self._flat = None
        return self

    def _build(self):
        params = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if not params:
            raise ValueError("FlatAdam: no trainable parameter")
        dev = params[0].device
        if any(p.device != dev or p.dtype != torch.float32 or not (p.is_contiguous() or _dense(p)) for p in params):
            raise ValueError("FlatAdam: parameters must be dense fp32 tensors on one device")
        n = sum(p.numel() for p in params)
        gs = self._grad_sync
        n_cond = self._cond[2] if self._cond is not None else 0
        if gs is not None:
            if gs.flat is None or gs.n_extra < n_cond:
                gs.n_extra = max(gs.n_extra, n_cond)
                gs._build()
            if len(gs.params) != len(params) or any(a is not b for a, b in zip(gs.params, params)):
                raise ValueError("FlatAdam: grad_sync was built over a different parameter list")
            fgrad = gs.flat if gs.n_extra == 0 else gs.flat[:n]
            owner = getattr(gs, "_div_owner", None)
            if owner is not None and owner() is not None and owner() is not self:
                raise ValueError("FlatAdam: this grad_sync already feeds another FlatAdam (its deferred division by the world size belongs to ONE optimizer)")
            gs.defer_div = True         # this optimizer divides by the world size inside its launch
            gs._div_owner = weakref.ref(self)
            cond_t = gs.extra[:n_cond] if n_cond else None
            if cond_t is not None and cond_t.data_ptr() != self._cond_buf.data_ptr():
                cond_t.copy_(self._cond_buf)                 # what the step has already written (the first forward precedes this build)
        else:
            fgrad = torch.zeros(n, dtype=torch.float32, device=dev)
            cond_t = self._cond_buf if n_cond else None
        if self._cond is not None:
            self._cond[0].set_grad_cond_buffer(cond_t)      # the step writes its counts straight into this tensor
            self._cond_buf = cond_t
        fparam = torch.empty(n, dtype=torch.float32, device=dev)
        fm = torch.zeros(n, dtype=torch.float32, device=dev)
        fv = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        gviews, offs, steps0, cond_ids = [], [0], [], []
        prev_steps = self._device_steps() if self._flat is not None else None       # a re-home keeps the counters
        with torch.no_grad():
            for k, p in enumerate(reversed(params)):                       # backward order, as FlatGradAllReduce lays the gradients out
                v = _view_like(fparam, off, p)
                v.copy_(p)
                old_state = self.state.get(p, {})
                gviews.append(_view_like(fgrad, off, p))
                p.data = v
                st = {"step": torch.tensor(0.0), "exp_avg": _view_like(fm, off, p), "exp_avg_sq": _view_like(fv, off, p)}
                if "exp_avg" in old_state:                    # state loaded (or stepped by plain Adam) before the first flat step
                    st["exp_avg"].copy_(old_state["exp_avg"])
                    st["exp_avg_sq"].copy_(old_state["exp_avg_sq"])
                    st["step"] = torch.tensor(float(old_state["step"]))
                if prev_steps is not None and len(prev_steps) == len(params):
                    st["step"] = torch.tensor(float(prev_steps[k]))
                steps0.append(int(float(st["step"])))
                self.state[p] = st
                off += p.numel()
                offs.append(off)
                cond_ids.append(self._cond[1].get(id(p), -1) if self._cond is not None else -1)
        self._params = params
        self._rparams = list(reversed(params))
        self._pptr = [p.data_ptr() for p in self._rparams]       # where each parameter must still live at every step
        self._gviews = gviews                       # flat-gradient views, in `_rparams` order
        self._gdirty = [False] * len(gviews)        # view holds a gradient of an earlier step
        self._flat = (fparam, fgrad, fm, fv)
        nseg = len(params)
        self._seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._seg_step = torch.tensor(steps0, dtype=torch.int32, device=dev)
        self._seg_cond = torch.tensor(cond_ids, dtype=torch.int32, device=dev) if n_cond else None
        self._cond_t = cond_t
        self._present_host = bytearray(b"\x01" * nseg)                          # what the device copy currently holds
        self._present_dev = torch.ones(nseg, dtype=torch.uint8, device=dev)
        self._present_pin = torch.ones(nseg, dtype=torch.uint8).pin_memory() if dev.type == "cuda" else torch.ones(nseg, dtype=torch.uint8)
        self._present_ev = None
        self._ws = torch.empty(max(16 * nseg, 16), dtype=torch.uint8, device=dev)
        self._steps_host = None

    def _device_steps(self):
        return [int(v) for v in self._seg_step.cpu().tolist()]

    def state_dict(self):
        """torch.optim.Adam's format; the per-parameter step counters live on the device and are read back here (one small copy)"""
        if self._flat is not None:
            for p, t in zip(self._rparams, self._device_steps()):
                self.state[p]["step"] = torch.tensor(float(t))
        return super().state_dict()

    def zero_grad(self, set_to_none=True):
        if self._flat is None:
            self._build()
        if self._grad_sync is not None:
            self._grad_sync.zero()
        else:
            for p in self._params:
                p.grad = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = None           # re-adopt the loaded moments and step counters at the next step (`_build` copies them into the flat buffers)

    def _agree_present(self, flags):
        """the host-side presence flags come from rank-LOCAL accumulate hooks; ranks that disagreed would take different skip
        decisions and their weights and step counters would drift apart silently.  With N > 1 the flags of the FIRST step are combined
        once (MAX all-reduce over the process group: every rank reaches its first step, so the collective sequence stays rank-symmetric)
        and that agreed set is used from then on -- the set is structural (the reference FPN's unused P1 convolutions; data-dependent
        absences are the device-side conditions, which already travel with the gradient all-reduce).  A later step in which this rank
        holds a gradient for a parameter OUTSIDE the agreed set raises instead of diverging; a parameter inside the set without a local
        gradient contributes its zeros like in every DDP."""
        gs = self._grad_sync
        if not gs._active() or dist.get_world_size() == 1:
            return flags
        agreed = getattr(self, "_present_agreed", None)
        if agreed is None or len(agreed) != len(flags):
            dev = self._flat[0].device if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.tensor(list(flags), dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            agreed = self._present_agreed = bytearray(int(v) for v in t.cpu().tolist())
        extra = [i for i in range(len(flags)) if flags[i] and not agreed[i]]
        if extra:
            raise RuntimeError("FlatAdam: rank %d holds gradients for %d parameter(s) that no rank had in the first step (first: #%d, shape %s) -- the "
                               "ranks' gradient structure diverged; data-dependent absences belong in net.grad_condition_spec()" % (
                                   dist.get_rank(), len(extra), extra[0], tuple(self._rparams[extra[0]].shape)))
        return bytearray(agreed)

    def _set_present(self, flags):
        """host-known gradient presence per segment (bytes, `_rparams` order) -> device, only when it changed"""
        if flags == self._present_host:
            return
        if self._present_ev is not None:
            self._present_ev.synchronize()      # the pinned staging buffer of the previous change must have been read (rare path)
        self._present_pin.copy_(torch.frombuffer(flags, dtype=torch.uint8))
        self._present_dev.copy_(self._present_pin, non_blocking=True)
        if self._present_dev.is_cuda:
            self._present_ev = torch.cuda.Event()
            self._present_ev.record()
        self._present_host = bytearray(flags)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._flat is None:
            self._build()
        g = self.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            raise ValueError("FlatAdam: amsgrad / maximize / capturable / differentiable are not supported")
        if any(p.data_ptr() != q for p, q in zip(self._rparams, self._pptr)):
            if getattr(self, "_graph_attached", False):
                raise RuntimeError("FlatAdam: a parameter was re-pointed while a GraphedTrainStep holds the old addresses; re-capture the step")
            self._build()                # a parameter was re-pointed: re-home all of them (moments and counters are carried over by `_build`)
        fparam, fgrad, fm, fv = self._flat
        flags = bytearray(b"\x01" * len(self._rparams))
        if self._grad_sync is not None:
            touched = self._grad_sync.touched
            for i, p in enumerate(self._rparams):       # a dropped view (someone's zero_grad(set_to_none=True)) would silently freeze p
                if p.grad is None:
                    raise RuntimeError("FlatAdam: a gradient view was dropped; clear gradients with FlatAdam.zero_grad()")
                if touched is not None and not touched[i]:
                    flags[i] = 0
            flags = self._agree_present(flags)
        else:                            # gather autograd's gradient tensors into the flat buffer: one multi-tensor copy
            dst, src = [], []
            for i, p in enumerate(self._rparams):
                g_i = p.grad
                if g_i is None:
                    flags[i] = 0
                    if self._gdirty[i]:              # last step's gradient must not be applied again (zero_after_first reads the buffer)
                        self._gviews[i].zero_()
                        self._gdirty[i] = False
                elif g_i.data_ptr() != self._gviews[i].data_ptr():
                    dst.append(self._gviews[i])
                    src.append(g_i)
                    self._gdirty[i] = True
            if dst:
                torch._foreach_copy_(dst, src)
        self._set_present(flags)
        gdiv = 1.0
        if self._grad_sync is not None:      # the all-reduce left SUMS: the division by the world size rides this launch
            gdiv, self._grad_sync.pending_div = float(self._grad_sync.pending_div), 1.0
        self._update(fparam, fgrad, fm, fv, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                     float(g["weight_decay"]), gdiv)
        return loss

    def _update(self, fparam, fgrad, fm, fv, lr, beta1, beta2, eps, weight_decay, grad_div=1.0):
        """the one C call (csrc/adam.hip); no CPU implementation in the product -- the gloo tests substitute this method"""
        from . import _lib
        with torch.cuda.device(fparam.device):
            rc = _lib.lib().mdt_adam_flat_segments(
                fparam.data_ptr(), fgrad.data_ptr(), fm.data_ptr(), fv.data_ptr(), fparam.numel(), self._seg_off.data_ptr(), len(self._rparams),
                self._seg_step.data_ptr(), self._present_dev.data_ptr(), self._seg_cond.data_ptr() if self._seg_cond is not None else None,
                self._cond_t.data_ptr() if self._cond_t is not None else None, self._policy, self._arith, lr, beta1, beta2, eps, weight_decay,
                grad_div, self._ws.data_ptr(), self._ws.numel(), _lib.raw_stream(fparam))
        _lib.check(rc, "mdt_adam_flat_segments")
        from .utils import fused_epilogue
        fused_epilogue.weights_changed()        # the kernel rewrote the parameters behind torch's version counters (cached flipped filters are stale)


class FlatGradAllReduce(object):
    """Gradient averaging across ranks for patch-level data parallelism (one process per GPU, RCCL over xGMI).

    * The gradients LIVE in one flat fp32 buffer: every `p.grad` is a view (with the parameter's own strides, so
      channels_last weights work) into `self.flat`; autograd accumulates straight into it and the optimizer reads it --
      no copy-in / copy-out kernels.  `zero()` clears all gradients with one fill.
    * The buffer is cut into a few buckets in REVERSE parameter order (the order backward produces gradients).  A
      bucket's all-reduce is launched asynchronously from a post-accumulate hook as soon as its last gradient is ready,
      so the collectives of the head / top-down buckets run under the backward of the encoder (model: 19.75 MB,
      SURVEY.md 8(e)).  Buckets are always launched IN ORDER, so every rank issues the same collective sequence even if
      a parameter gets no gradient on some rank (the reference FPN constructs P1_conv1 / P1_conv2 and Mask R-CNN never
      uses them, backbone.py:112,118; heads may see no positive RoI): such parameters keep their zeros and their bucket
      goes out in `finish()`.  No find_unused_parameters machinery, no rank-dependent hangs.
    """

    def __init__(self, net, n_buckets=4, overlap=True, force=False):
        # force=True: run the collectives even at world size 1 (RCCL self-test on a 1-GPU box: the sum over one rank
        # must leave every gradient bit-identical)
        self.force = bool(force)
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.n_buckets = max(1, int(n_buckets))
        self.overlap = overlap
        self.flat = None
        self._hooks = []
        self._handles = []
        self.suspended = False       # True: the backward hooks launch nothing (GraphedTrainStep: the collectives run after the replay)
        # defer_div: a FlatAdam built over this buffer folds the averaging (sum / world) into its one launch (mdt_adam_flat's grad_div)
        # instead of a separate pass over the buffer; until its step() the buffer then holds the SUM over ranks, pending_div the divisor
        self.defer_div = False
        self.pending_div = 1.0
        # n_extra float slots behind the gradients, all-reduced (summed) with the last bucket: training.FlatAdam keeps the step's
        # "this head had a positive sample" values there (FlatAdam.attach_conditions), so that every rank takes the same skip decision
        self.n_extra = 0
        self.extra = None
        # touched[i]: an accumulate hook fired for parameter i (backward order) since zero() (recorded with and without overlap)
        self.touched = None

    # -- setup (lazy: parameters must already live on their device)
    def _build(self):
        dev = self.params[0].device
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.flat = torch.zeros(self.numel + self.n_extra, dtype=torch.float32, device=dev)
        self.extra = self.flat[self.numel:]
        order = list(reversed(self.params))                      # backward order
        self._index = {p: i for i, p in enumerate(order)}
        target = (self.numel + self.n_buckets - 1) // self.n_buckets
        self.bucket_of, self.bucket_range, self.bucket_left0 = {}, [], []
        off, start, count, b = 0, 0, 0, 0
        for i, p in enumerate(order):
            n = p.numel()
            chunk = self.flat[off:off + n]
            p.grad = chunk.as_strided(p.size(), p.stride()) if p.is_contiguous() is False and _dense(p) else chunk.view_as(p)
            self.bucket_of[p] = b
            off += n
            count += 1
            if (off - start >= target and b < self.n_buckets - 1) or i == len(order) - 1:
                self.bucket_range.append((start, off if i < len(order) - 1 else off + self.n_extra))
                self.bucket_left0.append(count)
                start, count, b = off, 0, b + 1
        self._left = list(self.bucket_left0)
        self._next = 0
        # the hooks always record WHICH parameters received a gradient (`touched`: FlatAdam's presence flags); only with overlap do they
        # also launch the bucket collectives during backward
        self.touched = bytearray(len(order))
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _active(self):
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force)

    def _launch_ready(self, force=False):
        while self._next < len(self.bucket_range) and (force or self._left[self._next] == 0):
            s, e = self.bucket_range[self._next]
            self._handles.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, async_op=True))
            self._next += 1

    def _on_grad(self, p):
        self.touched[self._index[p]] = 1
        if self.suspended or not self.overlap or not self._active():
            return
        b = self.bucket_of[p]
        self._left[b] -= 1
        self._launch_ready()

    # -- per step
    def zero(self):
        """replaces optimizer.zero_grad(): all gradients are views of the flat buffer"""
        if self.flat is None:
            self._build()
        self.flat[:self.numel].zero_()       # (the extra slots are overwritten by the step, not accumulated into)
        if self.touched is not None:
            self.touched = bytearray(len(self.touched))
        self._left = list(self.bucket_left0)
        self._next = 0
        self._handles = []
        for p in self.params:            # an optimizer.zero_grad(set_to_none=True) elsewhere must not detach the views
            if p.grad is None:
                raise RuntimeError("a gradient view was dropped (zero_grad(set_to_none=True)); use FlatGradAllReduce.zero()")

    def finish(self):
        """launch what is still pending (in order), wait, average"""
        if not self._active():
            return
        self._launch_ready(force=True)
        for h in self._handles:
            h.wait()
        self._average()

    def _average(self):
        w = dist.get_world_size()
        if self.defer_div:
            self.pending_div = float(w)
        elif w > 1:
            self.flat.div_(w)

    @property
    def grad_scale(self):
        """what the gradients in the flat buffer (and every p.grad view) must be multiplied by to be the MEAN over ranks: 1 except between
        finish() and the step() of the FlatAdam built over this sync (deferred division: the buffer holds the SUM).  Gradient-norm logging or
        clipping between the two must use it."""
        return 1.0 / self.pending_div

    def finish_all(self):
        """after a captured backward (GraphedTrainStep: hooks suspended): all buckets in order, wait, average"""
        if not self._active():
            return
        self._next = 0
        self._handles = []
        self._launch_ready(force=True)
        for h in self._handles:
            h.wait()
        self._average()

    def __call__(self):          # round-1 spelling
        self.finish()


def _dense(p):
    """non-overlapping and dense: its n elements occupy exactly n storage slots (true for channels_last weights)"""
    sizes_strides = sorted(zip(p.stride(), p.size()))
    expect = 1
    for st, sz in sizes_strides:
        if sz == 1:
            continue
        if st != expect:
            return False
        expect *= sz
    return True


def train_step(net, optimizer, batch, grad_sync=None, monitor=False):
    """exec.py:68-74: results = net.train_forward(batch); zero_grad; loss.backward(); optimizer.step()."""
    if isinstance(optimizer, FlatAdam) and optimizer._grad_sync is not grad_sync:
        raise ValueError("train_step: this FlatAdam was built %s, the step was called %s: the optimizer would read a gradient buffer the "
                         "collective never touches" % ("without a grad_sync" if optimizer._grad_sync is None else "over another grad_sync",
                                                       "with one" if grad_sync is not None else "without"))
    if grad_sync is not None and grad_sync.defer_div and not isinstance(optimizer, FlatAdam):
        # after a FlatAdam was built over this sync, finish() leaves the SUM over ranks in the buffer (grad_scale says by what
        # to multiply): any other optimizer would step on gradients world_size times too large
        raise ValueError("train_step: this grad_sync defers the division by the world size to the FlatAdam built over it; "
                         "%s would see the SUM over ranks (grad_sync.grad_scale = 1 / world)" % type(optimizer).__name__)
    results = net.train_forward(batch, monitor=monitor)
    if isinstance(optimizer, FlatAdam):
        optimizer.zero_grad()            # one fill (and the bucket bookkeeping of its grad_sync, if any)
    elif grad_sync is not None:
        grad_sync.zero()
    else:
        optimizer.zero_grad(set_to_none=True)
    results["torch_loss"].backward()
    acc = getattr(net, "_pyramid_grad_acc", None)
    if acc is not None:
        acc.check()                      # every consumer of the pyramid maps handed its gradient to the shared buffers
    if grad_sync is not None:
        grad_sync.finish()
    optimizer.step()
    return results


class GraphedTrainStep(object):
    """exec.py:68-74 with the DEVICE half of the step -- forward, target layer, heads, matching, losses, backward: ~1500 kernel launches
    that cost 38-40 ms of host time against 43 ms of GPU time when issued one by one -- captured once in ONE hipGraph and
    replayed with a single hipGraphLaunch (0.3 ms of host time for the FPN + RPN segment, tools/graph_segment_whole_probe.py).

    What makes the step capturable: fixed-size masked tensors everywhere in the glue (round 2), GT counts read on the device by the
    batched matching kernel (no per-element launches, no host integers in launch parameters), a fixed-size GT table (`gmax` objects
    per element) and a fixed-capacity uint8 mask stack (`max_masks`) as STATIC inputs, the device RNG (torch registers its Philox
    state with the graph), no host synchronisation inside the step.  Per call: the batch is copied into the static inputs (device
    batches: device-to-device; host numpy batches: pinned staging + one asynchronous upload each), the graph is replayed, then --
    eagerly, three launches -- the gradient all-reduce (N > 1; bucketed, after the replay) and the Adam update.
    The eager step (`train_step`) stays the reference implementation: `tests/test_graph_step_gpu.py` requires bit-identical losses
    and gradients from both on a deterministic batch.

    monitor=True: the read-out tensors are packed inside the graph (net.monitor_pack) and leave with ONE device->host copy per step."""

    def __init__(self, net, optimizer, grad_sync=None, gmax=8, max_masks=None, monitor=False, with_masks=False, warmup=2):
        if not hasattr(net, "train_forward_device"):
            raise TypeError("GraphedTrainStep needs a model with the split step (prepare_batch / train_forward_device): models.mrcnn.net")
        if isinstance(optimizer, FlatAdam) and optimizer._grad_sync is not grad_sync:
            raise ValueError("GraphedTrainStep: optimizer and step disagree about the grad_sync")
        self.net, self.opt, self.sync = net, optimizer, grad_sync
        self.gmax, self.monitor, self.with_masks, self.warmup = int(gmax), (monitor if monitor == "deferred" else bool(monitor)), bool(with_masks), int(warmup)
        self._deferred = None
        self.max_masks = max_masks
        self.graph = None
        self.static = None
        self.host_ms = None          # set to a dict to collect host-side wall time per phase of __call__ (diagnosis; bench.py exec leg)

    # -- static inputs
    def _alloc(self, batch):
        net, cf, dev = self.net, self.net.cf, self.net.device_
        B = len(batch["bb_target"])
        shape = tuple(int(v) for v in batch["data"].shape)
        mm = self.max_masks if self.max_masks is not None else B * min(self.gmax, 4)
        self.static = {"img": torch.zeros(shape, dtype=torch.float32, device=dev),
                       "gt": torch.zeros((B, self.gmax, 2 * cf.dim + 4), dtype=torch.float64, device=dev),
                       "masks": torch.zeros((mm, 1) + shape[2:], dtype=torch.uint8, device=dev)}
        self.static["gt"][:, :, 2 * cf.dim + 2] = -1.0

    def _load(self, batch):
        """copy one batch into the static inputs (asynchronous; nothing here waits for the GPU)"""
        from .models.mrcnn import GtOnDevice
        from .utils import model_utils as mutils
        st, cf, dev = self.static, self.net.cf, self.net.device_
        ev = batch.get("ready_event")            # DevicePrefetcher: the uploads ran on its side stream
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for k in ("data", "roi_masks_device"):
                if torch.is_tensor(batch.get(k)):
                    batch[k].record_stream(cur)     # allocated on the side stream, read here: do not recycle before this read ran
        # the small GT table goes through its own persistent pinned ring: a fresh pinned allocation per step (torch's caching host
        # allocator hands a block out again only after its copy has EXECUTED) is a hipHostMalloc -- and an implicit device sync --
        # every step once the host runs ahead of the GPU, which it does by a whole step here (measured: 41 ms of "host issue" per step)
        stage, _, _ = GtOnDevice.stage(batch["bb_target"], batch["roi_labels"], cf.dim, gmax=self.gmax, pin=False)
        st["gt"].copy_(mutils.stage_pinned(stage, dev, "gt_table"), non_blocking=True)
        mutils.stage_release(dev, "gt_table")
        data = batch["data"]
        if torch.is_tensor(data) and data.is_cuda:
            st["img"].copy_(data, non_blocking=True)
        else:
            st["img"].copy_(mutils.stage_pinned(data, dev, "data"), non_blocking=True)
            mutils.stage_release(dev, "data")
        if "roi_masks_device" in batch:
            m = batch["roi_masks_device"]
            n = 0 if m is None else int(m.shape[0])
            if n > st["masks"].shape[0]:
                raise ValueError("the batch holds %d GT masks, the static mask stack %d (GraphedTrainStep(max_masks=...))" % (n, st["masks"].shape[0]))
            if n:
                st["masks"][:n].copy_(m, non_blocking=True)
        else:
            parts = [p for p in batch["roi_masks"] if len(p) > 0]
            n = sum(int(p.shape[0]) for p in parts)
            if n > st["masks"].shape[0]:
                raise ValueError("the batch holds %d GT masks, the static mask stack %d (GraphedTrainStep(max_masks=...))" % (n, st["masks"].shape[0]))
            if n:
                st["masks"][:n].copy_(mutils.stage_pinned(parts, dev, "masks"), non_blocking=True)
                mutils.stage_release(dev, "masks")

    # -- the captured body
    def _body(self):
        from .models.mrcnn import GtOnDevice
        net, st = self.net, self.static
        gt_dev = GtOnDevice(None, None, net.cf.dim, net.device_, table=st["gt"])
        out = net.train_forward_device(st["img"], gt_dev, st["masks"], with_masks=self.with_masks)
        if self.sync is not None:
            self.sync.zero()
        else:
            for p in self._params:
                p.grad = None
        out["loss"].backward()
        acc = getattr(net, "_pyramid_grad_acc", None)
        if acc is not None:
            acc.check()
        packed = net.monitor_pack(out) if self.monitor else None
        return out, packed

    def capture(self, batch):
        """warm up and capture on `batch` (its shapes fix the static inputs); __call__ does this on its first call"""
        import medicaldetectiontoolkit_amd as pkg
        from .cuda_functions import _roi_align_impl
        if not pkg.graph_runtime_safe() and not os.environ.get("MDT_ALLOW_UNSAFE_GRAPH"):
            raise RuntimeError("GraphedTrainStep: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in place before the HIP runtime could initialise (export it, "
                               "or call medicaldetectiontoolkit_amd.graph_env_setup() before `import torch`): with the runtime's graph packet "
                               "capture on, replays of this step fault (see medicaldetectiontoolkit_amd/__init__.py); use training.train_step")
        if _roi_align_impl.PROFILE is not None:
            raise RuntimeError("GraphedTrainStep: switch the RoIAlign event profile off before the capture (events cannot be recorded inside a graph)")
        self._params = [p for p in self.net.parameters() if p.requires_grad]
        # ORDER MATTERS on this stack: capture BEFORE any eager backward of these parameters ran in the process (or after every autograd
        # graph of such a step has died).  A parameter's AccumulateGrad node is created once, on the stream of its first backward, and
        # lives as long as any autograd graph references it; an eager step leaves such graphs behind (the net keeps its last feature maps),
        # the capture's backward then runs those nodes on the DEFAULT stream and hipStreamEndCapture segfaults (bench.py r04: eager steps,
        # then capture).  Dropping the net's references to the last step makes the old nodes die in the common case:
        import gc
        for attr in ("mrcnn_feature_maps", "rpn_feature_maps", "rpn_rois_batch_info", "batch_mrcnn_class_scores"):
            if hasattr(self.net, attr):
                setattr(self.net, attr, None)
        gc.collect()
        if isinstance(self.opt, FlatAdam) and self.opt._flat is None:
            self.opt._build()            # re-homes the parameters: must happen before their addresses are baked into the graph
        prev_suspended = None
        if self.sync is not None:
            if self.sync.flat is None:
                self.sync._build()
            # the hooks fire (in Python) during the capture's backward: they must launch nothing there -- the bucket all-reduces follow
            # the replay (finish_all).  Restored afterwards, also when the capture raises: eager steps keep their overlap.
            prev_suspended, self.sync.suspended = self.sync.suspended, True
        try:
            self._alloc(batch)
            self._load(batch)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):    # warm-up on a side stream (MIOpen find, workspaces, cached constants), as torch's capture protocol asks
                for _ in range(max(1, self.warmup)):
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            from . import _lib
            graph = torch.cuda.CUDAGraph()
            _lib.CAPTURING += 1              # cached device workspaces are bypassed: everything the graph touches lives in its own pool
            try:
                with torch.cuda.graph(graph):
                    self._out, self._packed = self._body()
            finally:
                _lib.CAPTURING -= 1
            self.graph = graph
        finally:
            if self.sync is not None:
                self.sync.suspended = prev_suspended
        self._grads = [p.grad for p in self._params]
        self._pptr = [p.data_ptr() for p in self._params]
        if isinstance(self.opt, FlatAdam):
            self.opt._graph_attached = True      # a re-home of the parameters would leave the graph training orphaned buffers: FlatAdam raises instead
        torch.cuda.synchronize()

    def __call__(self, batch):
        import time
        if self.graph is None:
            self.capture(batch)
        if any(p.data_ptr() != q for p, q in zip(self._params, self._pptr)):
            raise RuntimeError("GraphedTrainStep: a parameter moved since the capture (load_state_dict / .to(...) / another optimizer): the graph would "
                               "keep training the old buffers -- build a new GraphedTrainStep")
        hm = self.host_ms
        from .utils import model_utils as _mu
        w0 = _mu.RING_WAIT_S[0]
        t0 = time.perf_counter()
        self._load(batch)
        t1 = time.perf_counter()
        ring_wait = _mu.RING_WAIT_S[0] - w0
        self.graph.replay()
        t2 = time.perf_counter()
        if self.sync is None:
            for p, g in zip(self._params, self._grads):      # (an eager zero_grad() in between may have dropped them)
                p.grad = g
        else:
            self.sync.finish_all()
        self.opt.step()
        t3 = time.perf_counter()
        out = self._out
        # NOTE: these are the graph's STATIC output tensors -- the next call overwrites them; read (or clone) them before it
        res = {"torch_loss": out["loss"].detach(), "loss_terms": out["terms"], "sample_counts": out["sample_counts"]}
        if self.monitor == "deferred":             # step i's read-out is handed out by call i + 1: no sync, the host never waits for the GPU
            from .utils import model_utils as mu
            if self._deferred is None:
                self._deferred = mu.DeferredReadout()
            prev = self._deferred.push(self._packed, (batch, tuple(self.static["img"].shape)))
            if prev is not None:
                packed, (pbatch, pshape) = mu.DeferredReadout.resolve(prev)
                r = self.net.monitor_results(packed, pbatch, pshape, False, detection_masks=None)
                r["monitor_of_previous_step"] = True
                res.update(r)
        elif self.monitor:
            host = self._packed[0].detach().cpu()                 # the ONE device->host copy (and the one sync) of the step
            t4 = time.perf_counter()
            res.update(self.net.monitor_results((host.numpy(), self._packed[1]), batch, tuple(self.static["img"].shape), False,
                                                detection_masks=None))
            if hm is not None:
                hm["readout_wait"] = hm.get("readout_wait", 0.0) + (t4 - t3) * 1e3
                hm["readout_host"] = hm.get("readout_host", 0.0) + (time.perf_counter() - t4) * 1e3
        if hm is not None:
            hm["load"] = hm.get("load", 0.0) + (t1 - t0 - ring_wait) * 1e3
            hm["ring_wait_backpressure"] = hm.get("ring_wait_backpressure", 0.0) + ring_wait * 1e3
            hm["replay"] = hm.get("replay", 0.0) + (t2 - t1) * 1e3
            hm["collective_adam"] = hm.get("collective_adam", 0.0) + (t3 - t2) * 1e3
            hm["calls"] = hm.get("calls", 0) + 1
        return res


def flush_deferred_monitor(step_or_net):
    """the read-out entries of the LAST step of a monitor="deferred" run (train_step: pass the net; GraphedTrainStep: pass the step)"""
    if isinstance(step_or_net, GraphedTrainStep):
        d = step_or_net._deferred
        entry = d.flush() if d is not None else None
        if entry is None:
            return None
        from .utils import model_utils as mu
        packed, (pbatch, pshape) = mu.DeferredReadout.resolve(entry)
        r = step_or_net.net.monitor_results(packed, pbatch, pshape, False, detection_masks=None)
        r["monitor_of_previous_step"] = True
        return r
    return step_or_net.flush_deferred_monitor()


class DevicePrefetcher(object):
    """Iterator over the reference's batch dicts (host numpy arrays, what batchgenerators' MultiThreadedAugmenter delivers to
    exec.py:67) that hands them out with the bulky entries already in HBM: a background thread copies batch i + 1 into pinned ring
    buffers (GIL-free memcpy) and enqueues its uploads on a SIDE stream while the GPU runs step i, so the PCIe time of
    `torch.FloatTensor(img).cuda()` (mrcnn.py:869) disappears behind the step.  Each batch comes out as a dict with 'data' (device
    fp32), 'roi_masks_device' (stacked uint8 masks), every other entry untouched, and 'ready_event' (recorded on the side stream after
    the uploads; training.GraphedTrainStep and train_step wait for it on the compute stream)."""

    def __init__(self, batches, device, depth=2):
        import queue
        import threading
        self.device = torch.device(device)
        self.q = queue.Queue(maxsize=max(1, int(depth)))
        self.src = iter(batches)
        self.stream = torch.cuda.Stream(device=self.device)
        self._stop = False
        self.thread = threading.Thread(target=self._run, name="mdt-prefetch", daemon=True)
        self.thread.start()

    def _run(self):
        from .utils import model_utils as mutils
        try:
            with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
                for batch in self.src:
                    if self._stop:
                        break
                    out = dict(batch)
                    if not (torch.is_tensor(batch["data"]) and batch["data"].is_cuda):
                        out["data"] = mutils.stage_pinned(batch["data"], self.device, "prefetch_data").to(self.device, non_blocking=True)
                        mutils.stage_release(self.device, "prefetch_data")
                    if "roi_masks_device" not in batch:
                        parts = [m for m in batch["roi_masks"] if len(m) > 0]
                        if parts:
                            out["roi_masks_device"] = mutils.stage_pinned(parts, self.device, "prefetch_masks").to(self.device, non_blocking=True)
                            mutils.stage_release(self.device, "prefetch_masks")
                        else:
                            out["roi_masks_device"] = None
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                    out["ready_event"] = ev
                    self.q.put(out)
            self.q.put(None)
        except BaseException as e:      # hand the failure to the consumer instead of dying silently
            self.q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self):
        self._stop = True
