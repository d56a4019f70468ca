"""Training step driver: the reference's exec.py:67-79 hot loop (forward, zero_grad, backward, Adam step),
plus patch-level data parallelism -- one process per GPU, gradients averaged with a single flat-bucket
all-reduce over RCCL (the whole model is 4.94 M fp32 parameters = 19.75 MB)."""
import torch
import torch.distributed as dist


def build_optimizer(net, cf, fused=False, flat=False, grad_sync=None):
    """exec.py:39: Adam(lr=cf.learning_rate[0], weight_decay=cf.weight_decay); fused=True uses torch's single-kernel
    multi-tensor implementation (same update rule); flat=True: FlatAdam (one launch of csrc/adam.hip over flat buffers)."""
    if flat:
        return FlatAdam(net.parameters(), lr=cf.learning_rate[0], weight_decay=cf.weight_decay, grad_sync=grad_sync)
    kw = {"fused": True} if fused else {}
    return torch.optim.Adam(net.parameters(), lr=cf.learning_rate[0], weight_decay=cf.weight_decay, **kw)


def _view_like(flat, off, p):
    """the n = p.numel() slots of `flat` from `off` on, seen with p's own shape AND strides (channels_last weights are dense,
    permuted views: element k of the slice is the same storage slot k of p)"""
    chunk = flat[off:off + p.numel()]
    return chunk.as_strided(p.size(), p.stride()) if p.is_contiguous() is False and _dense(p) else chunk.view_as(p)


class FlatAdam(torch.optim.Adam):
    """torch.optim.Adam (exec.py:39) with the whole model in four flat fp32 buffers -- parameters, gradients, exp_avg, exp_avg_sq
    (4.94 M values = 19.75 MB each) -- and the update as ONE launch of mdt_adam_flat (csrc/adam.hip) instead of ~20 multi-tensor
    launches and 3.4 ms of host time per step (profiles/r03_host_profile.txt).

    * every `p.data`, `p.grad`, `state[p]['exp_avg']`, `state[p]['exp_avg_sq']` is a VIEW (with p's strides) into the flat buffers,
      in reverse parameter order (= FlatGradAllReduce's order, whose gradient buffer is adopted when one is passed), so
      `state_dict()` / `load_state_dict()` keep torch.optim.Adam's format (checkpoints of the reference's Adam load unchanged);
    * gradients: with a FlatGradAllReduce (N > 1) they LIVE in its flat buffer (`p.grad` views; `zero_grad()` = one fill).  Without
      one, `zero_grad()` sets them to None so that autograd hands its gradient tensors over without an accumulate-add per
      parameter (130 tiny launches per step), and `step()` gathers them into the flat buffer with one `torch._foreach_copy_`;
    * a parameter that received no gradient in a step sees a ZERO gradient (torch skips it): identical while it never had one
      (the unused P1 convolutions of the reference FPN, backbone.py:112,118) and for weight_decay = 0 otherwise up to the decay of its
      moments -- the same convention the multi-GPU path has (FlatGradAllReduce).  One step counter for all parameters.
    * amsgrad / maximize / capturable / differentiable are not supported (the reference never sets them)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, grad_sync=None):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        if len(self.param_groups) != 1:
            raise ValueError("FlatAdam: one parameter group (the reference's exec.py:39 passes net.parameters())")
        self._grad_sync = grad_sync
        self._flat = None           # (param, grad, exp_avg, exp_avg_sq)
        self._steps = 0

    def _build(self):
        params = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if not params:
            raise ValueError("FlatAdam: no trainable parameter")
        dev = params[0].device
        if any(p.device != dev or p.dtype != torch.float32 or not (p.is_contiguous() or _dense(p)) for p in params):
            raise ValueError("FlatAdam: parameters must be dense fp32 tensors on one device")
        n = sum(p.numel() for p in params)
        gs = self._grad_sync
        if gs is not None:
            if gs.flat is None:
                gs._build()
            if len(gs.params) != len(params) or any(a is not b for a, b in zip(gs.params, params)):
                raise ValueError("FlatAdam: grad_sync was built over a different parameter list")
            fgrad = gs.flat
        else:
            fgrad = torch.zeros(n, dtype=torch.float32, device=dev)
        fparam = torch.empty(n, dtype=torch.float32, device=dev)
        fm = torch.zeros(n, dtype=torch.float32, device=dev)
        fv = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        gviews = []
        with torch.no_grad():
            for p in reversed(params):                       # backward order, as FlatGradAllReduce lays the gradients out
                v = _view_like(fparam, off, p)
                v.copy_(p)
                old_state = self.state.get(p, {})
                gviews.append(_view_like(fgrad, off, p))
                p.data = v
                st = {"step": torch.tensor(float(self._steps)), "exp_avg": _view_like(fm, off, p), "exp_avg_sq": _view_like(fv, off, p)}
                if "exp_avg" in old_state:                    # state loaded (or stepped by plain Adam) before the first flat step
                    st["exp_avg"].copy_(old_state["exp_avg"])
                    st["exp_avg_sq"].copy_(old_state["exp_avg_sq"])
                    st["step"] = torch.tensor(float(old_state["step"]))
                    self._steps = max(self._steps, int(float(old_state["step"])))
                self.state[p] = st
                off += p.numel()
        self._params = params
        self._rparams = list(reversed(params))
        self._gviews = gviews                       # flat-gradient views, in `_rparams` order
        self._gdirty = [False] * len(gviews)        # view holds a gradient of an earlier step
        self._flat = (fparam, fgrad, fm, fv)

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
        self._flat = None           # re-adopt the loaded moments at the next step (`_build` copies them into the flat buffers)
        self._steps = 0

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
        fparam, fgrad, fm, fv = self._flat
        if self._grad_sync is not None:
            for p in self._params:       # a dropped view (someone's zero_grad(set_to_none=True)) would silently freeze p
                if p.grad is None:
                    raise RuntimeError("FlatAdam: a gradient view was dropped; clear gradients with FlatAdam.zero_grad()")
        else:                            # gather autograd's gradient tensors into the flat buffer: one multi-tensor copy
            dst, src = [], []
            for i, p in enumerate(self._rparams):
                g_i = p.grad
                if g_i is None:
                    if self._gdirty[i]:              # last step's gradient must not be applied again
                        self._gviews[i].zero_()
                        self._gdirty[i] = False
                elif g_i.data_ptr() != self._gviews[i].data_ptr():
                    dst.append(self._gviews[i])
                    src.append(g_i)
                    self._gdirty[i] = True
            if dst:
                torch._foreach_copy_(dst, src)
        self._steps += 1
        self._update(fparam, fgrad, fm, fv, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                     float(g["weight_decay"]), self._steps)
        for p in self._params:
            self.state[p]["step"].fill_(float(self._steps))   # CPU scalars (torch.optim.Adam's own format)
        return loss

    def _update(self, fparam, fgrad, fm, fv, lr, beta1, beta2, eps, weight_decay, step):
        """the one launch (csrc/adam.hip); no CPU implementation in the product -- the gloo tests substitute this method"""
        from . import _lib
        with torch.cuda.device(fparam.device):
            rc = _lib.lib().mdt_adam_flat(fparam.data_ptr(), fgrad.data_ptr(), fm.data_ptr(), fv.data_ptr(), fparam.numel(), lr, beta1, beta2,
                                          eps, weight_decay, step, _lib.raw_stream(fparam))
        _lib.check(rc, "mdt_adam_flat")


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

    # -- setup (lazy: parameters must already live on their device)
    def _build(self):
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        order = list(reversed(self.params))                      # backward order
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
                self.bucket_range.append((start, off))
                self.bucket_left0.append(count)
                start, count, b = off, 0, b + 1
        self._left = list(self.bucket_left0)
        self._next = 0
        if self.overlap:
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
        if not self._active():
            return
        b = self.bucket_of[p]
        self._left[b] -= 1
        self._launch_ready()

    # -- per step
    def zero(self):
        """replaces optimizer.zero_grad(): all gradients are views of the flat buffer"""
        if self.flat is None:
            self._build()
        self.flat.zero_()
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
        self.flat.div_(dist.get_world_size())

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
    results = net.train_forward(batch, monitor=monitor)
    if isinstance(optimizer, FlatAdam):
        optimizer.zero_grad()            # one fill (and the bucket bookkeeping of its grad_sync, if any)
    elif grad_sync is not None:
        grad_sync.zero()
    else:
        optimizer.zero_grad(set_to_none=True)
    results["torch_loss"].backward()
    if grad_sync is not None:
        grad_sync.finish()
    optimizer.step()
    return results
