"""Training step driver: the reference's exec.py:67-79 hot loop (forward, zero_grad, backward, Adam step),
plus patch-level data parallelism -- one process per GPU, gradients averaged with a single flat-bucket
all-reduce over RCCL (the whole model is 4.94 M fp32 parameters = 19.75 MB)."""
import torch
import torch.distributed as dist


def build_optimizer(net, cf, fused=False):
    """exec.py:39: Adam(lr=cf.learning_rate[0], weight_decay=cf.weight_decay); fused=True uses torch's single-kernel
    multi-tensor implementation (same update rule)."""
    kw = {"fused": True} if fused else {}
    return torch.optim.Adam(net.parameters(), lr=cf.learning_rate[0], weight_decay=cf.weight_decay, **kw)


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
    if grad_sync is not None:
        grad_sync.zero()
    else:
        optimizer.zero_grad(set_to_none=True)
    results["torch_loss"].backward()
    if grad_sync is not None:
        grad_sync.finish()
    optimizer.step()
    return results
