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
    """Averages gradients across ranks with ONE collective per step.  Parameters without a gradient (the
    reference FPN always constructs P1_conv1 / P1_conv2 but Mask R-CNN never uses them, backbone.py:112,118;
    heads may see no positive RoI on a rank) contribute zeros, so every rank issues the same collective --
    no find_unused_parameters machinery and no rank-dependent hangs."""

    def __init__(self, net):
        self.params = [p for p in net.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def __call__(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


def train_step(net, optimizer, batch, grad_sync=None, monitor=False):
    """exec.py:68-74: results = net.train_forward(batch); zero_grad; loss.backward(); optimizer.step()."""
    results = net.train_forward(batch, monitor=monitor)
    optimizer.zero_grad(set_to_none=True)
    results["torch_loss"].backward()
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    return results
