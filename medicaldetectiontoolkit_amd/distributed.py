"""Patch-level data parallelism helpers: one process per GPU, torch.distributed
(backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

Training: independent patches per rank, ONE flat gradient all-reduce per step (training.FlatGradAllReduce).
Inference: the patch list of a patient is sharded round-robin over ranks (patches are independent forwards);
the one exchange step is an all_gather of the per-rank detection tables before weighted box clustering."""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_items, rank=None, world_size=None):
    """Round-robin shard of range(n_items): rank r owns r, r + W, r + 2W, ...  (patch list of one patient)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_items, world_size))


def gather_rows(local):
    """all_gather of row tables with different row counts: local [n_r, k] -> [sum n_r, k] (rank order) on every
    rank.  Counts are exchanged first, rows are padded to the maximum so ONE fixed-size collective suffices."""
    rank, w = world()
    if w == 1:
        return local
    dev = local.device
    k = local.shape[1]
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(w)]
    dist.all_gather(cnts, cnt)
    cnts = [int(c.item()) for c in cnts]
    m = max(max(cnts), 1)
    pad = torch.zeros((m, k), dtype=local.dtype, device=dev)
    pad[:local.shape[0]] = local
    bufs = [torch.zeros_like(pad) for _ in range(w)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, cnts)], 0)
