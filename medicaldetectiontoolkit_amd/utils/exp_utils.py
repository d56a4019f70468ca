"""Checkpoint format of the reference (utils/exp_utils.py:147-204), rank-0-only under data parallelism.

last_checkpoint/params.pth = {'epoch', 'state_dict', 'optimizer'} + monitor_metrics.pickle (+ epoch_ranking.npy);
<epoch>_best_checkpoint/params.pth = bare state_dict.  Module names are the reference's, so files are interchangeable.
"""
import os
import pickle

import numpy as np
import torch
import torch.distributed as dist


def _is_rank0():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_last_checkpoint(fold_dir, net, optimizer, epoch, monitor_metrics=None, epoch_ranking=None):
    """exp_utils.py:178-192.  Only rank 0 writes; every rank returns after a barrier."""
    if _is_rank0():
        save_dir = os.path.join(fold_dir, "last_checkpoint")
        os.makedirs(save_dir, exist_ok=True)
        state = {"epoch": epoch, "state_dict": net.state_dict(), "optimizer": optimizer.state_dict()}
        tmp = os.path.join(save_dir, "params.pth.tmp")
        torch.save(state, tmp)
        os.replace(tmp, os.path.join(save_dir, "params.pth"))
        np.save(os.path.join(save_dir, "epoch_ranking"), np.asarray(epoch_ranking if epoch_ranking is not None else [epoch]))
        with open(os.path.join(save_dir, "monitor_metrics.pickle"), "wb") as handle:
            pickle.dump(monitor_metrics if monitor_metrics is not None else {}, handle)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def save_best_checkpoint(fold_dir, net, epoch, monitor_metrics=None):
    """exp_utils.py:164-169: bare state_dict under '<epoch>_best_checkpoint'."""
    if _is_rank0():
        save_dir = os.path.join(fold_dir, "{}_best_checkpoint".format(epoch))
        os.makedirs(save_dir, exist_ok=True)
        torch.save(net.state_dict(), os.path.join(save_dir, "params.pth"))
        with open(os.path.join(save_dir, "monitor_metrics.pickle"), "wb") as handle:
            pickle.dump(monitor_metrics if monitor_metrics is not None else {}, handle)


def load_checkpoint(checkpoint_path, net, optimizer=None, map_location=None):
    """exp_utils.py:196-204.  Accepts both the {'epoch','state_dict','optimizer'} form and a bare state_dict.
    Returns (starting_epoch, monitor_metrics)."""
    params = torch.load(os.path.join(checkpoint_path, "params.pth"), map_location=map_location)
    if isinstance(params, dict) and "state_dict" in params:
        net.load_state_dict(params["state_dict"])
        if optimizer is not None and "optimizer" in params:
            optimizer.load_state_dict(params["optimizer"])
        starting_epoch = params.get("epoch", 0) + 1
    else:
        net.load_state_dict(params)
        starting_epoch = 1
    metrics = {}
    mm = os.path.join(checkpoint_path, "monitor_metrics.pickle")
    if os.path.exists(mm):
        with open(mm, "rb") as handle:
            metrics = pickle.load(handle)
    return starting_epoch, metrics
