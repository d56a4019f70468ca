"""Deterministic weights / input for the backbone numerics check: every parameter is
filled from an RNG seeded by its state-dict NAME, so the reference FPN (CPU, make_backbone_golden.py) and the FPN of this
repo (MIOpen / CK on the GPU) get identical weights without a checkpoint being stored."""
import zlib

import numpy as np
import torch

PATCH = [64, 64, 32]
KEEP = 12000    # output elements kept per pyramid level (evenly spaced flat indices)


def make_cf(**kw):
    from medicaldetectiontoolkit_amd.configs import Configs
    return Configs(dim=3, model="mrcnn", patch_size=PATCH, batch_size=2, **kw)


def fill_by_name(module):
    with torch.no_grad():
        for name, p in sorted(module.state_dict().items()):
            rng = np.random.default_rng(zlib.crc32(name.encode()))
            shape = tuple(p.shape)
            if len(shape) > 1:
                fan_in = int(np.prod(shape[1:]))
                v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)      # He: keeps activations O(1) through the ResNet
            else:
                v = rng.uniform(-0.05, 0.05, size=shape)
            p.copy_(torch.from_numpy(v.astype(np.float32)))


def make_input():
    rng = np.random.default_rng(21)
    return rng.standard_normal([2, 1] + PATCH).astype(np.float32)


def sample_index(n):
    return np.linspace(0, n - 1, min(n, KEEP)).astype(np.int64)
