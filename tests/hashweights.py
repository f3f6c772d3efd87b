"""Deterministic, RNG-free parameter fill used by the golden fixtures.

Every tensor element is a pure integer hash of (state_dict key, flat index), so
the same weights can be regenerated anywhere (no torch RNG / version
dependence) and loaded both into the reference classes (at fixture-generation
time, tests/golden/gen_golden.py) and into pdgn_amd / oracle modules (at test
time).
"""
import numpy as np
import torch


def _fnv1a(s):
    h = 0x811C9DC5
    for ch in s.encode():
        h = ((h ^ ch) * 0x01000193) & 0xFFFFFFFF
    return h


def unit_hash(key, n, salt=0):
    """n floats in [-1, 1), a function of (key, index, salt) only."""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(0x9E3779B1) + np.uint64(_fnv1a(key) ^ (salt * 0x85EBCA6B & 0xFFFFFFFF))) \
        & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return (h.astype(np.float64) / 2.0 ** 31 - 1.0).astype(np.float32)


def hash_tensor(key, shape, scale=1.0, offset=0.0, salt=0):
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy(unit_hash(key, n, salt) * np.float32(scale) + np.float32(offset)).view(*shape)


@torch.no_grad()
def fill_module(module, salt=0, gain=1.7):
    """Fill every parameter / buffer of `module` from the hash rule.

    conv / linear weights: uniform(-1,1) * gain / sqrt(fan_in); biases: 0.1*u;
    BatchNorm weight: 1 + 0.2*u, bias 0.1*u; running_mean 0, running_var 1.
    """
    sd = module.state_dict()
    for key, t in sd.items():
        if key.endswith("num_batches_tracked"):
            t.zero_()
        elif key.endswith("running_mean"):
            t.zero_()
        elif key.endswith("running_var"):
            t.fill_(1.0)
        elif t.dim() >= 2:
            fan_in = int(np.prod(t.shape[1:]))
            t.copy_(hash_tensor(key, t.shape, gain / np.sqrt(fan_in), salt=salt))
        else:
            is_bn_w = key.endswith("weight")          # 1-D weight => a BatchNorm scale
            t.copy_(hash_tensor(key, t.shape, 0.2 if is_bn_w else 0.1,
                                1.0 if is_bn_w else 0.0, salt=salt))
    return module


def lattice_points(key, shape, bits=8, salt=0):
    """Coordinates on a 2^-bits grid in [-1, 1): squared distances are exact in fp32
    under any summation order / FMA contraction (<= 2*bits+4 significant bits)."""
    u = unit_hash(key, int(np.prod(shape)), salt).astype(np.float64)
    q = np.floor(u * 2 ** bits) / 2 ** bits
    return q.astype(np.float32).reshape(shape)
