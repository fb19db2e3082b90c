"""Tiny stand-in for the jax.random calls on the hot path (model.py:29,227; vae.py:56,125).

jax's threefry streams cannot be reproduced without JAX: what is kept is the
interface -- explicit, splittable keys -- and the distributions.  A key is a pair of 64-bit
integers; `split` derives children with SplitMix64; draws happen ON THE DEVICE through a
torch.Generator seeded from the key (Philox on ROCm)."""
from __future__ import annotations

import torch

_MASK = (1 << 64) - 1


def _mix(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


class Key(tuple):
    """(hi, lo) 64-bit words."""


def PRNGKey(seed: int) -> Key:
    return Key((0, int(seed) & _MASK))


def split(key: Key, num: int = 2):
    hi, lo = key
    base = _mix(hi ^ _mix(lo))
    return tuple(Key((_mix(base + 2 * i + 1), _mix(base + 2 * i + 2))) for i in range(num))


def _generator(key: Key, device):
    g = torch.Generator(device=device)
    g.manual_seed((key[0] ^ _mix(key[1])) & ((1 << 63) - 1))
    return g


def normal(key: Key, shape, device="cuda", dtype=torch.float32):
    return torch.randn(*shape, generator=_generator(key, device), device=device, dtype=dtype)


def truncated_normal(key: Key, lower, upper, shape, device="cpu", dtype=torch.float32):
    out = torch.empty(*shape, device=device, dtype=dtype)
    torch.nn.init.trunc_normal_(out, 0.0, 1.0, lower, upper, generator=_generator(key, device))
    return out
