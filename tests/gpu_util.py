"""Helpers shared by the -m gpu parity tests (HIP path through the C ABI vs the oracle)."""
import json
import os

import numpy as np
import torch

from oracle import elbo_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    f = np.load(os.path.join(GOLD, f"{name}.npz"))
    return f, json.loads(str(f["meta"]))


def engine_for(cfg, B, **kw):
    from vae_training_amd.engine import Engine
    return Engine(B, cfg.D, cfg.L, cfg.enc_sizes[:-1], cfg.dec_sizes[:-1], cfg.epsilon, cfg.tdv, cfg.sigmoid, **kw)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda().contiguous()


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def check_layout(eng, cfg):
    """The library's flat leaf order must equal the oracle's (include/vaek.h conventions)."""
    want = cfg.leaves()
    got = list(eng.leaves.items())
    assert [n for n, _ in want] == [n for n, _ in got]
    off = 0
    for (n, shape), (_, (o, s)) in zip(want, got):
        assert tuple(shape) == tuple(s) and o == off, (n, shape, s, o, off)
        off += int(np.prod(shape))
    assert off == eng.P == cfg.n_params()


def rel_err(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def random_problem(cfg, dk, B, seed=11):
    D, sampler = O.make_dataset(**dk)
    rng = np.random.default_rng(seed)
    p = O.init_params(cfg, seed=3)
    for k in p:
        if not k.endswith("kernel"):
            p[k] = p[k] + 0.2 * rng.standard_normal(p[k].shape)
    # round inputs to float32 so both sides see identical numbers
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    x = sampler(rng, B).astype(np.float32).astype(np.float64)
    z = rng.standard_normal((B, cfg.L + cfg.D)).astype(np.float32).astype(np.float64)
    z1, z2 = O.split_latents(z, cfg.L)
    return p, x, np.ascontiguousarray(z1), np.ascontiguousarray(z2)
