"""Generates tests/golden/*.npz from the float64 oracle (oracle/elbo_oracle.py).

These are NOT reference outputs (the reference cannot run here and ships none: parity
unpinned); they freeze the oracle so that accidental edits to it, and the HIP
path's agreement with it, are both checked against committed numbers.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import elbo_oracle as O  # noqa: E402
from tests.cases import CASES, build  # noqa: E402

N_STEPS = 3


def generate(name):
    cfg, dk, B, lr = build(name)
    D, sampler = O.make_dataset(**dk)
    assert D == cfg.D
    rng = np.random.default_rng(1)
    p = O.init_params(cfg, seed=0)
    # perturb biases / epsilon_p / epsilon off their trivial inits so every term is exercised
    prng = np.random.default_rng(7)
    for k in p:
        if not k.endswith("kernel"):
            p[k] = p[k] + 0.1 * prng.standard_normal(p[k].shape)
    xs, z1s, z2s = [], [], []
    for _ in range(N_STEPS):
        xs.append(sampler(rng, B))
        z = rng.standard_normal((B, cfg.L + cfg.D))
        z1, z2 = O.split_latents(z, cfg.L)
        z1s.append(z1.copy()); z2s.append(z2.copy())
    out = {"params0": O.flatten(cfg, p), "x": np.stack(xs), "z1": np.stack(z1s), "z2": np.stack(z2s)}
    loss0, g0 = O.loss_and_grad(cfg, p, xs[0], z1s[0], z2s[0])
    ev = O.loss_eval(cfg, p, xs[0], z1s[0], z2s[0])
    out["loss0"] = np.float64(loss0)
    out["grad0"] = O.flatten(cfg, g0)
    out["eval0"] = np.array([ev[0], ev[1], ev[2]])
    st = O.adam_init(p)
    losses = []
    for s in range(N_STEPS):
        p, st, l = O.train_step(cfg, p, st, xs[s], z1s[s], z2s[s], lr)
        losses.append(l)
    out["losses"] = np.array(losses)
    out["params_final"] = O.flatten(cfg, p)
    out["m_final"] = O.flatten(cfg, st["m"])
    out["v_final"] = O.flatten(cfg, st["v"])
    meta = dict(case=name, cfg=CASES[name][0], dataset=dk, B=B, lr=lr, n_steps=N_STEPS,
                leaves=[[n, list(s)] for n, s in cfg.leaves()])
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
    return out


if __name__ == "__main__":
    for name in CASES:
        o = generate(name)
        print(name, "loss0 =", float(o["loss0"]), "P =", o["params0"].size)
