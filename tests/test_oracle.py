"""Oracle self-consistency (CPU): hand-derived NumPy backward == torch autograd == finite
differences; Adam == torch.optim.Adam; frozen golden fixtures reproduce."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from oracle import elbo_torch as T
from tests.cases import CASES, build

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _inputs(cfg, dk, B, seed=11):
    D, sampler = O.make_dataset(**dk)
    rng = np.random.default_rng(seed)
    p = O.init_params(cfg, seed=3)
    for k in p:
        if not k.endswith("kernel"):
            p[k] = p[k] + 0.2 * rng.standard_normal(p[k].shape)
    x = sampler(rng, B)
    z1, z2 = O.split_latents(rng.standard_normal((B, cfg.L + cfg.D)), cfg.L)
    return p, x, z1, z2


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


@pytest.mark.parametrize("name", list(CASES))
def test_numpy_backward_equals_torch_autograd(name):
    cfg, dk, B, lr = build(name)
    p, x, z1, z2 = _inputs(cfg, dk, B)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    m = T.TorchVAE(cfg, p)
    tl, tdkl, tmse = m.elbo(torch.as_tensor(x), torch.as_tensor(z1), torch.as_tensor(z2))
    tl.backward()
    tg = T.grads_tree(m)
    assert abs(loss - tl.item()) <= 1e-12 * abs(loss)
    ev = O.loss_eval(cfg, p, x, z1, z2)
    assert abs(ev[1] - tdkl.item()) <= 1e-12 * abs(ev[1]) and abs(ev[2] - tmse.item()) <= 1e-12 * abs(ev[2])
    for k in g:
        assert g[k].shape == tg[k].shape
        assert _rel(g[k], tg[k]) <= 1e-11, k


@pytest.mark.parametrize("name", list(CASES))
def test_three_steps_equal_torch_adam(name):
    cfg, dk, B, lr = build(name)
    p, x, z1, z2 = _inputs(cfg, dk, B)
    m = T.TorchVAE(cfg, p)
    opt = T.make_adam(m, lr)
    st = O.adam_init(p)
    rng = np.random.default_rng(5)
    for _ in range(3):
        z1, z2 = O.split_latents(rng.standard_normal((B, cfg.L + cfg.D)), cfg.L)
        p, st, l = O.train_step(cfg, p, st, x, z1, z2, lr)
        tl = T.train_step(m, opt, torch.as_tensor(x), torch.as_tensor(z1), torch.as_tensor(z2))
        assert abs(l - tl.item()) <= 1e-12 * abs(l)
    tp = T.params_tree(m)
    for k in p:
        assert np.max(np.abs(p[k] - tp[k])) <= 1e-12 * max(1.0, np.max(np.abs(tp[k]))), k


@pytest.mark.parametrize("name", ["c1_linear_L20", "c2_sigmoid_mlp", "c3_sphere_mlp"])
def test_finite_differences(name):
    cfg, dk, B, lr = build(name)
    p, x, z1, z2 = _inputs(cfg, dk, B)
    _, g = O.loss_and_grad(cfg, p, x, z1, z2)
    rng = np.random.default_rng(0)
    h = 1e-6
    for k in p:
        for _ in range(2):
            idx = tuple(rng.integers(0, s) for s in p[k].shape)
            pp = {a: b.copy() for a, b in p.items()}
            pm = {a: b.copy() for a, b in p.items()}
            pp[k][idx] += h
            pm[k][idx] -= h
            fd = (O.loss_eval(cfg, pp, x, z1, z2)[0] - O.loss_eval(cfg, pm, x, z1, z2)[0]) / (2 * h)
            assert abs(fd - g[k][idx]) <= 1e-6 * max(1.0, abs(fd)), (k, idx, fd, g[k][idx])


@pytest.mark.parametrize("name", list(CASES))
def test_golden_fixture_reproduces(name):
    """Frozen numbers in tests/golden/*.npz (made by tests/golden/make_golden.py)."""
    cfg, dk, B, lr = build(name)
    f = np.load(os.path.join(GOLD, f"{name}.npz"))
    meta = json.loads(str(f["meta"]))
    assert [n for n, _ in meta["leaves"]] == [n for n, _ in cfg.leaves()]
    p = O.unflatten(cfg, f["params0"])
    loss0, g0 = O.loss_and_grad(cfg, p, f["x"][0], f["z1"][0], f["z2"][0])
    assert abs(loss0 - f["loss0"]) <= 1e-13 * abs(loss0)
    assert _rel(O.flatten(cfg, g0), f["grad0"]) <= 1e-12
    st = O.adam_init(p)
    for s in range(meta["n_steps"]):
        p, st, l = O.train_step(cfg, p, st, f["x"][s], f["z1"][s], f["z2"][s], lr)
        assert abs(l - f["losses"][s]) <= 1e-12 * abs(l)
    assert _rel(O.flatten(cfg, p), f["params_final"]) <= 1e-12
    assert _rel(O.flatten(cfg, st["m"]), f["m_final"]) <= 1e-12
    assert _rel(O.flatten(cfg, st["v"]), f["v_final"]) <= 1e-12


def test_float32_torch_restatement_tracks_oracle():
    """The f32 torch restatement timed as cpu_baseline agrees with the f64 oracle to 1e-5 rel ELBO."""
    cfg, dk, B, lr = build("c1_linear_L20")
    p, x, z1, z2 = _inputs(cfg, dk, 128)
    loss, _ = O.loss_and_grad(cfg, p, x, z1, z2)
    m = T.TorchVAE(cfg, p, dtype=torch.float32)
    f = lambda a: torch.as_tensor(a, dtype=torch.float32)
    tl, _, _ = m.elbo(f(x), f(z1), f(z2))
    assert abs(tl.item() - loss) <= 1e-5 * abs(loss)
