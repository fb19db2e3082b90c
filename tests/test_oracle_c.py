"""The C + OpenMP CPU port (oracle/elbo_ref.c: checker and bench.py cpu_baseline) against the float64
NumPy oracle on the same inputs: loss 1e-5 relative, gradients 2e-5 of max-abs, 3 Adam steps."""
import numpy as np
import pytest

from oracle import elbo_oracle as O
from oracle import elbo_ref as R
from tests.cases import CASES, build


@pytest.mark.parametrize("name", list(CASES))
def test_c_port_matches_numpy_oracle(name):
    cfg, dk, B, lr = build(name)
    lib = R.load()
    c = R.make_cfg(cfg.D, cfg.L, cfg.enc_sizes[:-1], cfg.dec_sizes[:-1], cfg.epsilon, cfg.tdv, cfg.sigmoid)
    assert lib.elbo_ref_param_count(c) == cfg.n_params()
    _, sampler = O.make_dataset(**dk)
    rng = np.random.default_rng(5)
    r32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    p = O.init_params(cfg, seed=1)
    for k in p:
        if not k.endswith("kernel"):
            p[k] = p[k] + 0.2 * rng.standard_normal(p[k].shape)
    p = {k: r32(v).astype(np.float64) for k, v in p.items()}
    B = 64
    x = r32(sampler(rng, B)); z = r32(rng.standard_normal((B, cfg.L + cfg.D)))
    z1, z2 = r32(z[:, :cfg.L]), r32(z[:, cfg.L:])
    loss, g = O.loss_and_grad(cfg, p, x.astype(np.float64), z1.astype(np.float64), z2.astype(np.float64))
    P = cfg.n_params()
    params = r32(O.flatten(cfg, p)); m = np.zeros(P, np.float32); v = np.zeros(P, np.float32); grads = np.zeros(P + 4, np.float32)
    for nt in (1, 3):
        l = R.step(lib, c, params, m, v, 1, x, z1, z2, lr, grads=grads, apply=False, nthreads=nt)
        assert abs(l - loss) <= 1e-5 * abs(loss)
        want = O.flatten(cfg, g)
        assert np.max(np.abs(grads[:P] - want)) <= 2e-5 * np.max(np.abs(want))
    st = O.adam_init(p)
    for t in range(1, 4):
        p, st, lo = O.train_step(cfg, p, st, x.astype(np.float64), z1.astype(np.float64), z2.astype(np.float64), lr)
        l = R.step(lib, c, params, m, v, t, x, z1, z2, lr)
        assert abs(l - lo) <= 1e-5 * abs(lo)
    assert np.max(np.abs(params - O.flatten(cfg, p))) <= 0.02 * lr


def test_c_port_shard_divisor():
    cfg = O.Config(12, 20, (), (), -1.0, True, "linear_gaussian")
    lib = R.load()
    c = R.make_cfg(12, 20, (), (), -1.0, True, False)
    rng = np.random.default_rng(0)
    P = cfg.n_params()
    params = O.flatten(cfg, O.init_params(cfg, 0)).astype(np.float32)
    x = rng.standard_normal((64, 12)).astype(np.float32); z1 = rng.standard_normal((64, 20)).astype(np.float32); z2 = rng.standard_normal((64, 12)).astype(np.float32)
    full = np.zeros(P + 4, np.float32); a = np.zeros(P + 4, np.float32); b = np.zeros(P + 4, np.float32); z = np.zeros(P, np.float32)
    R.step(lib, c, params, z, z, 1, x, z1, z2, 0, grads=full, apply=False)
    R.step(lib, c, params, z, z, 1, np.ascontiguousarray(x[:32]), np.ascontiguousarray(z1[:32]), np.ascontiguousarray(z2[:32]), 0, grads=a, apply=False, batch_total=64)
    R.step(lib, c, params, z, z, 1, np.ascontiguousarray(x[32:]), np.ascontiguousarray(z1[32:]), np.ascontiguousarray(z2[32:]), 0, grads=b, apply=False, batch_total=64)
    assert np.max(np.abs(a + b - full)) <= 2e-6 * np.max(np.abs(full))
