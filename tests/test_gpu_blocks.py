"""-m gpu: the building-block entry points of include/vaek.h against float64 NumPy."""
import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.gpu_util import dev, host, rel_err

pytestmark = pytest.mark.gpu


def _eng(B=1024, D=64, L=32, hidden=(96,)):
    from vae_training_amd.engine import Engine
    return Engine(B, D, L, hidden, hidden, -1.0, True, False)


@pytest.mark.parametrize("rows,n_in,n_out", [(1, 1, 1), (7, 3, 5), (64, 64, 64), (65, 17, 33), (300, 12, 20),
                                             (1000, 96, 64), (129, 130, 70), (512, 7, 256), (256, 256, 6)])
@pytest.mark.parametrize("relu", [False, True])
def test_dense_fwd_bwd(rows, n_in, n_out, relu):
    eng = _eng()
    rng = np.random.default_rng(rows * 131 + n_in * 7 + n_out)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    x = f32(rng.standard_normal((rows, n_in))); w = f32(rng.standard_normal((n_in, n_out)) / np.sqrt(n_in))
    b = f32(rng.standard_normal(n_out)); dy = f32(rng.standard_normal((rows, n_out)))
    y = x @ w + b
    if relu:
        y = np.maximum(y, 0)
    gy = eng.dense_fwd(dev(x), dev(w), dev(b), relu=relu)
    assert rel_err(host(gy), y) <= 2e-6
    # dX with the relu mask of the INPUT activation
    xpost = np.maximum(x, 0) if relu else x
    dx = dy @ w.T
    if relu:
        dx = dx * (xpost > 0)
    gdx = eng.dense_bwd_dx(dev(dy), dev(w), dev(xpost) if relu else None, relu=relu)
    assert rel_err(host(gdx), dx) <= 2e-6
    gdx2 = eng.dense_bwd_dx(dev(dy), dev(w), dev(xpost) if relu else None, relu=relu, out=gdx.clone(), accumulate=True)
    assert rel_err(host(gdx2), 2 * dx) <= 2e-6
    # dW | db
    if (n_in + 1) * n_out <= eng.P:
        dwb = np.concatenate([x.T @ dy, dy.sum(0, keepdims=True)], axis=0)
        gdwb = eng.dense_bwd_dw(dev(x), dev(dy))
        assert rel_err(host(gdwb), dwb) <= 5e-6


@pytest.mark.parametrize("rows,n_wide,n_thin", [(8192, 1024, 20), (8211, 2048, 32), (9001, 1152, 7), (16384, 4096, 20)])
def test_tall_skinny_streaming_dense(rows, n_wide, n_thin):
    """The long-reduction / skinny-output shapes (C4's 4096 -> 20 encoder forward with the reparameterisation, and the decoder's input
    gradient) run on ts_gemm_kernel (gemm_f32.hip): four waves split K, partial tiles summed in a fixed order.  Ragged row counts,
    thin sides off the 4-grid, accumulate."""
    eng = _eng()
    rng = np.random.default_rng(rows + n_wide + n_thin)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    x = f32(rng.standard_normal((rows, n_wide))); w = f32(rng.standard_normal((n_wide, n_thin)) / np.sqrt(n_wide))
    b = f32(rng.standard_normal(n_thin)); z1 = f32(rng.standard_normal((rows, n_thin))); lv = f32(0.3 * rng.standard_normal(n_thin))
    mu, samples = eng.dense_fwd_reparam(dev(x), dev(w), dev(b), dev(z1), dev(lv))
    want_mu = x @ w + b
    assert rel_err(host(mu), want_mu) <= 2e-6 and rel_err(host(samples), want_mu + np.exp(lv / 2) * z1) <= 2e-6
    mu2, _ = eng.dense_fwd_reparam(dev(x), dev(w), dev(b), dev(z1), dev(lv))
    assert torch.equal(mu, mu2)                                   # fixed-order sums: bitwise repeatable
    wd = f32(rng.standard_normal((n_thin, n_wide)) / np.sqrt(n_wide))      # a decoder kernel [thin, wide]; dy [rows, wide]
    dx = x @ wd.T
    gdx = eng.dense_bwd_dx(dev(x), dev(wd))
    assert rel_err(host(gdx), dx) <= 2e-6
    gdx2 = eng.dense_bwd_dx(dev(x), dev(wd), out=gdx.clone(), accumulate=True)
    assert rel_err(host(gdx2), 2 * dx) <= 2e-6


@pytest.mark.parametrize("sig", [False, True])
@pytest.mark.parametrize("rows,D,L", [(5, 3, 2), (256, 12, 20), (1000, 7, 6), (777, 33, 9)])
def test_elbo_block(rows, D, L, sig):
    eng = _eng(B=1024)
    rng = np.random.default_rng(rows + D + L)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    x = f32(rng.standard_normal((rows, D))); ylin = f32(rng.standard_normal((rows, D)))
    ysig = f32(rng.standard_normal((rows, D))) if sig else None
    z2 = f32(rng.standard_normal((rows, D))); mu = f32(rng.standard_normal((rows, L)))
    lv = f32(rng.standard_normal(L)); eps = -1.25
    sg = 1 / (1 + np.exp(-ysig)) if sig else 0.0
    xh = ylin + sg + z2 * np.exp(eps / 2)
    loss, dkl, mse = O.elbo_terms(x, xh, mu, lv, eps)
    r = xh - x
    d = r * np.exp(-eps) / rows
    deps = (-0.5 * r * r * np.exp(-eps) + 0.5 + 0.5 * np.exp(eps / 2) * z2 * r * np.exp(-eps)).sum() / rows
    out4, dl, ds = eng.elbo_fwd_bwd(dev(x), dev(ylin), dev(ysig) if sig else None, dev(z2), dev(mu), dev(lv), eps)
    o = host(out4)
    assert abs(o[0] - loss) <= 1e-5 * abs(loss) and abs(o[1] - dkl) <= 1e-5 * abs(loss) and abs(o[2] - mse) <= 1e-5 * abs(loss)
    assert abs(o[3] - deps) <= 1e-5 * max(1.0, abs(deps))
    assert rel_err(host(dl), d) <= 2e-6
    if sig:
        assert rel_err(host(ds), d * sg * (1 - sg)) <= 2e-6


@pytest.mark.parametrize("n", [1, 533, 100000])
def test_adam_block(n):
    eng = _eng()
    rng = np.random.default_rng(n)
    p = {"w": rng.standard_normal(n).astype(np.float32).astype(np.float64)}
    st = O.adam_init(p)
    gp, gm, gv = dev(p["w"]), dev(np.zeros(n)), dev(np.zeros(n))
    step_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    for t in range(1, 6):
        g = {"w": rng.standard_normal(n).astype(np.float32).astype(np.float64) * (10.0 ** rng.integers(-6, 2))}
        p, st = O.adam_update(p, g, st, 1e-3)
        if t % 2:
            eng.adam_step(gp, dev(g["w"]), gm, gv, 1e-3, step=t)
        else:
            step_dev.fill_(t)
            eng.adam_step(gp, dev(g["w"]), gm, gv, 1e-3, step_dev=step_dev)
        assert np.max(np.abs(host(gp) - p["w"])) <= 2e-6
    assert rel_err(host(gm), st["m"]["w"]) <= 5e-6 and rel_err(host(gv), st["v"]["w"]) <= 5e-6


def test_adam_bias_correction_long_run():
    """t up to 150000 (sphere/sigmoid scripts): 1 - beta^t in-kernel stays accurate."""
    eng = _eng()
    for t in (1, 2, 10, 1000, 100000, 150000):
        g = np.array([0.5, -2.0, 1e-3])
        gp, gm, gv = dev(np.zeros(3)), dev(np.array([0.1, -0.2, 0.3])), dev(np.array([0.01, 0.02, 0.03]))
        m = 0.9 * np.array([0.1, -0.2, 0.3]) + 0.1 * g
        v = 0.999 * np.array([0.01, 0.02, 0.03]) + 0.001 * g * g
        want = -1e-3 * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) + 1e-8)
        eng.adam_step(gp, dev(g), gm, gv, 1e-3, step=t)
        assert np.max(np.abs(host(gp) - want)) <= 2e-6 * np.max(np.abs(want)) + 1e-9, t
