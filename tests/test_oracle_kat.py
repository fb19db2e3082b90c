"""Closed-form known-answer tests for the oracle."""
import math

import numpy as np

from oracle import elbo_oracle as O


def _zero_nets(cfg, p):
    for k in p:
        if "FC" in k:
            p[k] = np.zeros_like(p[k])
    return p


def _cfg(**kw):
    base = dict(data_dim=12, latent_dim=20, epsilon=-1.0, tunable_decoder_var=True, dataset_name="linear_gaussian")
    base.update(kw)
    return O.Config(**base)


def test_kat_kl():
    # Encoder 0 -> mu = 0; epsilon_p = 1 (its init) -> Dkl_b = 0.5 L (e - 2)
    cfg = _cfg()
    p = _zero_nets(cfg, O.init_params(cfg))
    rng = np.random.default_rng(0)
    x = rng.standard_normal((8, 12)); z1 = rng.standard_normal((8, 20)); z2 = rng.standard_normal((8, 12))
    _, dkl, _, _, _ = O.loss_eval(cfg, p, x, z1, z2)
    assert abs(dkl - 0.5 * 20 * (math.e - 2.0)) < 1e-12
    assert abs(0.5 * 20 * (math.e - 2.0) - 7.18281828459045) < 1e-12


def test_kat_rec():
    # decoder 0, z2 = 0 -> mse_b = 0.5 e^-eps |x|^2 + 0.5 D (log 2pi + eps)
    cfg = _cfg()
    p = _zero_nets(cfg, O.init_params(cfg))
    rng = np.random.default_rng(1)
    x = rng.standard_normal((8, 12)); z1 = rng.standard_normal((8, 20)); z2 = np.zeros((8, 12))
    _, _, mse, _, eps = O.loss_eval(cfg, p, x, z1, z2)
    assert eps.shape == (1,) and eps[0] == -1.0
    want = np.mean(0.5 * math.e * np.sum(x * x, axis=1) + 0.5 * 12 * (O.LOG_2PI - 1.0))
    assert abs(mse - want) < 1e-12 * abs(want)


def test_kat_noise():
    # x = 0, decoder 0, z2 != 0 -> (x_hat - x)^2 / e^eps = z2^2 exactly
    for eps_cli in (-3.0, -1.0, 0.5):
        cfg = _cfg(epsilon=eps_cli)
        p = _zero_nets(cfg, O.init_params(cfg))
        rng = np.random.default_rng(2)
        x = np.zeros((8, 12)); z1 = rng.standard_normal((8, 20)); z2 = rng.standard_normal((8, 12))
        _, _, mse, _, _ = O.loss_eval(cfg, p, x, z1, z2)
        want = np.mean(0.5 * np.sum(z2 * z2, axis=1) + 0.5 * 12 * (O.LOG_2PI + eps_cli))
        assert abs(mse - want) < 1e-12 * abs(want)


def test_kat_adam_first_step():
    # after step 1 every param moves by -lr * g / (|g| + 1e-8)
    cfg = _cfg()
    p = O.init_params(cfg)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((8, 12)); z1 = rng.standard_normal((8, 20)); z2 = rng.standard_normal((8, 12))
    _, g = O.loss_and_grad(cfg, p, x, z1, z2)
    p2, st, _ = O.train_step(cfg, p, O.adam_init(p), x, z1, z2, 1e-3)
    assert st["step"] == 1
    for k in p:
        want = p[k] - 1e-3 * g[k] / (np.abs(g[k]) + 1e-8)
        assert np.max(np.abs(p2[k] - want)) < 1e-12


def test_kat_dp_shards_sum_to_full_batch():
    # mean of W shard gradients (each a shard mean) == full-batch gradient; equivalently the
    # SUM of shard gradients computed with the global divisor
    cfg = O.Config(7, 6, (16,), (16,), -3.0, True, "sigmoid")
    p = O.init_params(cfg, seed=4)
    rng = np.random.default_rng(4)
    B, W = 64, 8
    x = rng.standard_normal((B, 7)); z1 = rng.standard_normal((B, 6)); z2 = rng.standard_normal((B, 7))
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    acc, lacc = None, 0.0
    for w in range(W):
        s = slice(w * B // W, (w + 1) * B // W)
        lw, gw = O.loss_and_grad(cfg, p, x[s], z1[s], z2[s], batch_total=B)
        lacc += lw
        acc = gw if acc is None else {k: acc[k] + gw[k] for k in gw}
    assert abs(lacc - loss) < 1e-12 * abs(loss)
    for k in g:
        assert np.max(np.abs(acc[k] - g[k])) < 1e-12 * max(1.0, np.max(np.abs(g[k])))


def test_kat_sampling_mode():
    # sampling=True -> samples == z1 (networks.py:62-65,73-74); epsilon is the caller's
    cfg = _cfg()
    p = O.init_params(cfg)
    rng = np.random.default_rng(5)
    z1 = rng.standard_normal((8, 20)); z2 = rng.standard_normal((8, 12))
    (x_hat, mu, lv, eps), c = O.vae_forward(cfg, p, None, z1, z2, sampling=True, epsilon=-1.0)
    assert np.array_equal(c["samples"], z1) and np.all(mu == 0) and np.all(lv == 0)
    want = z1 @ p["Decoder/FC0/kernel"] + p["Decoder/FC0/bias"] + z2 * math.exp(-0.5)
    assert np.max(np.abs(x_hat - want)) < 1e-12


def test_kat_shapes():
    assert O.layer_sizes("", 5) == [5] and O.layer_sizes("200|200|200", 6) == [200, 200, 200, 6]
    D, _ = O.make_dataset("sigmoid", 69, dd=3, pad=3)
    assert D == 3 + 1 + 3                       # datasets.py:235
    z = np.zeros((4, 20 + 12))
    z1, z2 = O.split_latents(z, 20)             # model.py:227, vae.py:127-128
    assert z1.shape == (4, 20) and z2.shape == (4, 12)
    assert _cfg().n_params() == 533 and _cfg(latent_dim=2).n_params() == 65
    assert O.Config(7, 6, (256,), (256,), -3.0, True, "sigmoid").n_params() == 10779
    assert O.Config(6, 6, (512,) * 3, (512,) * 3, -3.0, True, "sphere").n_params() == 1063955
    assert O.Config(4096, 20, (), (), -1.0, True, "linear_gaussian").n_params() == 167977


def test_kat_epsilon_zero_cli_gives_zero_epsilon_grad():
    # eps = param * eps_cli: with the CLI default 0 the tunable parameter gets exactly zero gradient
    cfg = _cfg(epsilon=0.0)
    p = O.init_params(cfg)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((8, 12)); z1 = rng.standard_normal((8, 20)); z2 = rng.standard_normal((8, 12))
    _, g = O.loss_and_grad(cfg, p, x, z1, z2)
    assert g["epsilon"][0] == 0.0
