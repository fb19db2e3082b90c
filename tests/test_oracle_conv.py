"""CPU: the convolutional VAE oracle (oracle/conv_vae_oracle.py -- BASELINE config 5; NO reference counterpart, the architecture
is this repository's own specification) against an independently written torch-autograd restatement, central
finite differences and its frozen fixtures."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conv_vae_oracle as CO

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _problem(cfg, B, seed=0):
    rng = np.random.default_rng(seed)
    p = CO.init_params(cfg, seed=seed + 1)
    for k in p:                                  # biases / epsilon_p / epsilon off their symmetric initial values
        if not k.endswith("kernel"):
            p[k] = p[k] + 0.1 * rng.standard_normal(p[k].shape)
    x = rng.random((B, cfg.S, cfg.S, 1))
    return p, x, rng.standard_normal((B, cfg.L)), rng.standard_normal((B, cfg.S, cfg.S, 1))


def _torch_loss(cfg, tp, x, z1, z2):
    """Written against torch's own layer functions (NCHW, OIHW), not against the oracle's im2col."""
    h = x.permute(0, 3, 1, 2)
    for i in range(4):
        w = tp[f"Encoder/Conv{i}/kernel"].permute(3, 2, 0, 1)               # HWIO -> OIHW
        h = torch.relu(F.conv2d(h, w, tp[f"Encoder/Conv{i}/bias"], stride=2, padding=1))
    flat = h.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                    # the oracle flattens NHWC
    mu = flat @ tp["Encoder/FC/kernel"] + tp["Encoder/FC/bias"]
    lv = tp["epsilon_p"]
    samples = mu + torch.exp(lv / 2) * z1
    d = torch.relu(samples @ tp["Decoder/FC/kernel"] + tp["Decoder/FC/bias"])
    h = d.reshape(x.shape[0], cfg.S // 16, cfg.S // 16, cfg.widths[3]).permute(0, 3, 1, 2)
    for i in range(4):
        w = tp[f"Decoder/ConvT{i}/kernel"].permute(3, 2, 0, 1)              # [kh, kw, co, ci] -> [ci, co, kh, kw]
        h = F.conv_transpose2d(h, w, tp[f"Decoder/ConvT{i}/bias"], stride=2, padding=1)
        if i < 3:
            h = torch.relu(h)
    eps = tp["epsilon"][0] * cfg.epsilon if cfg.tdv else torch.tensor(cfg.epsilon, dtype=torch.float64)
    x_hat = h.permute(0, 2, 3, 1) + z2 * torch.exp(eps / 2)
    dkl = -0.5 * torch.sum(1 + lv - torch.exp(lv) - mu ** 2, dim=-1)
    mse = (0.5 * ((x_hat - x) ** 2).reshape(x.shape[0], -1) / torch.exp(eps) + 0.5 * (math.log(2 * math.pi) + eps)).sum(dim=-1)
    return (dkl + mse).mean()


@pytest.mark.parametrize("size,widths,L,B,tdv", [(16, (3, 4, 5, 6), 5, 3, True), (32, (4, 8, 8, 16), 7, 2, False), (64, (2, 3, 4, 8), 6, 2, True)])
def test_conv_oracle_matches_torch_autograd(size, widths, L, B, tdv):
    cfg = CO.ConvConfig(size, widths, L, -1.5, tdv)
    p, x, z1, z2 = _problem(cfg, B)
    loss, g = CO.loss_and_grad(cfg, p, x, z1, z2)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    tl = _torch_loss(cfg, tp, *(torch.tensor(a, dtype=torch.float64) for a in (x, z1, z2)))
    tl.backward()
    assert abs(loss - tl.item()) <= 1e-11 * abs(tl.item())
    for k, v in g.items():
        want = tp[k].grad.numpy()
        assert v.shape == want.shape, k
        assert np.max(np.abs(v - want)) <= 1e-11 * (np.max(np.abs(want)) + 1e-30), k


def test_conv_layers_are_adjoint_pairs():
    """<conv(x), y> == <x, conv_t(y)> with the same kernel array: the definition the specification uses."""
    rng = np.random.default_rng(3)
    x, K = rng.standard_normal((2, 8, 8, 3)), rng.standard_normal((4, 4, 3, 5))
    y = rng.standard_normal((2, 4, 4, 5))
    lhs = np.sum(CO.conv_fwd(x, K, np.zeros(5)) * y)
    rhs = np.sum(x * CO.conv_t_fwd(y, K, np.zeros(3)))
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs)


def test_conv_oracle_finite_differences():
    cfg = CO.ConvConfig(16, (2, 3, 3, 4), 4, -2.0, True)
    p, x, z1, z2 = _problem(cfg, 2, seed=5)
    _, g = CO.loss_and_grad(cfg, p, x, z1, z2)
    rng = np.random.default_rng(0)
    for name in ("Encoder/Conv0/kernel", "Encoder/Conv3/bias", "Encoder/FC/kernel", "Decoder/FC/bias", "Decoder/ConvT0/kernel",
                 "Decoder/ConvT3/kernel", "Decoder/ConvT3/bias", "epsilon_p", "epsilon"):
        for _ in range(3):
            idx = tuple(rng.integers(0, s) for s in p[name].shape)
            h = 1e-6
            q = {k: v.copy() for k, v in p.items()}
            q[name][idx] += h
            up = CO.loss_and_grad(cfg, q, x, z1, z2)[0]
            q[name][idx] -= 2 * h
            dn = CO.loss_and_grad(cfg, q, x, z1, z2)[0]
            fd = (up - dn) / (2 * h)
            assert abs(fd - g[name][idx]) <= 1e-6 * max(1.0, abs(fd)), (name, idx, fd, g[name][idx])


def test_baseline_config5_shape_bookkeeping():
    cfg = CO.ConvConfig()                        # 64 x 64, widths 32 | 64 | 128 | 256, L = 32
    assert cfg.bott == 4096
    assert cfg.n_params() == 2 * (16 * (1 * 32 + 32 * 64 + 64 * 128 + 128 * 256)) + (32 + 64 + 128 + 256) + (128 + 64 + 32 + 1) \
        + 4096 * 32 + 32 + 32 * 4096 + 4096 + 32 + 1


@pytest.mark.parametrize("name", ["conv_vae_small", "conv_vae_64"])
def test_conv_golden_fixtures(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = CO.ConvConfig(int(z["size"]), tuple(int(w) for w in z["widths"]), int(z["latent_dim"]), float(z["epsilon"]), bool(z["tdv"]))
    p = {k[2:]: z[k] for k in z.files if k.startswith("p:")}
    loss, g = CO.loss_and_grad(cfg, p, z["x"], z["z1"], z["z2"])
    assert abs(loss - float(z["loss"])) <= 1e-12 * abs(float(z["loss"]))
    for k, v in g.items():
        assert np.max(np.abs(v - z["g:" + k])) <= 1e-12 * (np.max(np.abs(v)) + 1e-30), k
