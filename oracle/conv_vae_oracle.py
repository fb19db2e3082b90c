"""CPU oracle for a CONVOLUTIONAL VAE train step (BASELINE.json config 5) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NO REFERENCE COUNTERPART.  virajmehta/vae-training has no convolutional model (its only image code is
/root/reference/utils.py:129-133, a cv2 plotting helper); BASELINE config 5 names "conv encoder/decoder VAE on 64x64
images", so the architecture below is this repository's own specification.  What IS taken from the
reference is everything around the two networks, restated from oracle/elbo_oracle.py:

* the latent log-variance is the batch-independent parameter vector ``epsilon_p``            networks.py:68-72
* reparameterisation ``samples = mu + exp(logvar_e / 2) * z1``                               networks.py:73-74
* decoder noise ``x_hat + z2 * exp(eps / 2)`` and the tunable decoder variance ``epsilon``   networks.py:70-71, 81-83
* the loss: closed-form KL + Gaussian log-likelihood, batch mean                              networks.py:94-98

PARITY UNPINNED (there is nothing to pin against): the file is checked by an independently written torch-autograd
restatement (tests/test_oracle_conv.py, <= 1e-11), central finite differences, and frozen fixtures
(tests/golden/conv_vae_*.npz, written by tests/golden/make_conv_golden.py from this file).

Specification (NHWC activations, HWIO kernels, float64):

    encoder   x [B, S, S, 1]  -> conv 4x4 / stride 2 / pad 1, C0 -> relu -> ... four of them, channels widths[0..3]
              (S -> S/2 -> S/4 -> S/8 -> S/16), flatten [B, (S/16)^2 * widths[3]], Dense -> mu [B, L]
    decoder   samples [B, L] -> Dense -> [B, (S/16)^2 * widths[3]] -> relu -> reshape [B, S/16, S/16, widths[3]]
              -> transposed conv 4x4 / stride 2 / pad 1 -> relu -> ... four of them, channels widths[2], widths[1],
              widths[0], 1 (no relu after the last) -> x_hat [B, S, S, 1]
    BASELINE config 5: S = 64, widths = (32, 64, 128, 256), i.e. 64 -> 32 -> 16 -> 8 -> 4 and a 4096-wide bottleneck.

A transposed convolution is DEFINED here as the adjoint of the forward convolution with the same kernel array:
``conv_t(y, K) = d<conv(x, K), y>/dx`` with K [4, 4, C_out_of_conv_t, C_in_of_conv_t] (the HWIO kernel of the
convolution it is the adjoint of).  torch's ``conv_transpose2d(y, w, stride=2, padding=1)`` computes the same map for
``w[ci, co, kh, kw] = K[kh, kw, co, ci]``.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)
KS, STRIDE, PAD = 4, 2, 1


class ConvConfig:
    def __init__(self, size=64, widths=(32, 64, 128, 256), latent_dim=32, epsilon=-3.0, tunable_decoder_var=True):
        assert size % 16 == 0 and len(widths) == 4
        self.S, self.widths, self.L, self.epsilon, self.tdv = size, tuple(widths), latent_dim, float(epsilon), tunable_decoder_var
        self.bott = (size // 16) ** 2 * widths[3]

    def leaves(self):
        """(name, shape) in the flat parameter order: encoder convs, encoder Dense, decoder Dense, decoder transposed
        convs, epsilon_p, [epsilon] -- the order of oracle/elbo_oracle.Config.leaves for the MLP models."""
        out, cin = [], 1
        for i, c in enumerate(self.widths):
            out += [(f"Encoder/Conv{i}/kernel", (KS, KS, cin, c)), (f"Encoder/Conv{i}/bias", (c,))]
            cin = c
        out += [("Encoder/FC/kernel", (self.bott, self.L)), ("Encoder/FC/bias", (self.L,)),
                ("Decoder/FC/kernel", (self.L, self.bott)), ("Decoder/FC/bias", (self.bott,))]
        chans = [self.widths[3], self.widths[2], self.widths[1], self.widths[0], 1]
        for i in range(4):
            # kernel of the convolution this layer is the adjoint of: [kh, kw, C_out_of_this_layer, C_in_of_this_layer]
            out += [(f"Decoder/ConvT{i}/kernel", (KS, KS, chans[i + 1], chans[i])), (f"Decoder/ConvT{i}/bias", (chans[i + 1],))]
        out.append(("epsilon_p", (self.L,)))
        if self.tdv:
            out.append(("epsilon", (1,)))
        return out

    def n_params(self):
        return sum(int(np.prod(s)) for _, s in self.leaves())


def init_params(cfg: ConvConfig, seed=0):
    """LeCun-normal kernels (fan-in = kh * kw * C_in for a convolution; for a transposed layer the fan-in of the map it
    computes, kh * kw * C_in_of_this_layer / stride^2), zero biases, epsilon_p = 0, epsilon = 1 (networks.py:68-71)."""
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    for name, shape in cfg.leaves():
        if name.endswith("kernel"):
            if "ConvT" in name:
                fan_in = KS * KS * shape[3] / STRIDE ** 2
            elif "Conv" in name:
                fan_in = KS * KS * shape[2]
            else:
                fan_in = shape[0]
            p[name] = rng.standard_normal(shape) / math.sqrt(fan_in)
        elif name == "epsilon":
            p[name] = np.ones(shape)
        else:
            p[name] = np.zeros(shape)
    return p


# ---- the two layer kinds -----------------------------------------------------------------------------------------------------
def _patches(x):
    """im2col of a 4x4 / stride 2 / pad 1 convolution: x [B, H, W, C] -> [B, H/2, W/2, 4, 4, C] (a copy)."""
    B, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (PAD, PAD), (PAD, PAD), (0, 0)))
    sb, sh, sw, sc = xp.strides
    v = np.lib.stride_tricks.as_strided(xp, (B, H // STRIDE, W // STRIDE, KS, KS, C), (sb, sh * STRIDE, sw * STRIDE, sh, sw, sc))
    return np.ascontiguousarray(v)


def _scatter_patches(cols, H, W):
    """adjoint of _patches: cols [B, H/2, W/2, 4, 4, C] -> [B, H, W, C] (sum over the overlapping windows)."""
    B, Ho, Wo, _, _, C = cols.shape
    xp = np.zeros((B, H + 2 * PAD, W + 2 * PAD, C), cols.dtype)
    for kh in range(KS):
        for kw in range(KS):
            xp[:, kh:kh + STRIDE * Ho:STRIDE, kw:kw + STRIDE * Wo:STRIDE, :] += cols[:, :, :, kh, kw, :]
    return xp[:, PAD:PAD + H, PAD:PAD + W, :]


def conv_fwd(x, K, b):
    """y[n, i, j, o] = b[o] + sum_{kh, kw, c} x_pad[n, 2 i + kh, 2 j + kw, c] K[kh, kw, c, o]"""
    return np.tensordot(_patches(x), K, axes=([3, 4, 5], [0, 1, 2])) + b


def conv_bwd(x, K, dy):
    """(dx, dK, db) of conv_fwd."""
    dK = np.tensordot(_patches(x), dy, axes=([0, 1, 2], [0, 1, 2]))
    dcols = np.tensordot(dy, K, axes=([3], [3]))                   # [B, Ho, Wo, kh, kw, C]
    return _scatter_patches(dcols, x.shape[1], x.shape[2]), dK, dy.sum(axis=(0, 1, 2))


def conv_t_fwd(y, K, b):
    """the adjoint of conv_fwd(., K, 0) applied to y [B, h, w, C_in] -> [B, 2 h, 2 w, C_out], K [4, 4, C_out, C_in], + bias"""
    return _scatter_patches(np.tensordot(y, K, axes=([3], [3])), 2 * y.shape[1], 2 * y.shape[2]) + b


def conv_t_bwd(y, K, dout):
    """(dy, dK, db) of conv_t_fwd: the adjoint of an adjoint is the convolution itself."""
    cols = _patches(dout)                                          # [B, h, w, kh, kw, C_out]
    dy = np.tensordot(cols, K, axes=([3, 4, 5], [0, 1, 2]))
    dK = np.tensordot(cols, y, axes=([0, 1, 2], [0, 1, 2]))        # [kh, kw, C_out, C_in]
    return dy, dK, dout.sum(axis=(0, 1, 2))


# ---- the model ---------------------------------------------------------------------------------------------------------------
def effective_epsilon(cfg: ConvConfig, p):
    return float(p["epsilon"][0]) * cfg.epsilon if cfg.tdv else cfg.epsilon


def forward(cfg: ConvConfig, p, x, z1, z2):
    """x, z2 [B, S, S, 1]; z1 [B, L].  Returns (x_hat incl. decoder noise, mu, logvar_e, eps) and the cache for backward."""
    B = x.shape[0]
    c = {"enc_in": [], "dec_in": []}
    h = x
    for i in range(4):
        c["enc_in"].append(h)
        h = np.maximum(conv_fwd(h, p[f"Encoder/Conv{i}/kernel"], p[f"Encoder/Conv{i}/bias"]), 0.0)
    c["flat"] = h.reshape(B, -1)
    mu = c["flat"] @ p["Encoder/FC/kernel"] + p["Encoder/FC/bias"]
    lv = p["epsilon_p"]
    stdevs = np.exp(lv / 2.0)
    samples = mu + stdevs * z1
    d = np.maximum(samples @ p["Decoder/FC/kernel"] + p["Decoder/FC/bias"], 0.0)
    c["dec_fc_out"] = d
    h = d.reshape(B, cfg.S // 16, cfg.S // 16, cfg.widths[3])
    for i in range(4):
        c["dec_in"].append(h)
        h = conv_t_fwd(h, p[f"Decoder/ConvT{i}/kernel"], p[f"Decoder/ConvT{i}/bias"])
        if i < 3:
            h = np.maximum(h, 0.0)
    eps = effective_epsilon(cfg, p)
    stdev = math.exp(eps / 2.0)
    x_hat = h + z2 * stdev
    c.update(mu=mu, samples=samples, stdevs=stdevs, stdev=stdev, enc_out=c["flat"])
    return (x_hat, mu, lv, eps), c


def elbo_terms(x, x_hat, mu, lv, eps):
    """networks.py:94-98 with the pixel axes flattened: (loss.mean(), Dkl.mean(), mse.mean())."""
    B = x.shape[0]
    dkl = -0.5 * np.sum(1.0 + lv - np.exp(lv) - np.square(mu), axis=-1)
    mse = (0.5 * np.square(x_hat - x).reshape(B, -1) * math.exp(-eps) + 0.5 * (LOG_2PI + eps)).sum(axis=-1)
    return (dkl + mse).mean(), dkl.mean(), mse.mean()


def loss_and_grad(cfg: ConvConfig, p, x, z1, z2, batch_total=None):
    """Loss and the gradient tree (hand-derived; what jax.value_and_grad would return at networks.py:99)."""
    B = x.shape[0]
    Bt = float(B if batch_total is None else batch_total)
    (x_hat, mu, lv, eps), c = forward(cfg, p, x, z1, z2)
    inv_var = math.exp(-eps)
    r = x_hat - x
    dkl = -0.5 * np.sum(1.0 + lv - np.exp(lv) - np.square(mu), axis=-1)
    mse = (0.5 * np.square(r).reshape(B, -1) * inv_var + 0.5 * (LOG_2PI + eps)).sum(axis=-1)
    loss = (dkl + mse).sum() / Bt
    g = {}
    d = r * inv_var / Bt                                            # dL/dx_hat, [B, S, S, 1]
    for i in reversed(range(4)):
        h_in = c["dec_in"][i]
        d, g[f"Decoder/ConvT{i}/kernel"], g[f"Decoder/ConvT{i}/bias"] = conv_t_bwd(h_in, p[f"Decoder/ConvT{i}/kernel"], d)
        d = d * (h_in > 0.0)                                        # relu behind the previous layer (behind the Dense for i = 0)
    d = d.reshape(B, -1)
    g["Decoder/FC/kernel"] = c["samples"].T @ d
    g["Decoder/FC/bias"] = d.sum(axis=0)
    d_s = d @ p["Decoder/FC/kernel"].T
    d_mu = d_s + mu / Bt
    g["Encoder/FC/kernel"] = c["flat"].T @ d_mu
    g["Encoder/FC/bias"] = d_mu.sum(axis=0)
    d = (d_mu @ p["Encoder/FC/kernel"].T).reshape(B, cfg.S // 16, cfg.S // 16, cfg.widths[3])
    h_out = c["flat"].reshape(d.shape)
    for i in reversed(range(4)):
        d = d * (h_out > 0.0)
        h_in = c["enc_in"][i]
        d, g[f"Encoder/Conv{i}/kernel"], g[f"Encoder/Conv{i}/bias"] = conv_bwd(h_in, p[f"Encoder/Conv{i}/kernel"], d)
        h_out = h_in
    g["epsilon_p"] = 0.5 * c["stdevs"] * (d_s * z1).sum(axis=0) - 0.5 * (1.0 - np.exp(lv)) * (B / Bt)
    if cfg.tdv:
        d_eps = (-0.5 * np.square(r) * inv_var + 0.5 + 0.5 * c["stdev"] * z2 * r * inv_var).sum() / Bt
        g["epsilon"] = np.array([cfg.epsilon * d_eps])
    return loss, OrderedDict((n, g[n]) for n, _ in cfg.leaves())
