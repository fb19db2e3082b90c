"""CPU oracle for the ELBO train step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``vae_training_amd``) never routes through it.

PARITY UNPINNED: the reference (virajmehta/vae-training) ships no tests, golden vectors or
fixtures for this path and its JAX/Flax dependencies are not
importable here (ordinary ModuleNotFoundError), so no reference output exists to pin
against.  This file is a float64 NumPy restatement of the arithmetic at the reference's
call sites with a HAND-DERIVED backward pass.  It is cross-checked against an
independently written torch-autograd restatement (``oracle/elbo_torch.py``), central
finite differences, and closed-form known answers (``tests/test_oracle_kat.py``).

Reference lines each function follows (paths are /root/reference/...):

* ``fcn_forward``        networks.py:26-44   FullyConnectedNetwork.apply
* ``vae_forward``        networks.py:61-84   VAE.apply
* ``elbo_terms``         networks.py:94-98   (train) / :106-113 (eval twin VAE.loss)
* ``loss_and_grad``      networks.py:87-99   value_and_grad(loss_fn)
* ``adam_update``        networks.py:100 + vae.py:112-113  flax.optim.Adam.apply_gradient
* ``layer_sizes``        vae.py:53-54
* ``split_latents``      vae.py:126-128, model.py:225-228

Third-party semantics (flax.nn.Dense, flax.optim.Adam, jax initialisers) are not under
/root/reference and are unpinned there (README.md:8-13).  Everything restated from API
knowledge is marked ASSUMED-FROM-API and kept in this one file.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)

# ASSUMED-FROM-API: flax.optim.Adam defaults (beta1, beta2, eps, weight_decay)
ADAM_BETA1 = 0.9
ADAM_BETA2 = 0.999
ADAM_EPS = 1e-8


# --------------------------------------------------------------------------------------
# configuration / parameter tree
# --------------------------------------------------------------------------------------
def layer_sizes(spec: str, last: int):
    """vae.py:53-54 -- "a|b|c" -> [a, b, c, last]; "" -> [last]."""
    return ([int(s) for s in spec.split("|")] if spec != "" else []) + [int(last)]


class Config:
    """Static description of one VAE (the arguments bound by VAE.partial, vae.py:57-59)."""

    def __init__(self, data_dim, latent_dim, enc_hidden=(), dec_hidden=(), epsilon=0.0,
                 tunable_decoder_var=False, dataset_name=None):
        self.D = int(data_dim)
        self.L = int(latent_dim)
        self.enc_sizes = list(enc_hidden) + [self.L]      # vae.py:53
        self.dec_sizes = list(dec_hidden) + [self.D]      # vae.py:54
        self.epsilon = float(epsilon)
        self.tdv = bool(tunable_decoder_var)
        self.sigmoid = dataset_name == "sigmoid"           # networks.py:75
        self.dataset_name = dataset_name

    @property
    def networks(self):
        nets = [("Encoder", self.D, self.enc_sizes), ("Decoder", self.L, self.dec_sizes)]
        if self.sigmoid:
            nets.append(("SigDecoder", self.L, self.dec_sizes))
        return nets

    def leaves(self):
        """Flat leaf order shared with the HIP library (include/vaek.h):
        Encoder FC0 kernel, FC0 bias, ..., Decoder ..., [SigDecoder ...], epsilon_p, [epsilon]."""
        out = []
        for name, fan_in, sizes in self.networks:
            k = fan_in
            for i, n in enumerate(sizes):
                out.append((f"{name}/FC{i}/kernel", (k, n)))
                out.append((f"{name}/FC{i}/bias", (n,)))
                k = n
        out.append(("epsilon_p", (self.L,)))
        if self.tdv:
            out.append(("epsilon", (1,)))
        return out

    def n_params(self):
        return sum(int(np.prod(s)) for _, s in self.leaves())


def init_params(cfg: Config, seed=0, dtype=np.float64):
    """vae.py:60 init_by_shape.  ASSUMED-FROM-API: flax.nn.Dense default kernel_init is
    jax.nn.initializers.lecun_normal() = truncated normal on [-2, 2] scaled by
    sqrt(1/fan_in)/0.87962566103423978; bias zeros.  epsilon_p / epsilon: ones
    (networks.py:69-71).  The RNG stream is this oracle's own (jax.random streams cannot be
    reproduced without JAX); only the distribution is restated."""
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    for name, shape in cfg.leaves():
        if name.endswith("kernel"):
            fan_in = shape[0]
            std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
            w = rng.standard_normal(shape)
            bad = np.abs(w) > 2.0
            while bad.any():                      # rejection = truncated normal on [-2, 2]
                w[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(w) > 2.0
            p[name] = (w * std).astype(dtype)
        elif name.endswith("bias"):
            p[name] = np.zeros(shape, dtype)
        else:
            p[name] = np.ones(shape, dtype)
    return p


def flatten(cfg: Config, tree, dtype=None):
    v = np.concatenate([np.asarray(tree[n]).reshape(-1) for n, _ in cfg.leaves()])
    return v.astype(dtype) if dtype is not None else v


def unflatten(cfg: Config, flat):
    out, o = OrderedDict(), 0
    for n, s in cfg.leaves():
        k = int(np.prod(s))
        out[n] = np.asarray(flat[o:o + k]).reshape(s)
        o += k
    return out


def split_latents(z, latent_dim):
    """vae.py:126-128 -- z has L + D columns (model.py:227); z1 = first L, z2 = the rest."""
    return z[..., :latent_dim], z[..., latent_dim:]


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
def _sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


def fcn_forward(p, name, x, sizes, if_sigmoid=False):
    """networks.py:26-44.  Dense (ASSUMED-FROM-API: y = x @ kernel + bias) per layer, relu
    (utils.py:29-30) after every layer except the last, optional final sigmoid.
    Returns the output and the list of layer inputs (needed by the backward)."""
    acts = [x]
    h = x
    for i in range(len(sizes)):
        h = h @ p[f"{name}/FC{i}/kernel"] + p[f"{name}/FC{i}/bias"]
        if i + 1 < len(sizes):
            h = np.maximum(h, 0.0)
            acts.append(h)
    if if_sigmoid:
        h = _sigmoid(h)
    return h, acts


def effective_epsilon(cfg: Config, p):
    """networks.py:70-71 -- with -tdv: param('epsilon',(1,),ones) * epsilon_cli, shape (1,);
    otherwise the python float bound by VAE.partial (vae.py:57)."""
    if cfg.tdv:
        return p["epsilon"] * cfg.epsilon
    return np.float64(cfg.epsilon)


def vae_forward(cfg: Config, p, x, z1, z2, sampling=False, epsilon=None):
    """networks.py:61-84.  ``sampling=True`` (networks.py:62-65): mu = 0, logvar_e = 0,
    epsilon is whatever the caller passes (vae.py:199 passes current_epsilon)."""
    cache = {}
    if sampling:
        mu = np.zeros_like(z1)
        logvar_e = np.zeros(z1.shape[-1], z1.dtype)
        eps = np.asarray(cfg.epsilon if epsilon is None else epsilon, dtype=z1.dtype)
    else:
        mu, cache["enc_acts"] = fcn_forward(p, "Encoder", x, cfg.enc_sizes)
        logvar_e = p["epsilon_p"]
        eps = effective_epsilon(cfg, p)
    stdevs = np.exp(logvar_e / 2.0)                      # :73
    samples = mu + stdevs * z1                           # :74
    if cfg.sigmoid:                                      # :75-78
        sg, cache["sig_acts"] = fcn_forward(p, "SigDecoder", samples, cfg.dec_sizes, if_sigmoid=True)
        lin, cache["dec_acts"] = fcn_forward(p, "Decoder", samples, cfg.dec_sizes)
        x_hat = sg + lin
        cache["sg"] = sg
    else:                                                # :80
        x_hat, cache["dec_acts"] = fcn_forward(p, "Decoder", samples, cfg.dec_sizes)
    stdev = np.exp(eps / 2.0)                            # :81
    x_hat = x_hat + z2 * stdev                           # :82-83
    cache.update(mu=mu, samples=samples, stdevs=stdevs, stdev=stdev)
    return (x_hat, mu, logvar_e, eps), cache


def elbo_terms(x, x_hat, mu, logvar_e, eps):
    """networks.py:94-98 / :106-112.  Returns (loss.mean(), Dkl.mean(), mse.mean())."""
    dkl = -0.5 * np.sum(1.0 + logvar_e - np.exp(logvar_e) - np.square(mu), axis=-1)
    var_d = np.exp(eps)
    mse = (0.5 * np.square(x_hat - x) / var_d + 0.5 * (LOG_2PI + eps)).sum(axis=-1)
    return (dkl + mse).mean(), dkl.mean(), mse.mean()


def loss_eval(cfg: Config, p, x, z1, z2):
    """VAE.loss, networks.py:103-113: (loss, Dkl.mean(), mse.mean(), logvar_e, epsilon)."""
    (x_hat, mu, lv, eps), _ = vae_forward(cfg, p, x, z1, z2)
    loss, dkl, mse = elbo_terms(x, x_hat, mu, lv, eps)
    return loss, dkl, mse, lv, eps


# --------------------------------------------------------------------------------------
# hand-derived backward (what jax.value_and_grad computes at networks.py:99)
# --------------------------------------------------------------------------------------
def _fcn_backward(p, g, name, acts, d_out, sizes, need_dx=True):
    """Back-propagate d_out (gradient w.r.t. the PRE-sigmoid output of the last Dense)
    through the Dense/relu stack.  acts[i] is the input of layer i (post-relu for i>0)."""
    d = d_out
    for i in reversed(range(len(sizes))):
        h_in = acts[i]
        g[f"{name}/FC{i}/kernel"] = h_in.T @ d
        g[f"{name}/FC{i}/bias"] = d.sum(axis=0)
        if i > 0 or need_dx:
            d = d @ p[f"{name}/FC{i}/kernel"].T
            if i > 0:
                d = d * (h_in > 0.0)               # relu'(pre) == [post > 0]
    return d


def loss_and_grad(cfg: Config, p, x, z1, z2, batch_total=None):
    """Loss and gradient tree.  ``batch_total`` (default: this batch) is the divisor of the
    batch mean: a data-parallel shard passes the GLOBAL batch size so that the SUM of shard
    gradients equals the full-batch gradient."""
    B = x.shape[0]
    Bt = float(B if batch_total is None else batch_total)
    (x_hat, mu, lv, eps), c = vae_forward(cfg, p, x, z1, z2)
    eps_s = float(np.asarray(eps).reshape(-1)[0])
    inv_var = math.exp(-eps_s)
    r = x_hat - x
    dkl = -0.5 * np.sum(1.0 + lv - np.exp(lv) - np.square(mu), axis=-1)
    mse = (0.5 * np.square(r) * inv_var + 0.5 * (LOG_2PI + eps_s)).sum(axis=-1)
    loss = (dkl + mse).sum() / Bt

    g = OrderedDict()
    d_xhat = r * inv_var / Bt                                  # dL/dx_hat
    if cfg.sigmoid:
        sg = c["sg"]
        d_s = _fcn_backward(p, g, "SigDecoder", c["sig_acts"], d_xhat * sg * (1.0 - sg), cfg.dec_sizes)
        d_s = d_s + _fcn_backward(p, g, "Decoder", c["dec_acts"], d_xhat, cfg.dec_sizes)
    else:
        d_s = _fcn_backward(p, g, "Decoder", c["dec_acts"], d_xhat, cfg.dec_sizes)
    d_mu = d_s + mu / Bt                                       # reparam + KL
    _fcn_backward(p, g, "Encoder", c["enc_acts"], d_mu, cfg.enc_sizes, need_dx=False)
    # logvar_e = epsilon_p (shared over the batch): reparam path + KL path
    g["epsilon_p"] = 0.5 * c["stdevs"] * (d_s * z1).sum(axis=0) - 0.5 * (1.0 - np.exp(lv)) * (B / Bt)
    if cfg.tdv:
        # d/d eps of [0.5 r^2 e^-eps + 0.5 eps] with x_hat = ... + z2 e^{eps/2}
        d_eps = (-0.5 * np.square(r) * inv_var + 0.5 + 0.5 * c["stdev"] * z2 * r * inv_var).sum() / Bt
        g["epsilon"] = np.array([cfg.epsilon * d_eps])        # eps = param * eps_cli
    # re-order like cfg.leaves()
    g = OrderedDict((n, g[n]) for n, _ in cfg.leaves())
    return loss, g


# --------------------------------------------------------------------------------------
# Adam (ASSUMED-FROM-API: flax.optim.Adam, pre-Linen)
# --------------------------------------------------------------------------------------
def adam_init(p):
    return {"step": 0,
            "m": OrderedDict((k, np.zeros_like(v)) for k, v in p.items()),
            "v": OrderedDict((k, np.zeros_like(v)) for k, v in p.items())}


def adam_update(p, g, state, lr, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS):
    """m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; t = step+1 ;
    p -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)."""
    t = state["step"] + 1
    new_p, new_m, new_v = OrderedDict(), OrderedDict(), OrderedDict()
    for k in p:
        m = beta1 * state["m"][k] + (1.0 - beta1) * g[k]
        v = beta2 * state["v"][k] + (1.0 - beta2) * np.square(g[k])
        mh = m / (1.0 - beta1 ** t)
        vh = v / (1.0 - beta2 ** t)
        new_p[k] = p[k] - lr * mh / (np.sqrt(vh) + eps)
        new_m[k], new_v[k] = m, v
    return new_p, {"step": t, "m": new_m, "v": new_v}


def train_step(cfg: Config, p, state, x, z1, z2, lr):
    """VAE.train_step, networks.py:87-101: returns (new params, new opt state, loss)."""
    loss, g = loss_and_grad(cfg, p, x, z1, z2)
    p, state = adam_update(p, g, state, lr)
    return p, state, loss


# --------------------------------------------------------------------------------------
# synthetic inputs (datasets.py restated; the oracle's own RNG)
# --------------------------------------------------------------------------------------
def make_dataset(name, seed, dd=3, did=3, pad=0, var_added=0.0):
    """Returns (D, sampler(rng, B) -> (B, D)).  linear: datasets.py:164-195;
    sigmoid: datasets.py:231-249; sphere: datasets.py:56-84."""
    rng = np.random.default_rng(seed)
    if name == "linear_gaussian":
        A = rng.standard_normal((dd, did))
        while np.linalg.matrix_rank(A) != min(dd, did):
            A = rng.standard_normal((dd, did))
        D = dd + pad

        def sample(r, B):
            X = r.standard_normal((B, did))
            Y = np.concatenate([X @ A.T, np.zeros((B, pad))], axis=1)
            if var_added > 0:
                Y = Y + r.standard_normal((B, D)) * math.sqrt(var_added)
            return Y
        return D, sample
    if name == "sigmoid":
        a = rng.standard_normal((dd, 1))
        D = dd + 1 + pad

        def sample(r, B):
            z = r.standard_normal((B, dd))
            return np.concatenate([z, _sigmoid(z @ a), np.zeros((B, pad))], axis=1)
        return D, sample
    if name == "sphere":
        D = dd + pad

        def sample(r, B):
            gsn = r.standard_normal((B, dd))
            return np.concatenate([gsn / np.linalg.norm(gsn, axis=1, keepdims=True), np.zeros((B, pad))], axis=1)
        return D, sample
    raise ValueError(name)
