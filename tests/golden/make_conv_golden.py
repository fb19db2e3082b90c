"""Writes tests/golden/conv_vae_*.npz from oracle/conv_vae_oracle.py (no reference counterpart exists for the convolutional
model: these fixtures freeze the oracle against itself, as tests/golden/make_golden.py does for the MLP models).
    python tests/golden/make_conv_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import conv_vae_oracle as CO  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, size, widths, L, B, eps, tdv, seed):
    cfg = CO.ConvConfig(size, widths, L, eps, tdv)
    rng = np.random.default_rng(seed)
    p = CO.init_params(cfg, seed=seed + 1)
    for k in p:
        if not k.endswith("kernel"):
            p[k] = p[k] + 0.1 * rng.standard_normal(p[k].shape)
    x, z1, z2 = rng.random((B, size, size, 1)), rng.standard_normal((B, L)), rng.standard_normal((B, size, size, 1))
    loss, g = CO.loss_and_grad(cfg, p, x, z1, z2)
    out = dict(size=size, widths=np.array(widths), latent_dim=L, epsilon=eps, tdv=tdv, x=x, z1=z1, z2=z2, loss=loss)
    out.update({"p:" + k: v for k, v in p.items()})
    out.update({"g:" + k: v for k, v in g.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, cfg.n_params(), "parameters, loss", loss)


if __name__ == "__main__":
    make("conv_vae_small", 16, (3, 4, 5, 6), 5, 3, -1.5, True, 11)
    make("conv_vae_64", 64, (4, 8, 8, 16), 8, 2, -3.0, True, 12)
