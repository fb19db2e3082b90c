"""Independent torch-autograd restatement of the ELBO train step -- TEST INFRASTRUCTURE.

Second, separately written statement of /root/reference/networks.py:61-101 used (a) to
cross-check the hand-derived backward in ``oracle/elbo_oracle.py`` (float64, autograd
instead of hand formulas, torch.optim.Adam instead of the hand-written update) and (b) as
the float32 multi-threaded "CPU restatement (torch), not JAX" timed by ``bench.py``'s
``cpu_baseline`` leg.  PARITY UNPINNED, see elbo_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math

import torch

LOG_2PI = math.log(2.0 * math.pi)


class TorchVAE(torch.nn.Module):
    """Parameters are created from a name->array tree (oracle leaf names)."""

    def __init__(self, cfg, tree, dtype=torch.float64):
        super().__init__()
        self.cfg = cfg
        self.names = [n for n, _ in cfg.leaves()]
        self.p = torch.nn.ParameterDict(
            {n.replace("/", "__"): torch.nn.Parameter(torch.as_tensor(tree[n], dtype=dtype).clone())
             for n in self.names})

    def get(self, name):
        return self.p[name.replace("/", "__")]

    def mlp(self, name, h, sizes, final_sigmoid=False):
        # networks.py:26-44
        for i in range(len(sizes)):
            h = h @ self.get(f"{name}/FC{i}/kernel") + self.get(f"{name}/FC{i}/bias")
            if i + 1 < len(sizes):
                h = torch.clamp_min(h, 0.0)
        return torch.sigmoid(h) if final_sigmoid else h

    def forward(self, x, z1, z2):
        # networks.py:61-84 (non-sampling)
        cfg = self.cfg
        mu = self.mlp("Encoder", x, cfg.enc_sizes)
        logvar_e = self.get("epsilon_p")
        eps = self.get("epsilon") * cfg.epsilon if cfg.tdv else torch.tensor(cfg.epsilon, dtype=x.dtype)
        samples = mu + torch.exp(logvar_e / 2) * z1
        if cfg.sigmoid:
            x_hat = self.mlp("SigDecoder", samples, cfg.dec_sizes, True) + self.mlp("Decoder", samples, cfg.dec_sizes)
        else:
            x_hat = self.mlp("Decoder", samples, cfg.dec_sizes)
        x_hat = x_hat + z2 * torch.exp(eps / 2.0)
        return x_hat, mu, logvar_e, eps

    def elbo(self, x, z1, z2):
        # networks.py:90-98
        x_hat, mu, logvar_e, eps = self(x, z1, z2)
        dkl = -0.5 * torch.sum(1 + logvar_e - torch.exp(logvar_e) - mu * mu, dim=-1)
        mse = (0.5 * (x_hat - x) ** 2 / torch.exp(eps) + 0.5 * (LOG_2PI + eps)).sum(dim=-1)
        return (dkl + mse).mean(), dkl.mean(), mse.mean()


def make_adam(model, lr):
    # ASSUMED-FROM-API: flax.optim.Adam(learning_rate) == torch Adam(b=(0.9,0.999), eps=1e-8)
    return torch.optim.Adam(model.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8)


def train_step(model, opt, x, z1, z2):
    """networks.py:87-101."""
    opt.zero_grad(set_to_none=True)
    loss, _, _ = model.elbo(x, z1, z2)
    loss.backward()
    opt.step()
    return loss.detach()


def grads_tree(model):
    return {n: model.get(n).grad.detach().numpy().copy() for n in model.names}


def params_tree(model):
    return {n: model.get(n).detach().numpy().copy() for n in model.names}
