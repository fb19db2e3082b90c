"""The three synthetic manifold datasets reachable from the CLI (run.py:46-54), generated on the
device: SphereDataset (datasets.py:55-98), LinearGaussianDataset (:163-228), SigmoidDataset
(:230-279).  Same constructor arguments, `shape`, `dimension`, `get_batch(size,
return_latents=False)`, `score_batch`, `plot_batch`, `save`/`load` (no-ops there too).
Draws use this package's key-splitting RNG (random.py); only the distributions are pinned."""
from __future__ import annotations

import math

import torch

from . import random as vrandom


DEVICE_DRAW_MAX_DIM = 16      # csrc/rng.hip make_batch_args: -dd / -did above this are drawn with torch ops instead


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class DistributionDataset:
    is_epochs = False

    def __init__(self, seed, device=None):
        self.key = vrandom.PRNGKey(seed)
        self.device = torch.device(device) if device is not None else _device()

    def get_key(self):
        self.key, key = vrandom.split(self.key)
        return key

    # ---- on a GPU, batches come from libvaek's Philox kernel (csrc/rng.hip): one launch, counter-based --
    _draws = 0

    def _device_batch(self, size):
        """x[size, ndim] from vaek_make_batch (kind/A/dd/did/pad of device_spec), or None on a CPU-only host."""
        if self.device.type != "cuda":
            return None
        kind, A, dd, did, pad, var = self.device_spec()
        if dd > DEVICE_DRAW_MAX_DIM or did > DEVICE_DRAW_MAX_DIM:
            return None         # vaek_make_batch keeps a row's normals in registers (dd, did <= 16): wider manifolds use the torch draw
        from .engine import Engine
        eng = getattr(self, "_util_engine", None)
        if eng is None:
            eng = self._util_engine = Engine(1, self.ndim, 1, device=self.device.index)   # carries no model: x-only draws
        self._draws += 1
        x, _, _ = eng.make_batch(kind, A, dd, did, pad, var, size, seed=self.key[0] ^ self.key[1], step=self._draws, tag=1,
                                 want_z=False)
        return x

    @property
    def dimension(self):
        return int(math.prod(self.shape))

    def save(self, fn):
        pass

    def load(self, fn):
        pass

    def device_spec(self):
        """(kind, A, dd, did, pad, var_added) for libvaek's make_batch kernel (include/vaek.h)."""
        raise NotImplementedError

    def _pad(self, body, size):
        if self.padding_dim == 0:
            return body.contiguous()
        out = torch.zeros(size, self.ndim, dtype=torch.float32, device=self.device)
        out[:, :body.shape[1]] = body
        return out


class SphereDataset(DistributionDataset):
    def __init__(self, seed, dimension=3, padding_dimension=0, device=None):
        super().__init__(seed, device)
        self.R = 1
        self.dim, self.padding_dim = dimension, padding_dimension
        self.ndim = dimension + padding_dimension

    @property
    def shape(self):
        return (self.ndim,)

    def get_batch(self, size, return_latents=False):
        samps = self._device_batch(size)
        if samps is None:
            g = vrandom.normal(self.get_key(), (size, self.dim), self.device)
            samps = self._pad(g / g.norm(dim=1, keepdim=True), size)
        return (samps, None) if return_latents else samps

    def device_spec(self):
        return 2, None, self.dim, self.dim, self.padding_dim, 0.0

    def score_batch(self, batch):
        real, padding = batch[:, :self.dim], batch[:, self.dim:]
        return {"Sphere Error": ((real.norm(dim=1) - 1) ** 2).mean(), "Padding Error": (padding.norm(dim=1) ** 2).mean()}

    def plot_batch(self, batch, fn):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.hist(batch.norm(dim=1).cpu().numpy(), bins=[0.1 * i for i in range(13)])
        plt.savefig(fn)
        plt.close()


class LinearGaussianDataset(DistributionDataset):
    def __init__(self, seed, dimension=3, intrinsic_dimension=3, padding_dimension=0, var_added=0., device=None):
        super().__init__(seed, device)
        self.dim, self.intrinsic_dim, self.padding_dim = dimension, intrinsic_dimension, padding_dimension
        self.ndim = dimension + padding_dimension
        self.var_added = var_added
        while True:                                   # redraw until full rank (datasets.py:172-178)
            mat = vrandom.normal(self.get_key(), (dimension, intrinsic_dimension), "cpu")
            if int(torch.linalg.matrix_rank(mat)) == min(dimension, intrinsic_dimension):
                break
        self.A = mat.to(self.device)
        self.transformed_cov = self.A @ self.A.T

    @property
    def shape(self):
        return (self.ndim,)

    def get_batch(self, size, return_latents=False):
        Y = self._device_batch(size)
        if Y is None:
            X = vrandom.normal(self.get_key(), (size, self.intrinsic_dim), self.device)
            Y = self._pad(X @ self.A.T, size)
            if self.var_added > 0:
                Y = Y + vrandom.normal(self.get_key(), (size, self.ndim), self.device) * math.sqrt(self.var_added)
        return (Y, None) if return_latents else Y

    def device_spec(self):
        return 0, self.A.contiguous(), self.dim, self.intrinsic_dim, self.padding_dim, float(self.var_added)

    def score_batch(self, batch):
        return {"Squared Norm of padding dimensions": batch[:, self.dim:].square().sum(dim=1).mean()}

    def plot_batch(self, batch, fn):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        b = batch.cpu().numpy()
        if self.dim == 2:
            plt.scatter(b[:, 0], b[:, 1])
        else:
            import numpy as np
            plt.plot(np.sort(np.linalg.norm(b, axis=1)))
            plt.ylabel("Norm of points")
        plt.title(f"Gaussian with dimension {self.dim} and padding {self.padding_dim}")
        plt.savefig(fn)
        plt.close()


class SigmoidDataset(DistributionDataset):
    def __init__(self, seed, dimension=3, padding_dimension=0, device=None):
        super().__init__(seed, device)
        self.dim, self.padding_dim = dimension, padding_dimension
        self.ndim = dimension + padding_dimension + 1
        self.A = vrandom.normal(self.get_key(), (self.dim, 1), "cpu").to(self.device)

    @property
    def shape(self):
        return (self.ndim,)

    def get_batch(self, size, return_latents=False):
        Y = self._device_batch(size)
        if Y is None:
            z = vrandom.normal(self.get_key(), (size, self.dim), self.device)
            Y = torch.zeros(size, self.ndim, dtype=torch.float32, device=self.device)
            Y[:, :self.dim] = z
            Y[:, self.dim] = torch.sigmoid(z @ self.A).squeeze(1)
        return (Y, None) if return_latents else Y

    def device_spec(self):
        return 1, self.A.reshape(-1).contiguous(), self.dim, 1, self.padding_dim, 0.0

    def score_batch(self, batch):
        codomain = (batch[:, :self.dim] @ self.A)
        manifold_error = (batch[:, self.dim] - codomain).square().mean()   # broadcasts (B,) against (B,1) as the reference does
        mse = batch[:, self.dim + 1:].square().sum(dim=1).mean()
        return {"Squared Norm of Padding Dimensions": mse, "Squared Norm of Manifold Dimension": manifold_error}

    def plot_batch(self, batch, fn):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        true_batch = self.get_batch(batch.shape[0])
        for b in (batch, true_batch):
            plt.scatter((b[:, :self.dim] @ self.A).cpu().numpy(), b[:, self.dim].cpu().numpy())
        plt.savefig(fn)
        plt.close()
