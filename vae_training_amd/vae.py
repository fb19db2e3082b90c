"""VAEModel, the reference's per-experiment wrapper (vae.py:15-209): same constructor signature,
attributes (.model, .optimizer, .key, .vae_losses, .epsilon, .current_epsilon, .latent_dimension)
and methods (train_one_batch :123-130, compute_model_stats :132-141, sample_batch :191-201,
model_save_data :203-209), and the warm start of -ws (:62-107: host-side initialisation only, SURVEY.md 8f rank 4's sibling).
The correlation ratio (:143-179) is out of scope: it iterates an always-empty list."""
from __future__ import annotations

import math

import torch

from . import random as vrandom
from .model import GenerativeModel
from .networks import VAE, Model
from .optim import Adam


def warm_start_params(params, key, dataset, dataset_name, latent_dimension, data_size, latent_off_dimension=0):
    """The reference's -ws initialisation (vae.py:62-107) on the nested parameter dict, in place: the one-layer encoder / decoder
    start at the data manifold's own maps plus small noise.
      sigmoid (:63-79):          Encoder / Decoder kernels = identity with the [dim + 1:, dim + 1:] block zeroed, + 0.1 N(0, 1);
                                 SigDecoder kernel = 0.1 N(0, 1); epsilon_p = -3 on [dim + 1:], 0 before, + 0.1 N(0, 1).
      linear_gaussian (:81-106): Decoder kernel = ([A | N(0, 1) extra columns | 0] stacked over zero padding rows + 0.01 N(0, 1))^T,
                                 Encoder kernel = (pinv(A) padded with zero rows / columns + 0.01 N(0, 1))^T,
                                 epsilon_p = -3 on the first A.shape[1] + latent_off_dimension entries, 0 after, + 0.1 N(0, 1).
    Every draw uses `key` itself, as the reference does (self.key, never split there).  Other datasets: untouched, like the
    reference.  Shapes that the reference's concatenations would reject (hidden layers; an intrinsic dimension != dimension)
    raise ValueError here."""
    L = latent_dimension
    normal = lambda shape: vrandom.normal(key, shape, "cpu")

    def put(path, value):
        leaf = params
        for name in path[:-1]:
            leaf = leaf[name]
        if tuple(leaf[path[-1]].shape) != tuple(value.shape):
            raise ValueError(f"warm start: {'/'.join(path)} has shape {tuple(leaf[path[-1]].shape)}, the warm-start value {tuple(value.shape)} "
                             "(it needs the one-layer encoder and decoder of the reference's -ws experiments)")
        leaf[path[-1]] = value.to(torch.float32)

    if dataset_name == "sigmoid":
        dim, full = dataset.dim, int(dataset.dimension)
        if L != full:
            raise ValueError(f"warm start (sigmoid): latent_dim {L} must equal the data dimension {full} (vae.py:64)")
        eye = torch.eye(L)
        eye[dim + 1:, dim + 1:] = 0.0
        var = torch.zeros(L)
        var[dim + 1:] = -3.0
        put(("Decoder", "FC0", "kernel"), eye + 0.1 * normal((L, full)))
        put(("SigDecoder", "FC0", "kernel"), 0.1 * normal((L, full)))
        put(("epsilon_p",), var + 0.1 * normal((L,)))
        put(("Encoder", "FC0", "kernel"), eye + 0.1 * normal((full, L)))
    elif dataset_name == "linear_gaussian":
        dim, off = dataset.dim, int(latent_off_dimension)
        A = dataset.A.detach().to("cpu", torch.float32)
        if not dim + off < L:
            raise ValueError(f"warm start (linear_gaussian): dimension {dim} + latent_off_dimension {off} must be < latent_dim {L} (vae.py:82)")
        if A.shape[1] != dim:
            raise ValueError("warm start (linear_gaussian): the reference's concatenations need intrinsic_dimension == dimension")
        dec = torch.cat([A, normal((dim, off)), torch.zeros(dim, L - dim - off)], dim=1)
        dec = torch.cat([dec, torch.zeros(data_size - dim, L)], dim=0) + 0.01 * normal((int(dataset.dimension), L))
        put(("Decoder", "FC0", "kernel"), dec.T.contiguous())
        enc = torch.cat([torch.linalg.pinv(A), torch.zeros(L - dim, dim)], dim=0)
        enc = torch.cat([enc, torch.zeros(L, data_size - dim)], dim=1) + 0.01 * normal((L, int(dataset.dimension)))
        put(("Encoder", "FC0", "kernel"), enc.T.contiguous())
        var = torch.zeros(L)
        var[:A.shape[1] + off] = -3.0
        put(("epsilon_p",), var + 0.1 * normal((L,)))
    return params


class VAEModel(GenerativeModel):
    def __init__(self, dirname, num_batches, num_epochs, batch_size, learning_rate, layer_sizes,
                 encoder_layer_sizes, state_dict, data_fn, epsilon, tqdm, dataset, latent_dimension,
                 tunable_decoder_var=False, warm_start=False, dataset_name=None, latent_off_dimension=0,
                 exchange=None, global_batch=0, world=1, rank=0, force_generic=False, fast_loop=False, dtype="f32"):
        super().__init__(dirname=dirname, num_batches=num_batches, num_epochs=num_epochs, batch_size=batch_size,
                         learning_rate=learning_rate, latent_distribution="gaussian",
                         latent_dimension=latent_dimension, dataset=dataset, state_dict=state_dict,
                         data_fn=data_fn, tqdm=tqdm)
        self.fast_loop = fast_loop
        self.epsilon = epsilon
        self.current_epsilon = epsilon
        self.latent_dimension = latent_dimension
        data_size = int(math.prod(dataset.shape))
        enc = parse_layer_sizes(encoder_layer_sizes) + [latent_dimension]      # vae.py:53
        dec = parse_layer_sizes(layer_sizes) + [data_size]                     # vae.py:54
        vae_key, self.key = vrandom.split(self.key)
        device = dataset.device if getattr(dataset, "device", None) is not None else None
        vae_module = VAE.partial(epsilon=epsilon, encoder_layer_sizes=enc, decoder_layer_sizes=dec,
                                 tunable_decoder_var=tunable_decoder_var, dataset_name=dataset_name,
                                 device=device, world=world, rank=rank, force_generic=force_generic, dtype=dtype)
        _, initial_params = vae_module.init_by_shape(vae_key, [(data_size,), (latent_dimension,), (data_size,)])
        if warm_start:
            warm_start_params(initial_params, self.key, dataset, dataset_name, latent_dimension, data_size, latent_off_dimension)
        self.model = Model(vae_module, initial_params)
        self.optimizer = Adam(learning_rate=self.learning_rate).create(self.model, exchange=exchange,
                                                                       global_batch=global_batch)
        self.vae_losses, self.var_enc, self.var_dec = [], [], []
        self.gt_eigen, self.ht_eigen = [], []
        self.params_and_gradients, self.correlation_ratios = [], []
        if self.state_dict is not None:
            self.load()

    def _latent_pair(self, key, batch_size):
        """z1[B,L], z2[B,D] of model.py:227 / vae.py:127-128 as two contiguous tensors from ONE Philox launch
        (csrc/rng.hip); sample_latent() below still returns the reference's single (B, L+D) array."""
        eng = self.model.module.engine(batch_size, self.optimizer.global_batch)
        self._latent_draws = getattr(self, "_latent_draws", 0) + 1
        _, z1, z2 = eng.make_batch(0, None, 1, 1, eng.D - 1, 0.0, batch_size, seed=key[0] ^ key[1], step=self._latent_draws,
                                   tag=2, want_x=False)
        return z1, z2

    def sample_latent(self, key, batch_size):
        if self.dataset.device.type != "cuda":
            return super().sample_latent(key, batch_size)
        z1, z2 = self._latent_pair(key, batch_size)
        return torch.cat([z1, z2], dim=1)

    def train_one_batch(self, batch):
        batch = batch.reshape(batch.shape[0], -1)
        latent_batch_key, self.key = vrandom.split(self.key)
        z1, z2 = self._latent_pair(latent_batch_key, self.batch_size)
        self.optimizer, self.model, vae_loss = VAE.train_step(self.optimizer, batch, z1, z2, self.epsilon)
        self.vae_losses.append(vae_loss)

    def compute_model_stats(self, real_batch, fake_batch, latents):
        z1 = latents[..., :self.latent_dimension].contiguous()
        z2 = latents[..., self.latent_dimension:].contiguous()
        vae_loss, dkl, mse, logvar_e, epsilon = VAE.loss(self.model, real_batch, z1, z2, self.epsilon)
        self.vae_losses.append(vae_loss)
        self.var_enc.append(logvar_e.clone())
        self.var_dec.append(epsilon.clone() if torch.is_tensor(epsilon) else epsilon)
        self.current_epsilon = epsilon.clone() if torch.is_tensor(epsilon) else epsilon
        return {"VAE Loss": vae_loss, "KL divergence": dkl, "mse": mse}

    def sample_batch(self, key, batch_size, latents=None):
        z = latents if latents is not None else self.sample_latent(key, batch_size)
        z1 = z[..., :self.latent_dimension].contiguous()
        z2 = z[..., self.latent_dimension:].contiguous()
        x_hat, _, _, _ = self.model(None, z1, z2, sampling=True, epsilon=self.current_epsilon)
        return x_hat, z

    def model_save_data(self, final=False):
        loop = getattr(self, "_graph_loop", None)
        if loop is not None:                # fast loop: train losses live in the device ring, eval losses in the list
            self.vae_losses_train = loop.losses()
        data = {"VAE Loss": self.vae_losses if loop is None else list(self.vae_losses) + list(self.vae_losses_train), "Decoder Variance": self.var_dec, "Encoder Variance": self.var_enc}
        if final:
            data["Correlation Ratio"] = self.correlation_ratios       # always empty, as in the reference
        return data


def parse_layer_sizes(spec):
    """ "512|512" -> [512, 512]; "" -> [] (vae.py:53-54)."""
    return [int(s) for s in spec.split("|")] if spec != "" else []
