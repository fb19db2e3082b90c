"""VAEModel, the reference's per-experiment wrapper (vae.py:15-209): same constructor signature,
attributes (.model, .optimizer, .key, .vae_losses, .epsilon, .current_epsilon, .latent_dimension)
and methods (train_one_batch :123-130, compute_model_stats :132-141, sample_batch :191-201,
model_save_data :203-209).  Warm start (:62-107) and the correlation ratio (:143-179) are out of
scope: the former uses removed jax APIs and no shipped script sets it,
the latter iterates an always-empty list."""
from __future__ import annotations

import math

import torch

from . import random as vrandom
from .model import GenerativeModel
from .networks import VAE, Model
from .optim import Adam


class VAEModel(GenerativeModel):
    def __init__(self, dirname, num_batches, num_epochs, batch_size, learning_rate, layer_sizes,
                 encoder_layer_sizes, state_dict, data_fn, epsilon, tqdm, dataset, latent_dimension,
                 tunable_decoder_var=False, warm_start=False, dataset_name=None, latent_off_dimension=0,
                 exchange=None, global_batch=0, world=1, rank=0, force_generic=False, fast_loop=False, dtype="f32"):
        super().__init__(dirname=dirname, num_batches=num_batches, num_epochs=num_epochs, batch_size=batch_size,
                         learning_rate=learning_rate, latent_distribution="gaussian",
                         latent_dimension=latent_dimension, dataset=dataset, state_dict=state_dict,
                         data_fn=data_fn, tqdm=tqdm)
        if warm_start:
            raise NotImplementedError("-ws/--warm_start is out of scope (reference vae.py:62-107 relies on the "
                                      "removed jax.ops.index_update; no shipped experiment script uses it)")
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
