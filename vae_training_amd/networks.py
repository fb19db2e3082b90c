"""Host-side mirror of the reference's networks.py for the ELBO hot path.

Same names, argument meaning and calling convention as /root/reference/networks.py
(VAE.partial / init_by_shape as used at vae.py:57-60, Model(...) at vae.py:112, VAE.train_step
:87-101, VAE.loss :103-113, model(batch, z1, z2[, sampling=True, epsilon=...]) :61-84), but every
FLOP runs in libvaek's HIP kernels through the C ABI; tensors are PyTorch-ROCm device buffers.

Calling convention stays functional (``self.optimizer, self.model, loss = VAE.train_step(...)``,
vae.py:129): new wrapper objects are returned, but they SHARE the flat device buffers, which the
kernels update in place -- the objects passed in must be considered consumed, exactly as the
reference's rebinding treats them.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

from . import random as vrandom
from . import layout
from .engine import Engine, _f32


def _sizes(spec):
    return [int(s) for s in spec]


class FullyConnectedNetwork:
    """Shape bookkeeping of networks.py:26-44 (Dense stack, relu between layers).  The arithmetic
    lives in libvaek (vaek_dense_fwd & co.); this class only names layers like the reference."""

    @staticmethod
    def get_layer_name(i):
        return f"FC{i}"

    @staticmethod
    def param_shapes(fan_in, layer_sizes):
        shapes, k = OrderedDict(), fan_in
        for i, n in enumerate(layer_sizes):
            shapes[f"FC{i}"] = {"kernel": (k, n), "bias": (n,)}
            k = n
        return shapes


class VAEModule:
    """What ``VAE.partial(...)`` returns: the static configuration (vae.py:57-59)."""

    def __init__(self, epsilon, encoder_layer_sizes, decoder_layer_sizes, tunable_decoder_var=False,
                 dataset_name=None, device=None, dtype="f32", world=1, rank=0, force_generic=False):
        self.epsilon = float(epsilon)
        self.encoder_layer_sizes = _sizes(encoder_layer_sizes)      # last entry = latent dim (vae.py:53)
        self.decoder_layer_sizes = _sizes(decoder_layer_sizes)      # last entry = data dim   (vae.py:54)
        self.tunable_decoder_var = bool(tunable_decoder_var)
        self.dataset_name = dataset_name
        self.latent_dim = self.encoder_layer_sizes[-1]
        self.data_dim = self.decoder_layer_sizes[-1]
        self.device, self.dtype = device, dtype
        self.world, self.rank = world, rank
        self.force_generic = force_generic
        self._engines = {}
        self.leaves, self.n_params = layout.leaves(self.data_dim, self.latent_dim, self.encoder_layer_sizes[:-1],
                                                   self.decoder_layer_sizes[:-1], dataset_name == "sigmoid",
                                                   self.tunable_decoder_var)

    @property
    def torch_device(self):
        if self.device is not None:
            return torch.device(self.device) if not isinstance(self.device, int) else torch.device("cuda", self.device)
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    def engine(self, batch, global_batch=0) -> Engine:
        """One libvaek context per batch size (train batch, eval batch of 1000 rows, ...)."""
        key = (int(batch), int(global_batch))
        eng = self._engines.get(key)
        if eng is None:
            eng = Engine(batch, self.data_dim, self.latent_dim, self.encoder_layer_sizes[:-1],
                         self.decoder_layer_sizes[:-1], self.epsilon, self.tunable_decoder_var,
                         self.dataset_name == "sigmoid", device=self.torch_device.index if self.torch_device.type == "cuda" else None,
                         world=self.world, rank=self.rank,
                         global_batch=global_batch, dtype=self.dtype, force_generic=self.force_generic)
            self._engines[key] = eng
        return eng

    def init_by_shape(self, key, shapes):
        """vae.py:60.  Returns (output shapes, initial params) like flax's init_by_shape; params are a
        nested dict of float32 tensors (kernels lecun-normal = truncated normal on [-2, 2] times
        sqrt(1/fan_in)/0.8796..., biases 0, epsilon_p = epsilon = 1: networks.py:69-71)."""
        (d,), (l,), (d2,) = [tuple(s) for s in shapes]
        assert d == self.data_dim and l == self.latent_dim and d2 == self.data_dim
        nets = [("Encoder", d, self.encoder_layer_sizes), ("Decoder", l, self.decoder_layer_sizes)]
        if self.dataset_name == "sigmoid":
            nets.append(("SigDecoder", l, self.decoder_layer_sizes))
        params = OrderedDict()
        keys = vrandom.split(key, sum(len(s) for _, _, s in nets))
        ki = 0
        for name, fan_in, sizes in nets:
            params[name] = OrderedDict()
            for lname, shp in FullyConnectedNetwork.param_shapes(fan_in, sizes).items():
                k = shp["kernel"][0]
                w = vrandom.truncated_normal(keys[ki], -2.0, 2.0, shp["kernel"]) * (math.sqrt(1.0 / k) / 0.87962566103423978)
                ki += 1
                params[name][lname] = OrderedDict(kernel=w, bias=torch.zeros(shp["bias"]))
        params["epsilon_p"] = torch.ones(l)
        if self.tunable_decoder_var:
            params["epsilon"] = torch.ones(1)
        return ((d,), (l,), (l,), ()), params


class Model:
    """flax.nn.Model stand-in (vae.py:112): module + parameters.  ``params`` is the reference's
    nested dict ({'Encoder': {'FC0': {'kernel', 'bias'}}, ..., 'epsilon_p', ['epsilon']}) whose leaves
    are VIEWS into one flat device buffer (include/vaek.h layout)."""

    def __init__(self, module: VAEModule, params, _flat=None):
        self.module = module
        if _flat is None:
            # parameters live on the GPU; without one (host-logic tests) the tree is still usable on the
            # CPU, but every compute entry point below raises: there is no CPU fallback
            _flat = torch.zeros(module.n_params, dtype=torch.float32, device=module.torch_device)
            _copy_tree(params, layout.views(_flat, module.leaves))
        self.flat = _flat
        self.params = layout.views(_flat, module.leaves)

    def __call__(self, batch, z1, z2, sampling=False, epsilon=None):
        """VAE.apply, networks.py:61-84 -> (x_hat, mu, logvar_e, epsilon)."""
        m = self.module
        z1 = _f32(z1, self.flat.device); z2 = _f32(z2, self.flat.device)
        rows = z1.shape[0]
        eng = m.engine(rows)
        if sampling:
            eps_in = m.epsilon if epsilon is None else float(torch.as_tensor(epsilon).reshape(-1)[0])
            x_hat, _ = eng.forward(self.flat, None, z1, z2, sampling=True, eps=eps_in, want_mu=False)
            return x_hat, 0, 0, epsilon if epsilon is not None else m.epsilon
        x = _f32(batch, self.flat.device).reshape(rows, -1)
        x_hat, mu = eng.forward(self.flat, x, z1, z2)
        eps = self.params["epsilon"] * m.epsilon if m.tunable_decoder_var else m.epsilon
        return x_hat, mu, self.params["epsilon_p"], eps


def _copy_tree(src, dst):
    for k, v in dst.items():
        if isinstance(v, dict):
            _copy_tree(src[k], v)
        else:
            v.copy_(torch.as_tensor(src[k], dtype=torch.float32).reshape(v.shape))


class VAE:
    """Namespace with the reference's static entry points."""

    @staticmethod
    def partial(**kw) -> VAEModule:
        return VAEModule(**kw)

    @staticmethod
    def train_step(optimizer, batch, z1, z2, epsilon=None):
        """networks.py:87-101 -> (optimizer, optimizer.target, vae_loss).  `epsilon` is accepted and
        ignored exactly as in the reference (shadowed at :92; the model's epsilon was bound by
        VAE.partial).  The loss is a 0-dim device tensor; nothing synchronises (vae.py:130)."""
        model = optimizer.target
        dev = model.flat.device
        x = _f32(batch, dev)
        x = x.reshape(x.shape[0], -1)
        z1 = _f32(z1, dev); z2 = _f32(z2, dev)
        eng = model.module.engine(x.shape[0], optimizer.global_batch)
        st = optimizer.state
        lr = optimizer.optimizer_def.learning_rate
        if optimizer.exchange is None or optimizer.exchange.in_library:
            eng.train_step(model.flat, st.grads, st.m, st.v, st.step_dev, x, z1, z2, lr)
        elif eng.moment_len() > 0:
            # linear VAE over RCCL: the batch's second-moment matrix (additive over the shards) is what the collective sums --
            # vaek_train_steps' arithmetic without the P2P communicator
            optimizer.exchange.moments_step(model.flat, st.grads, st.m, st.v, st.step_dev, x, z1, z2, lr)
        elif eng.fused:
            eng.grads_only(model.flat, st.grads, st.step_dev, x, z1, z2)
            optimizer.exchange.all_reduce(st.grads)
            eng.apply(model.flat, st.grads, st.m, st.v, st.step_dev, lr)
        else:
            # layer-by-layer model over RCCL: per-layer gradient buckets all-reduced on a side stream while the
            # backward GEMMs of the earlier layers still run
            optimizer.exchange.overlapped_grads(model.flat, st.grads, st.step_dev, x, z1, z2)
            eng.apply(model.flat, st.grads, st.m, st.v, st.step_dev, lr)
        st.step += 1
        new_model = Model(model.module, None, _flat=model.flat)
        new_opt = optimizer._rebound(new_model)
        return new_opt, new_model, st.grads[eng.P].clone()

    @staticmethod
    def loss(model, batch, z1, z2, epsilon=None):
        """networks.py:103-113 -> (loss.mean(), Dkl.mean(), mse.mean(), logvar_e, epsilon)."""
        dev = model.flat.device
        x = _f32(batch, dev)
        x = x.reshape(x.shape[0], -1)
        eng = model.module.engine(x.shape[0])
        out = eng.loss_eval(model.flat, x, _f32(z1, dev), _f32(z2, dev))
        m = model.module
        eps = model.params["epsilon"] * m.epsilon if m.tunable_decoder_var else m.epsilon
        return out[0], out[1], out[2], model.params["epsilon_p"], eps
