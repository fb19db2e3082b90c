"""Training-loop layer of the reference (model.py:18-255) for the VAE path: Model /
GenerativeModel with train_distribution (:207-222), compute_stats (:153-168), write_stats
(:195-205), sample_latent (:225-228), save / save_model (:246-255, :85-89) and a WORKING
load()/--state_dict resume (dead code in the reference, :37-43, :91-94)."""
from __future__ import annotations

import os
import pickle
from collections import defaultdict
from copy import deepcopy

import numpy as np
import torch

from . import random as vrandom


class _CheckpointUnpickler(pickle.Unpickler):
    """Unpickler for model.pkl (save_model below / model.py:85-89): nested dicts of numpy arrays plus an int step.
    Only the globals numpy needs to rebuild an ndarray (and OrderedDict) resolve; anything else -- i.e. any pickle
    that would run code on load -- raises instead of executing."""
    _ALLOWED = {
        ("collections", "OrderedDict"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    }

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}: only dict / list / int / float / "
                                         "numpy.ndarray checkpoints written by this package are loaded")
        return super().find_class(module, name)


def load_checkpoint(path):
    with open(path, "rb") as f:
        return _CheckpointUnpickler(f).load()


class Model:
    def __init__(self, dirname, batch_size, learning_rate, state_dict, tqdm=False):
        self.dirname, self.batch_size, self.learning_rate = dirname, batch_size, learning_rate
        self.key = vrandom.PRNGKey(0)                       # model.py:29
        self.state_dict = state_dict
        self.optimizer = self.model = self.state = None
        self.tqdm = tqdm
        self.stats = defaultdict(list)

    def get_key(self):
        self.key, key = vrandom.split(self.key)
        return key

    def load_model(self):
        if self.optimizer is None or self.state_dict is None:
            return
        if isinstance(self.state_dict, (str, os.PathLike)):
            self.state_dict = load_checkpoint(self.state_dict)      # a checkpoint THIS package wrote (save_model)
        self.optimizer.load_state_dict(self.state_dict)
        self.model = self.optimizer.target

    def save_model(self):
        with open(os.path.join(self.dirname, "model.pkl"), "wb") as f:
            pickle.dump(self.optimizer.state_dict(), f)

    def load(self):
        if self.data_fn is not None:
            self.dataset.load(self.data_fn)
        self.load_model()


class GenerativeModel(Model):
    def __init__(self, dirname, num_batches, num_epochs, batch_size, learning_rate, latent_distribution,
                 state_dict, dataset, data_fn, tqdm=False, latent_dimension=None):
        super().__init__(dirname, batch_size, learning_rate, state_dict, tqdm)
        self.num_batches, self.num_epochs = num_batches, num_epochs
        self.latent_distribution = latent_distribution
        self.dataset = dataset
        self.n_plot, self.n_print = 50000, 5000            # model.py:123-124
        self.plot_batch_size = self.print_batch_size = 1000
        self.average_log_likelihoods = []
        self.latent_dim = latent_dimension if latent_dimension else self.dataset.dimension
        self.data_fn = data_fn
        self.epoch_num = 0
        self.batchnum = 0

    def plot(self):
        try:
            import matplotlib.pyplot as plt
            plt.clf()
        except Exception:
            pass

    def plot_epoch(self):
        if getattr(self, "rank", 0) != 0:
            return
        key, self.key = vrandom.split(self.key)
        batch = self.sample_batch(key, self.plot_batch_size)[0]
        fn = os.path.join(self.dirname, f"output_{self.batchnum}.png")
        try:
            self.dataset.plot_batch(batch, fn=fn)
        except ImportError:
            pass                                           # matplotlib is optional here

    def sample_latent(self, key, batch_size):
        """model.py:225-228: z ~ N(0,1)^(B x (latent_dim + dataset.dimension)), drawn on the device."""
        if self.latent_distribution != "gaussian":
            raise NotImplementedError(f"distribution {self.latent_distribution} is not implemented")
        return vrandom.normal(key, (batch_size, self.latent_dim + self.dataset.dimension), self.dataset.device)

    def compute_stats(self):
        key, self.key = vrandom.split(self.key)
        real_batch, latents = self.dataset.get_batch(self.print_batch_size, return_latents=True)
        if latents is None or latents.shape[-1] != self.latent_dim:
            latents = None
        fake_batch, latents = self.sample_batch(key, self.print_batch_size, latents=latents)
        stats = self.compute_model_stats(real_batch, fake_batch, latents)
        score = self.dataset.score_batch(fake_batch)
        if not isinstance(score, dict):
            stats["Average Log Likelihood"] = score
            self.average_log_likelihoods.append(score)
        else:
            stats.update(score)
        return stats

    def write_stats(self, stats):
        message = f"Batch | {self.batchnum}"
        for stat, val in stats.items():
            try:
                fval = float(val)
            except Exception:
                self.stats[stat].append(_to_numpy(val))
                continue
            self.stats[stat].append(fval)
            message += f" | {stat} | {fval:.3f}"
        self._write(message)

    def _write(self, message):
        if getattr(self, "rank", 0) != 0:
            return
        if self.tqdm:
            try:
                from tqdm import tqdm
                tqdm.write(message)
                return
            except ImportError:
                pass
        print(message)

    def train(self):
        self.train_distribution()

    # ---- data-parallel hygiene around the rank-0-only work (plot + save) -------------------------------------------
    def _dp_exchange(self):
        opt = getattr(self, "optimizer", None)
        ex = getattr(opt, "exchange", None)
        return ex if ex is not None and getattr(ex, "world", 1) > 1 else None

    def _dp_sync(self):
        """Every rank waits here after a block only rank 0 executes (matplotlib import, font cache, pickling can take
        seconds): without it the other ranks would enter the next step's in-kernel exchange and spin towards its
        give-up bound while rank 0 is still on the host."""
        ex = self._dp_exchange()
        if ex is not None:
            torch.cuda.synchronize()
            ex.dist.barrier()

    def check_replicas(self):
        """Data-parallel replicas apply the same update to the same sums: parameters and Adam moments must be BITWISE equal
        on every rank.  Compared through an element-wise MAX and MIN over ranks."""
        ex = self._dp_exchange()
        if ex is None:
            return True
        st = self.optimizer.state
        for name, t in (("params", self.model.flat), ("m", st.m), ("v", st.v)):
            hi = t.detach().clone() if ex.dist.get_backend() != "gloo" else t.detach().cpu()
            lo = hi.clone()
            ex.dist.all_reduce(hi, op=ex.dist.ReduceOp.MAX)
            ex.dist.all_reduce(lo, op=ex.dist.ReduceOp.MIN)
            if not torch.equal(hi, lo):
                raise RuntimeError(f"data-parallel replicas diverged: {name} differ between ranks "
                                   f"(max |difference| {float((hi - lo).abs().max()):.3e})")
        self._write(f"Data parallel: replicas identical on {ex.world} ranks after {self.optimizer.state.step} steps")
        return True

    def _dp_check(self):
        """An in-kernel exchange that gave up waiting for a peer has summed a stale granule: the replicas may have
        diverged, so stop loudly (every rank takes the same decision)."""
        ex = self._dp_exchange()
        if ex is None or not ex.in_library:
            return
        torch.cuda.synchronize()
        if not ex._all_ok(not ex.timed_out()):
            raise RuntimeError("data-parallel gradient exchange gave up waiting for a peer (vaek_comm_status): "
                               "replicas may have diverged; aborting")

    def train_distribution(self):
        """model.py:207-222: stats every n_print steps, plot+save every n_plot steps and at the end,
        one dataset batch + one train step per iteration."""
        eval_batch = self.dataset.get_batch(self.print_batch_size)
        score = self.dataset.score_batch(eval_batch)
        if getattr(self, "rank", 0) == 0:
            print(f"Score for real data: { {k: float(v) for k, v in score.items()} if isinstance(score, dict) else score}")
        it = range(self.num_batches)
        if self.tqdm:
            try:
                from tqdm import trange
                it = trange(self.num_batches)
            except ImportError:
                pass
        fast = getattr(self, "fast_loop", False)
        if fast is None:                 # auto: the models vaek_train_steps_gen covers (linear VAEs) take the loop built on it
            fast = self._fast_loop_qualifies()
        if fast:
            return self._train_distribution_fast()
        for self.batchnum in it:
            if self.batchnum % self.n_print == 0:
                self._dp_check()
                self.write_stats(self.compute_stats())
            if self.batchnum % self.n_plot == 0 or self.batchnum == self.num_batches - 1:
                self._dp_check()
                self.plot_epoch()
                self.save()
                self._dp_sync()
            self.train_one_batch(self.dataset.get_batch(self.batch_size))
        self._dp_check()

    def _fast_loop_qualifies(self):
        try:
            from .datasets import DEVICE_DRAW_MAX_DIM
            kind, _, dd, did, _, _ = self.dataset.device_spec()
            if dd > DEVICE_DRAW_MAX_DIM or did > DEVICE_DRAW_MAX_DIM:
                return False
            eng = self.model.module.engine(self.batch_size, self.optimizer.global_batch)
            return bool(eng.supports_train_steps_gen(kind))
        except Exception:
            return False

    def _train_distribution_fast(self):
        """Same schedule (stats every n_print, plot+save every n_plot and at the last step), but the steps in
        between run from a hipGraph with on-device batch generation (trainer.GraphLoop)."""
        from .trainer import GraphLoop
        loop = GraphLoop(self)
        self._graph_loop = loop
        events = sorted(set(list(range(0, self.num_batches, self.n_print)) + list(range(0, self.num_batches, self.n_plot))
                            + [self.num_batches - 1]))
        pos = 0
        for ev in events:
            loop.run(ev - pos)
            pos = ev
            self.batchnum = ev
            loop.check()
            self._dp_check()
            if ev % self.n_print == 0:
                self.write_stats(self.compute_stats())
            if ev % self.n_plot == 0 or ev == self.num_batches - 1:
                self.plot_epoch()
                self.save()
                self._dp_sync()
        loop.run(self.num_batches - pos)
        loop.check()
        self._dp_check()

    def save(self, final=False):
        if getattr(self, "rank", 0) != 0:          # replicas are identical: rank 0 writes losses.npz / model.pkl
            return
        data = self.model_save_data(final=final)
        data["Average Log Likelihood"] = np.array([_to_numpy(a) for a in self.average_log_likelihoods])
        stats = deepcopy(dict(self.stats))
        stats.update({k: _to_numpy(v) for k, v in data.items()})
        np.savez(os.path.join(self.dirname, "losses"), **{k: np.asarray(v, dtype=object) if _ragged(v) else np.asarray(v)
                                                         for k, v in stats.items()})
        self.save_model()
        self.dataset.save(os.path.join(self.dirname, "dataset.pk"))


def _to_numpy(v):
    if torch.is_tensor(v):
        return v.detach().cpu().numpy()
    if isinstance(v, (list, tuple)):
        return [_to_numpy(a) for a in v]
    return v


def _ragged(v):
    try:
        np.asarray(v, dtype=np.float64)
        return False
    except Exception:
        return True
