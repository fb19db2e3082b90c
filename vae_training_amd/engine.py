"""Host-side handle on a libvaek context: flat parameter layout, workspace, raw kernel calls.

Everything here passes raw device pointers of PyTorch-ROCm tensors plus torch's current HIP
stream through the C ABI (include/vaek.h); torch is plumbing (memory, streams), not compute.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch

from . import _lib, layout


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libvaek takes contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, device):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    return t.to(device=device, dtype=torch.float32).contiguous()


class Engine:
    """One libvaek context = one (batch, architecture) shape on one GPU."""

    def __init__(self, batch, data_dim, latent_dim, enc_hidden=(), dec_hidden=(), epsilon=0.0,
                 tunable_decoder_var=False, sigmoid_decoder=False, device=None, world=1, rank=0,
                 global_batch=0, dtype="f32", force_generic=False, fused_impl="auto"):
        if not torch.cuda.is_available():
            raise RuntimeError("vae_training_amd needs an MI355X (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        cfg = _lib.VaekConfig()
        cfg.struct_size = C.sizeof(_lib.VaekConfig)
        cfg.batch, cfg.data_dim, cfg.latent_dim = int(batch), int(data_dim), int(latent_dim)
        enc_hidden, dec_hidden = list(enc_hidden), list(dec_hidden)
        if len(enc_hidden) > _lib.VAEK_MAX_HIDDEN or len(dec_hidden) > _lib.VAEK_MAX_HIDDEN:
            raise ValueError(f"at most {_lib.VAEK_MAX_HIDDEN} hidden layers per network")
        cfg.n_enc_hidden, cfg.n_dec_hidden = len(enc_hidden), len(dec_hidden)
        for i, h in enumerate(enc_hidden):
            cfg.enc_hidden[i] = int(h)
        for i, h in enumerate(dec_hidden):
            cfg.dec_hidden[i] = int(h)
        cfg.sigmoid_decoder = int(bool(sigmoid_decoder))
        cfg.tunable_eps = int(bool(tunable_decoder_var))
        cfg.eps_cli = float(epsilon)
        cfg.dtype = {"f32": _lib.VAEK_F32, "bf16": _lib.VAEK_BF16}[dtype]
        cfg.device = self.device.index
        cfg.world, cfg.rank, cfg.global_batch = int(world), int(rank), int(global_batch)
        self.world, self.rank = int(world), int(rank)
        cfg.force_generic = int(bool(force_generic))
        cfg.reserved[0] = {"auto": 0, "mfma": 0, "valu": 1}[fused_impl]      # which fused linear-VAE kernel
        self.cfg = cfg
        h = C.c_void_p()
        _lib.check(self.lib.vaek_ctx_create(C.byref(cfg), C.byref(h)))
        self.h = h
        n = C.c_int64()
        _lib.check(self.lib.vaek_param_count(h, C.byref(n)))
        self.P = n.value
        _lib.check(self.lib.vaek_grad_len(h, C.byref(n)))
        self.grad_len = n.value
        nl = C.c_int32()
        _lib.check(self.lib.vaek_leaf_count(h, C.byref(nl)))
        self.leaves = OrderedDict()
        buf = C.create_string_buffer(64)
        for i in range(nl.value):
            off, r, c = C.c_int64(), C.c_int32(), C.c_int32()
            _lib.check(self.lib.vaek_leaf_info(h, i, buf, 64, C.byref(off), C.byref(r), C.byref(c)))
            name = buf.value.decode()
            shape = (c.value,) if r.value == 1 and not name.endswith("kernel") else (r.value, c.value)
            self.leaves[name] = (off.value, shape)
        want, want_p = layout.leaves(data_dim, latent_dim, enc_hidden, dec_hidden, sigmoid_decoder, tunable_decoder_var)
        assert want_p == self.P and list(want.items()) == list(self.leaves.items()), "layout.py disagrees with libvaek"
        ws = C.c_size_t()
        _lib.check(self.lib.vaek_workspace_bytes(h, C.byref(ws)))
        self.workspace = torch.empty(max(ws.value, 256), dtype=torch.uint8, device=self.device)
        assert self.workspace.data_ptr() % 256 == 0
        f = C.c_int32()
        _lib.check(self.lib.vaek_uses_fused_path(h, C.byref(f)))
        self.fused = bool(f.value)
        self.batch, self.D, self.L = int(batch), int(data_dim), int(latent_dim)

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.vaek_ctx_destroy(h)
            except Exception:
                pass

    # ---- flat buffers ----------------------------------------------------------------------
    def new_flat(self, n=None):
        return torch.zeros(self.P if n is None else n, dtype=torch.float32, device=self.device)

    def views(self, flat):
        return layout.views(flat, self.leaves)

    # ---- hot path ----------------------------------------------------------------------------
    def train_step(self, params, grads, m, v, step_dev, x, z1, z2, lr):
        _lib.check(self.lib.vaek_train_step(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(step_dev),
                                            _ptr(x), _ptr(z1), _ptr(z2), float(lr), _ptr(self.workspace), _stream()))

    def train_step_gen(self, params, grads, m, v, step_dev, cur, lr, kind, A, dd, did, pad, var_added, nxt, seed, counter,
                       which, tag=0, row0=0):
        """vaek_train_step on the batch `cur` = (x, z1, z2) and the draw of the next batch into `nxt` (its step is
        counter[which], and counter[which ^ 1] = step + 1 is stored) -- on the fused path inside the finalize launch."""
        assert counter.dtype == torch.int32 and counter.numel() == 2 and counter.is_cuda
        _lib.check(self.lib.vaek_train_step_gen(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(step_dev),
                                                _ptr(cur[0]), _ptr(cur[1]), _ptr(cur[2]), float(lr), _ptr(self.workspace),
                                                int(kind), _ptr(A), int(dd), int(did), int(pad), float(var_added),
                                                _ptr(nxt[0]), _ptr(nxt[1]), _ptr(nxt[2]), int(row0), int(seed) & (2**64 - 1),
                                                _ptr(counter), int(which), int(tag), _stream()))

    def supports_train_steps(self):
        f = C.c_int32()
        _lib.check(self.lib.vaek_supports_train_steps(self.h, C.byref(f)))
        return bool(f.value)

    def train_steps(self, params, grads, m, v, step_dev, batches, lr):
        """len(batches) consecutive train steps, batch i = (x, z1, z2) device tensors, software-pipelined over launches
        (vaek_train_steps; linear VAEs only).  Asynchronous; capturable into a hipGraph."""
        n = len(batches)
        arr = lambda k: (C.c_void_p * n)(*[C.c_void_p(b[k].data_ptr()) for b in batches])
        for b in batches:
            assert all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 for t in b)
        xs, z1s, z2s = arr(0), arr(1), arr(2)
        _lib.check(self.lib.vaek_train_steps(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(step_dev), xs, z1s, z2s, n,
                                             float(lr), _ptr(self.workspace), _stream()))

    def supports_train_steps_gen(self, kind):
        f = C.c_int32()
        _lib.check(self.lib.vaek_supports_train_steps_gen(self.h, int(kind), C.byref(f)))
        return bool(f.value)

    def train_steps_gen(self, params, grads, m, v, step_dev, n_steps, lr, kind, A, dd, did, pad, var_added, seed, tag=0, row0=0):
        """n_steps consecutive train steps whose batches are drawn INSIDE the launch (vaek_train_steps_gen): the batch of the step
        that takes the Adam counter from t to t + 1 is bit for bit the one make_batch(..., step = t) would write, but it never
        exists in HBM.  Asynchronous; capturable into a hipGraph."""
        _lib.check(self.lib.vaek_train_steps_gen(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(step_dev), int(kind), _ptr(A),
                                                 int(dd), int(did), int(pad), float(var_added), int(row0), int(seed) & (2**64 - 1), int(tag),
                                                 int(n_steps), float(lr), _ptr(self.workspace), _stream()))

    def plan_train_steps(self, params, grads, m, v, step_dev, batches, lr):
        """The same call with its arguments marshalled once: returns a function that issues vaek_train_steps on these buffers
        again (the pointer arrays, not the data, are frozen) -- for loops that repeat a group of steps, where building three
        ctypes arrays per call would cost more host time than the launch."""
        n = len(batches)
        for b in batches:
            assert all(t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 for t in b)
        arr = lambda k: (C.c_void_p * n)(*[C.c_void_p(b[k].data_ptr()) for b in batches])
        keep = (params, grads, m, v, step_dev, list(batches))                     # the plan keeps its tensors alive
        args = (self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(step_dev), arr(0), arr(1), arr(2), n, C.c_float(lr), _ptr(self.workspace))
        fn, check = self.lib.vaek_train_steps, _lib.check

        def run(_keep=keep):
            check(fn(*args, _stream()))
        return run

    def moment_len(self):
        """doubles in the second-moment image of a batch (0: the moment form does not cover this model)."""
        n = C.c_int64()
        _lib.check(self.lib.vaek_train_steps_moment_len(self.h, C.byref(n)))
        return int(n.value)

    def moments(self, x, z1, z2, M):
        """M (float64 device tensor of moment_len()) <- the second-moment image of this rank's batch shard (vaek_train_steps_moments)."""
        assert M.dtype == torch.float64 and M.numel() >= self.moment_len()
        _lib.check(self.lib.vaek_train_steps_moments(self.h, _ptr(x), _ptr(z1), _ptr(z2), _ptr(M), _ptr(self.workspace), _stream()))

    def moments_update(self, params, grads, m, v, step_dev, M, lr):
        """loss, gradients, Adam from the (globally summed) moment image (vaek_train_steps_update)."""
        _lib.check(self.lib.vaek_train_steps_update(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(step_dev), _ptr(M), float(lr),
                                                    _ptr(self.workspace), _stream()))

    def train_steps_gave_up(self):
        """Synchronous: True if a bounded wait inside vaek_train_steps' persistent launch ever expired."""
        f = C.c_int32()
        _lib.check(self.lib.vaek_train_steps_status(self.h, _ptr(self.workspace), C.byref(f)))
        self.train_steps_status_word = f.value & 0xffffffff
        return f.value != 0

    def grads_only(self, params, grads, step_dev, x, z1, z2):
        _lib.check(self.lib.vaek_train_step_grads_only(self.h, _ptr(params), _ptr(grads), _ptr(step_dev), _ptr(x),
                                                       _ptr(z1), _ptr(z2), _ptr(self.workspace), _stream()))

    def buckets(self):
        """[(offset, count)] of the gradient buckets in the order the backward pass completes them."""
        n = C.c_int32()
        _lib.check(self.lib.vaek_bucket_count(self.h, C.byref(n)))
        out = []
        for i in range(n.value):
            off, cnt = C.c_int64(), C.c_int64()
            _lib.check(self.lib.vaek_bucket_info(self.h, i, C.byref(off), C.byref(cnt)))
            out.append((off.value, cnt.value))
        return out

    def grads_bucketed(self, params, grads, step_dev, x, z1, z2, events):
        """Like grads_only, but records events[i] (torch.cuda.Event, already created on this device) the moment
        bucket i of `grads` is final."""
        arr = (C.c_void_p * len(events))(*[C.c_void_p(e.cuda_event) for e in events])
        _lib.check(self.lib.vaek_train_step_grads_bucketed(self.h, _ptr(params), _ptr(grads), _ptr(step_dev), _ptr(x), _ptr(z1),
                                                           _ptr(z2), arr, _ptr(self.workspace), _stream()))

    def apply(self, params, grads, m, v, step_dev, lr):
        _lib.check(self.lib.vaek_train_step_apply(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v),
                                                  _ptr(step_dev), float(lr), _stream()))

    def loss_eval(self, params, x, z1, z2):
        out = torch.empty(4, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.vaek_loss_eval(self.h, _ptr(params), _ptr(x), _ptr(z1), _ptr(z2), _ptr(out),
                                           _ptr(self.workspace), _stream()))
        return out

    def forward(self, params, x, z1, z2, sampling=False, eps=0.0, want_mu=True):
        rows = z1.shape[0]
        x_hat = torch.empty(rows, self.D, dtype=torch.float32, device=self.device)
        mu = torch.empty(rows, self.L, dtype=torch.float32, device=self.device) if want_mu else None
        _lib.check(self.lib.vaek_forward(self.h, _ptr(params), _ptr(x), _ptr(z1), _ptr(z2), int(bool(sampling)),
                                         float(eps), _ptr(x_hat), _ptr(mu), rows, _ptr(self.workspace), _stream()))
        return x_hat, mu

    # ---- on-device inputs (K7) and the loss ring ------------------------------------------------
    def make_batch(self, kind, A, dd, did, pad, var_added, rows, seed, step_dev=None, step=0, tag=0, row0=0,
                   want_x=True, want_z=True, out=None, counter=None, which=0):
        """x[rows,D] (or None), z1[rows,L], z2[rows,D] drawn by libvaek's Philox kernel.  With `counter` (device
        int32[2]) the step is counter[which] and the kernel stores counter[which ^ 1] = step + 1 (vaek_make_batch_next)."""
        if out is None:
            Dx = dd + pad + (1 if kind == 1 else 0)
            x = torch.empty(rows, Dx, dtype=torch.float32, device=self.device) if want_x else None
            z1 = torch.empty(rows, self.L, dtype=torch.float32, device=self.device) if want_z else None
            z2 = torch.empty(rows, self.D, dtype=torch.float32, device=self.device) if want_z else None
        else:
            x, z1, z2 = out
        if counter is not None:
            assert counter.dtype == torch.int32 and counter.numel() == 2 and counter.is_cuda
            _lib.check(self.lib.vaek_make_batch_next(self.h, int(kind), _ptr(A), int(dd), int(did), int(pad), float(var_added),
                                                     _ptr(x), _ptr(z1), _ptr(z2), int(rows), int(row0),
                                                     int(seed) & (2**64 - 1), _ptr(counter), int(which), int(tag), _stream()))
            return x, z1, z2
        _lib.check(self.lib.vaek_make_batch(self.h, int(kind), _ptr(A), int(dd), int(did), int(pad), float(var_added),
                                            _ptr(x), _ptr(z1), _ptr(z2), int(rows), int(row0), int(seed) & (2**64 - 1),
                                            _ptr(step_dev), int(step), int(tag), _stream()))
        return x, z1, z2

    def rng_fill(self, n, seed, step=0, tag=0, bits=False):
        out_n = torch.empty(n, dtype=torch.float32, device=self.device)
        out_u = torch.empty(n, dtype=torch.int32, device=self.device) if bits else None
        _lib.check(self.lib.vaek_rng_fill(self.h, _ptr(out_n), _ptr(out_u), int(n), int(seed) & (2**64 - 1), int(step),
                                          int(tag), _stream()))
        return (out_n, out_u) if bits else out_n

    def set_loss_history(self, buf):
        self._loss_hist = buf                      # keep it alive
        _lib.check(self.lib.vaek_set_loss_history(self.h, _ptr(buf), 0 if buf is None else buf.numel()))

    # ---- in-process kernel timing (bench.py) ------------------------------------------------
    def profile_begin(self, max_records=4096):
        _lib.check(self.lib.vaek_profile_begin(self.h, int(max_records)))

    def profile_report(self):
        import json
        buf = C.create_string_buffer(8192)
        _lib.check(self.lib.vaek_profile_report(self.h, buf, 8192))
        return json.loads(buf.value.decode())

    def measure_peaks(self, copy_bytes=1 << 30, reps=5):
        """Achievable HBM bandwidth (float4 stream copy, read + write bytes) and f32 / bf16 MFMA rates measured on
        this device with the library's own micro-benchmarks."""
        src = torch.empty(copy_bytes, dtype=torch.uint8, device=self.device).random_(0, 255)
        dst = torch.empty_like(src)
        scratch = torch.zeros(16, dtype=torch.float32, device=self.device)
        flops = C.c_double()
        st = _stream()
        plan = [("microbench_mfma_f32", 0, 20000, 1), ("microbench_mfma_bf16", 1, 40000, 1)]     # ~1 ms each, one wave per SIMD
        for _ in range(2):                                   # warm-up
            _lib.check(self.lib.vaek_microbench_copy(self.h, _ptr(src), _ptr(dst), copy_bytes, st))
            for _, kind, iters, wps in plan:
                _lib.check(self.lib.vaek_microbench_mfma(self.h, kind, iters, wps, _ptr(scratch), C.byref(flops), st))
        torch.cuda.synchronize()
        self.profile_begin(64)
        fl = {}
        for _ in range(reps):
            _lib.check(self.lib.vaek_microbench_copy(self.h, _ptr(src), _ptr(dst), copy_bytes, st))
            for name, kind, iters, wps in plan:
                _lib.check(self.lib.vaek_microbench_mfma(self.h, kind, iters, wps, _ptr(scratch), C.byref(flops), st))
                fl[name] = flops.value
        torch.cuda.synchronize()
        rep = self.profile_report()
        out = {"hbm_copy_GBps": 2.0 * copy_bytes * rep["microbench_stream_copy"]["count"] / (rep["microbench_stream_copy"]["total_ms"] * 1e-3) / 1e9}
        for name, _, _, _ in plan:
            out[name.replace("microbench_", "") + "_TFLOPs"] = fl[name] * rep[name]["count"] / (rep[name]["total_ms"] * 1e-3) / 1e12
        return out

    def measure_launch_floor(self, n=100, reps=20):
        """Launch-to-launch interval (us) of a kernel that does nothing, replayed from a hipGraph chain of n launches:
        what any two-kernel step pays before it computes anything."""
        import time
        launch = lambda k: _lib.check(self.lib.vaek_microbench_launch(self.h, 0, 1, k, None, None, _stream()))
        launch(4)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                launch(n)
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (n * reps) * 1e6

    # ---- building blocks -----------------------------------------------------------------------
    def dense_fwd(self, x, w, b, relu=False):
        rows, n_in = x.shape
        n_out = w.shape[1]
        y = torch.empty(rows, n_out, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.vaek_dense_fwd(self.h, _ptr(x), _ptr(w), _ptr(b), _ptr(y), rows, n_in, n_out,
                                           int(relu), _stream()))
        return y

    def dense_fwd_reparam(self, x, w, b, z1, logvar_e):
        """mu = x @ w + b, samples = mu + exp(logvar_e / 2) * z1 (networks.py:72-74) in one block."""
        rows, n_in = x.shape
        n_out = w.shape[1]
        mu = torch.empty(rows, n_out, dtype=torch.float32, device=self.device)
        samples = torch.empty_like(mu)
        _lib.check(self.lib.vaek_dense_fwd_reparam(self.h, _ptr(x), _ptr(w), _ptr(b), _ptr(mu), _ptr(samples), _ptr(z1), _ptr(logvar_e),
                                                   rows, n_in, n_out, _stream()))
        return mu, samples

    def reparam_bwd(self, d_samples, mu, z1, logvar_e, batch_total=0, out=None):
        """In place d_samples -> d_mu; returns d logvar_e (reparameterisation + KL parts)."""
        rows, L = mu.shape
        g = torch.empty(L, dtype=torch.float32, device=self.device) if out is None else out
        _lib.check(self.lib.vaek_reparam_bwd(self.h, _ptr(d_samples), _ptr(mu), _ptr(z1), _ptr(logvar_e), _ptr(g), rows, L,
                                             int(batch_total), _ptr(self.workspace), _stream()))
        return g

    def dense_bwd_dx(self, dy, w, x_post=None, relu=False, out=None, accumulate=False):
        rows, n_out = dy.shape
        n_in = w.shape[0]
        dx = torch.empty(rows, n_in, dtype=torch.float32, device=self.device) if out is None else out
        _lib.check(self.lib.vaek_dense_bwd_dx(self.h, _ptr(dy), _ptr(w), _ptr(x_post), _ptr(dx), rows, n_in, n_out,
                                              int(relu), int(accumulate), _stream()))
        return dx

    def dense_bwd_dw(self, x, dy):
        rows, n_in = x.shape
        n_out = dy.shape[1]
        dwb = torch.empty(n_in + 1, n_out, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.vaek_dense_bwd_dw(self.h, _ptr(x), _ptr(dy), _ptr(dwb), rows, n_in, n_out,
                                              _ptr(self.workspace), _stream()))
        return dwb

    def elbo_fwd_bwd(self, x, x_hat_lin, x_hat_sig, z2, mu, logvar_e, eps, batch_total=0, grads=True, eps_param=None):
        """eps_param (a one-element device tensor): eps = eps_param[0] * eps, read on the device (vaek_elbo_fwd_bwd_dev)."""
        rows, D = x.shape
        L = mu.shape[1]
        out4 = torch.empty(4, dtype=torch.float32, device=self.device)
        d_lin = torch.empty_like(x_hat_lin) if grads else None
        d_sig = torch.empty_like(x_hat_sig) if (grads and x_hat_sig is not None) else None
        if eps_param is not None:
            _lib.check(self.lib.vaek_elbo_fwd_bwd_dev(self.h, _ptr(x), _ptr(x_hat_lin), _ptr(x_hat_sig), _ptr(z2), _ptr(mu),
                                                      _ptr(logvar_e), _ptr(eps_param), float(eps), _ptr(d_lin), _ptr(d_sig), _ptr(out4),
                                                      rows, D, L, int(batch_total), _ptr(self.workspace), _stream()))
        else:
            _lib.check(self.lib.vaek_elbo_fwd_bwd(self.h, _ptr(x), _ptr(x_hat_lin), _ptr(x_hat_sig), _ptr(z2), _ptr(mu),
                                                  _ptr(logvar_e), float(eps), _ptr(d_lin), _ptr(d_sig), _ptr(out4),
                                                  rows, D, L, int(batch_total), _ptr(self.workspace), _stream()))
        return out4, d_lin, d_sig

    def adam_step(self, params, grads, m, v, lr, step=None, step_dev=None, grad_scale=1.0):
        _lib.check(self.lib.vaek_adam_step(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(),
                                           float(lr), int(step or 0), _ptr(step_dev), float(grad_scale), _stream()))
