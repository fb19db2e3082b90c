"""Data-parallel gradient exchange for the ELBO train step.

The reference is single-device; data parallelism is this build's addition.  Every rank holds a
full replica of params + Adam state and a shard of the minibatch; kernels pre-divide by the
GLOBAL batch, so one SUM all-reduce of the flat gradient buffer (P floats + loss/Dkl/mse in the
4 trailing slots) makes every replica apply the identical Adam update.

Two transports:
  * "rccl": torch.distributed all_reduce (backend "nccl" IS RCCL on ROCm) over xGMI, between
    vaek_train_step_grads_only and vaek_train_step_apply.  Also runs on gloo/CPU tensors, which is
    how the world_size-2 tests exercise this module without a GPU.
  * "p2p":  the library's one-shot peer-to-peer all-reduce over IPC-mapped peer buffers
    (vaek_comm_*), fused into vaek_train_step; the 2 KB gradient of the metric workload is far
    below RCCL's launch latency.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def shard_rows(n_rows: int, world: int, rank: int):
    """Rows [lo, hi) of the global batch owned by `rank` (equal shards)."""
    if n_rows % world:
        raise ValueError(f"global batch {n_rows} is not divisible by world size {world}")
    per = n_rows // world
    return rank * per, (rank + 1) * per


P2P_MAX_FLOATS = 1 << 16      # above this the exchange is bandwidth-, not latency-bound: RCCL's job


class GradExchange:
    def __init__(self, engine, dist, mode="auto"):
        self.engine, self.dist = engine, dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.in_library = False
        self.p2p_error = None
        want_p2p = mode == "p2p" or (mode == "auto" and engine is not None and engine.grad_len <= P2P_MAX_FLOATS)
        if want_p2p and engine is not None and self.world > 1:
            self.in_library = self._setup_p2p()
            if not self.in_library:
                _lib.check(engine.lib.vaek_comm_destroy(engine.h))
                if mode == "p2p":
                    raise RuntimeError(f"p2p gradient exchange unavailable: {self.p2p_error or 'failed on another rank'}")
        self.mode = "p2p" if self.in_library else "rccl"

    def _coll_device(self):
        return torch.device("cpu") if self.dist.get_backend() == "gloo" else self.engine.device

    def _all_ok(self, ok):
        """Every rank takes the same decision; every rank executes the same collectives whatever failed locally."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._coll_device())
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def _guard(self, fn):
        try:
            fn()
            return True
        except Exception as e:                          # p2p is an optimisation; RCCL is always correct
            self.p2p_error = str(e)
            return False

    def _setup_p2p(self):
        eng, dist = self.engine, self.dist
        lib = eng.lib
        handle = (C.c_uint8 * 64)()
        ok = self._guard(lambda: _lib.check(lib.vaek_comm_create(eng.h, handle)))
        mine = torch.tensor(list(handle), dtype=torch.uint8, device=self._coll_device())
        allh = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allh, mine)
        if not self._all_ok(ok):
            return False
        flat = torch.cat(allh).cpu().numpy().tobytes()
        buf = (C.c_uint8 * len(flat)).from_buffer_copy(flat)
        ok = self._guard(lambda: _lib.check(lib.vaek_comm_init(eng.h, buf)))
        if not self._all_ok(ok):
            return False
        dist.barrier()
        ok = False
        try:
            ok = self._selftest()
        except Exception as e:
            self.p2p_error = str(e)
        return self._all_ok(ok)

    def _selftest(self):
        """Three stand-alone all-reduces (both granule banks) of rank-dependent values against the sum
        formed locally in the same rank order; exact equality and no spin give-up required."""
        eng = self.engine
        n = min(eng.grad_len, 4096)
        idx = torch.arange(n, dtype=torch.float32, device=eng.device)
        for rnd in range(3):
            contrib = lambda r: (idx * 0.001 + (r + 1) * (rnd + 1)).to(torch.float32)
            buf = contrib(eng.cfg.rank).contiguous()
            want = torch.zeros(n, dtype=torch.float32, device=eng.device)
            for r in range(self.world):
                want = want + contrib(r)
            _lib.check(eng.lib.vaek_comm_allreduce(eng.h, C.c_void_p(buf.data_ptr()), n,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            if not torch.equal(buf, want):
                self.p2p_error = f"self-test round {rnd}: wrong sum"
                return False
        return not self.timed_out()

    def timed_out(self):
        t = C.c_int32()
        _lib.check(self.engine.lib.vaek_comm_status(self.engine.h, C.byref(t)))
        return bool(t.value)

    # ---- RCCL path with communication overlapped with the backward pass (layer-by-layer models) --------
    def overlapped_grads(self, params, grads, step_dev, x, z1, z2):
        """vaek_train_step_grads_bucketed + one all-reduce per gradient bucket on a side stream: the all-reduce
        of the decoder's last layer runs while the encoder is still in its dW / dX GEMMs (the order RCCL sees
        is the order the backward pass finishes the buckets in, identical on every rank).  On return the
        current stream has waited for every exchange: `grads` is the global sum."""
        eng = self.engine
        if not hasattr(self, "_buckets"):
            self._buckets = eng.buckets()
            self._events = [torch.cuda.Event() for _ in self._buckets]
            for e in self._events:
                e.record()                                  # materialise the hipEvent_t handles
            self._comm_stream = torch.cuda.Stream(device=eng.device)
        main = torch.cuda.current_stream()
        eng.grads_bucketed(params, grads, step_dev, x, z1, z2, self._events)
        for (off, cnt), ev in zip(self._buckets, self._events):
            self._comm_stream.wait_event(ev)
            with torch.cuda.stream(self._comm_stream):
                self.all_reduce(grads[off:off + cnt])
        main.wait_stream(self._comm_stream)
        return grads

    # ---- linear VAEs without the P2P communicator: the batch's second-moment matrix summed by the collective --------------------
    def moments_step(self, params, grads, m, v, step_dev, x, z1, z2, lr):
        """One data-parallel train step of a linear VAE through its sufficient statistic (csrc/linear_moments.hip): every rank
        forms the moment image of its shard, ONE all-reduce sums the 12 KB float64 image (instead of the flat gradient: the
        statistic is what the batch mean of networks.py:97-98 makes additive), and the same update runs everywhere --
        vaek_train_steps' arithmetic with a host collective where the persistent launch exchanges granules.  Replicas stay
        bitwise identical (every rank receives the same sum)."""
        eng = self.engine
        if not hasattr(self, "_M"):
            self._M = torch.zeros(eng.moment_len(), dtype=torch.float64, device=eng.device)
        eng.moments(x, z1, z2, self._M)
        self.all_reduce(self._M)
        eng.moments_update(params, grads, m, v, step_dev, self._M, lr)

    def all_reduce(self, grads: torch.Tensor):
        """SUM over ranks, in place (RCCL; gloo for CPU tensors and one-GPU rehearsals)."""
        if self.world == 1:
            return grads
        if grads.is_cuda and self.dist.get_backend() == "gloo":
            host = grads.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM)
            grads.copy_(host)
            return grads
        self.dist.all_reduce(grads, op=self.dist.ReduceOp.SUM)
        return grads
