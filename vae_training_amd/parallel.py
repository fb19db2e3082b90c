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


class GradExchange:
    def __init__(self, engine, dist, mode="auto"):
        self.engine, self.dist = engine, dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.in_library = False
        if mode in ("auto", "p2p") and engine is not None and self.world > 1:
            try:
                self._init_p2p()
                self.in_library = True
            except Exception as e:                      # p2p is an optimisation; RCCL is always correct
                if mode == "p2p":
                    raise
                self._p2p_error = str(e)
        self.mode = "p2p" if self.in_library else "rccl"

    def _init_p2p(self):
        eng, dist = self.engine, self.dist
        lib = eng.lib
        n = C.c_size_t()
        _lib.check(lib.vaek_comm_buffer_bytes(eng.h, C.byref(n)))
        if n.value == 0:
            raise RuntimeError("library built without the p2p communicator")
        # a dedicated hipMalloc allocation (torch's caching allocator sub-allocates; IPC needs the base)
        self.comm_buf = torch.zeros(n.value, dtype=torch.uint8, device=eng.device)
        handle = (C.c_uint8 * 64)()
        _lib.check(lib.vaek_comm_export(eng.h, C.c_void_p(self.comm_buf.data_ptr()), handle))
        mine = torch.tensor(list(handle), dtype=torch.uint8, device=eng.device)
        allh = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allh, mine)
        flat = torch.cat(allh).cpu().numpy().tobytes()
        buf = (C.c_uint8 * len(flat)).from_buffer_copy(flat)
        _lib.check(lib.vaek_comm_init(eng.h, C.c_void_p(self.comm_buf.data_ptr()), buf))
        dist.barrier()

    def all_reduce(self, grads: torch.Tensor):
        """SUM over ranks, in place."""
        if self.world == 1:
            return grads
        self.dist.all_reduce(grads, op=self.dist.ReduceOp.SUM)
        return grads
