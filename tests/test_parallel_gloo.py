"""World-size-2 gloo tests of the data-parallel exchange (CPU): shard rows, per-shard gradients with
the GLOBAL divisor, SUM all-reduce through vae_training_amd.parallel.GradExchange == full-batch
gradient; replicas then apply identical Adam updates (KAT-DP)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from oracle import elbo_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vae_training_amd.parallel import GradExchange, shard_rows
    cfg = O.Config(7, 6, (16,), (16,), -3.0, True, "sigmoid")
    rng = np.random.default_rng(0)                       # identical data on both ranks; each takes its shard
    B = 64
    p = O.init_params(cfg, seed=1)
    x = rng.standard_normal((B, 7)); z1 = rng.standard_normal((B, 6)); z2 = rng.standard_normal((B, 7))
    lo, hi = shard_rows(B, world, rank)
    loss_w, g_w = O.loss_and_grad(cfg, p, x[lo:hi], z1[lo:hi], z2[lo:hi], batch_total=B)
    buf = torch.tensor(np.concatenate([O.flatten(cfg, g_w), [loss_w, 0, 0, 0]]), dtype=torch.float64)
    ex = GradExchange(None, dist, mode="rccl")
    assert ex.mode == "rccl" and not ex.in_library and ex.world == world
    ex.all_reduce(buf)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    want = np.concatenate([O.flatten(cfg, g), [loss, 0, 0, 0]])
    err = float(np.max(np.abs(buf.numpy() - want)) / np.max(np.abs(want)))
    # identical update on every replica
    p2, _ = O.adam_update(p, O.unflatten(cfg, buf.numpy()[:cfg.n_params()]), O.adam_init(p), 1e-3)
    digest = torch.tensor([float(np.sum(O.flatten(cfg, p2)))], dtype=torch.float64)
    both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, digest)
    q.put((rank, err, float(both[0]) == float(both[1])))
    dist.destroy_process_group()


def test_shard_rows():
    from vae_training_amd.parallel import shard_rows
    assert [shard_rows(65536, 8, r) for r in (0, 7)] == [(0, 8192), (57344, 65536)]
    try:
        shard_rows(10, 4, 0)
        assert False
    except ValueError:
        pass


def test_two_rank_gradient_exchange_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, same in res:
        assert err < 1e-12 and same, (rank, err, same)
