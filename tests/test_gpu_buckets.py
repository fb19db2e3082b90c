"""-m gpu: gradient buckets of the layer-by-layer path (vaek_train_step_grads_bucketed) and the
overlapped exchange built on them (parallel.GradExchange.overlapped_grads)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import elbo_oracle as O
from tests.gpu_util import dev, engine_for, host, random_problem, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,dk,B", [
    (O.Config(7, 6, (64, 32), (48,), -3.0, True, "sigmoid"), dict(name="sigmoid", seed=69, dd=3, pad=3), 700),
    (O.Config(6, 70, (24,), (24,), -3.0, True, "sphere"), dict(name="sphere", seed=69, dd=3, pad=3), 300),     # wide latent tail
    (O.Config(12, 20, (), (), -1.0, True, "linear_gaussian"), dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=9), 512),   # fused: 1 bucket
])
def test_buckets_cover_the_gradient_and_match_the_monolithic_path(cfg, dk, B):
    p, x, z1, z2 = random_problem(cfg, dk, B)
    eng = engine_for(cfg, B)
    buckets = eng.buckets()
    cover = np.zeros(eng.grad_len, dtype=int)
    for off, cnt in buckets:
        cover[off:off + cnt] += 1
    assert (cover == 1).all()                                    # a partition of [0, P + 4)
    if eng.fused:
        assert len(buckets) == 1
    else:
        n_layers = len(cfg.enc_sizes) + len(cfg.dec_sizes) * (2 if cfg.sigmoid else 1)
        assert len(buckets) == n_layers + 1 and buckets[-1][0] == eng.leaves["epsilon_p"][0]
        assert buckets[0][0] == eng.leaves[f"Decoder/FC{len(cfg.dec_sizes) - 1}/kernel"][0]   # backward order
    params = dev(O.flatten(cfg, p)); step = torch.zeros(1, dtype=torch.int32, device="cuda")
    g_mono = eng.new_flat(eng.grad_len); g_b = eng.new_flat(eng.grad_len)
    eng.grads_only(params, g_mono, step, dev(x), dev(z1), dev(z2))
    events = [torch.cuda.Event() for _ in buckets]
    for e in events:
        e.record()
    torch.cuda.synchronize()
    eng.grads_bucketed(params, g_b, step, dev(x), dev(z1), dev(z2), events)
    events[-1].synchronize()
    assert all(e.query() for e in events)
    assert rel_err(host(g_b), host(g_mono)) <= 2e-6
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    assert abs(host(g_b)[eng.P] - loss) <= 1e-5 * abs(loss) and rel_err(host(g_b)[:eng.P], O.flatten(cfg, g)) <= 2e-5


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from vae_training_amd.engine import Engine
        from vae_training_amd.parallel import GradExchange, shard_rows
        cfg = O.Config(7, 6, (64, 32), (48,), -3.0, True, "sigmoid")
        B, lr = 512, 1e-3
        rng = np.random.default_rng(0)
        r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
        p = {k: r32(v) for k, v in O.init_params(cfg, seed=0).items()}
        x = r32(rng.standard_normal((B, 7))); z = r32(rng.standard_normal((B, 13)))
        z1, z2 = O.split_latents(z, 6)
        lo, hi = shard_rows(B, world, rank)
        eng = Engine(hi - lo, 7, 6, (64, 32), (48,), -3.0, True, True, world=world, rank=rank, global_batch=B)
        ex = GradExchange(eng, dist, mode="rccl")
        d = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()
        params = d(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        ex.overlapped_grads(params, grads, step, d(x[lo:hi]), d(z1[lo:hi]), d(z2[lo:hi]))
        eng.apply(params, grads, m, v, step, lr)
        torch.cuda.synchronize()
        loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
        p2, _ = O.adam_update(p, g, O.adam_init(p), lr)
        gerr = float(np.max(np.abs(grads.cpu().numpy()[:eng.P] - O.flatten(cfg, g))) / np.max(np.abs(O.flatten(cfg, g))))
        perr = float(np.max(np.abs(params.cpu().numpy() - O.flatten(cfg, p2))))
        q.put((rank, abs(float(grads[eng.P]) - loss) / abs(loss), gerr, perr, None))
        dist.barrier(); dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, 1, 1, 1, traceback.format_exc()))


def test_overlapped_exchange_two_ranks_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, lerr, gerr, perr, tb in res:
        assert tb is None, tb
        assert lerr <= 1e-5 and gerr <= 2e-5 and perr <= 0.02 * 1e-3, (rank, lerr, gerr, perr)
