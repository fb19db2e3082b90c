"""-m gpu: the library's peer-to-peer gradient exchange, rehearsed with W ranks sharing ONE GPU
(HIP IPC between processes, tagged granules, both banks, epochs = Adam steps).  Cross-device
coherence over xGMI cannot be exercised on a one-GPU box; the protocol, the IPC plumbing, the
in-finalize fusion and replica bit-identity can."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import elbo_oracle as O

pytestmark = pytest.mark.gpu


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
        cfg = O.Config(12, 20, (), (), -1.0, True, "linear_gaussian")
        B, lr, steps = 1024, 1e-3, 4
        rng = np.random.default_rng(0)
        r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
        p = {k: r32(v) for k, v in O.init_params(cfg, seed=0).items()}
        xs = [r32(rng.standard_normal((B, 12))) for _ in range(steps)]
        zs = [r32(rng.standard_normal((B, 32))) for _ in range(steps)]
        lo, hi = shard_rows(B, world, rank)
        eng = Engine(hi - lo, 12, 20, (), (), -1.0, True, False, world=world, rank=rank, global_batch=B)
        assert eng.fused
        ex = GradExchange(eng, dist, mode="p2p")
        assert ex.in_library and ex.mode == "p2p"
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()
        params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        st = O.adam_init(p)
        worst = 0.0
        for s in range(steps):
            z1, z2 = O.split_latents(zs[s], 20)
            p, st, loss_ref = O.train_step(cfg, p, st, xs[s], z1, z2, lr)          # full batch on the oracle
            eng.train_step(params, grads, m, v, step, dev(xs[s][lo:hi]), dev(z1[lo:hi]), dev(z2[lo:hi]), lr)
            torch.cuda.synchronize()
            worst = max(worst, abs(float(grads[eng.P]) - loss_ref) / abs(loss_ref))
        perr = float(np.max(np.abs(params.cpu().numpy().astype(np.float64) - O.flatten(cfg, p))))
        digest = torch.tensor(params.cpu().numpy().view(np.int32).astype(np.int64).sum().reshape(1))
        allg = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(allg, digest)
        q.put((rank, worst, perr, all(int(a) == int(allg[0]) for a in allg), ex.timed_out(), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:            # report instead of hanging the parent
        import traceback
        q.put((rank, 1.0, 1.0, False, True, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_exchange_inside_finalize(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, lerr, perr, same, timed_out, tb in res:
        assert tb is None, tb
        assert lerr <= 1e-5 and perr <= 0.02 * 1e-3 and same and not timed_out, (rank, lerr, perr, same, timed_out)


def _silent_peer_worker(rank, world, port, q):
    try:
        import ctypes as C
        import time
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from vae_training_amd import _lib
        from vae_training_amd.engine import Engine
        from vae_training_amd.parallel import GradExchange
        eng = Engine(256, 12, 20, (), (), -1.0, True, False, world=world, rank=rank, global_batch=512)
        ex = GradExchange(eng, dist, mode="p2p")
        assert ex.in_library
        out = None
        if rank == 0:                      # rank 1 never joins these two exchanges
            buf = torch.ones(64, device="cuda")
            times = []
            for _ in range(2):
                t0 = time.perf_counter()
                _lib.check(eng.lib.vaek_comm_allreduce(eng.h, C.c_void_p(buf.data_ptr()), 64,
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            out = (times, ex.timed_out())
        dist.barrier()
        q.put((rank, out, None))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, None, traceback.format_exc()))


def test_p2p_exchange_gives_up_on_a_silent_peer_and_then_fails_fast():
    """A peer that never stores its granules must cost seconds, not minutes: the first exchange gives up within the
    bounded spin (comm_dev.h: kSpinLimit), raises the status word the host polls before falling back to RCCL
    (parallel.GradExchange.timed_out), and later exchanges on that rank do not wait at all."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_silent_peer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in procs:
        rank, out, tb = q.get(timeout=240)
        assert tb is None, tb
        res[rank] = out
    for p in procs:
        p.join(timeout=60)
    (t_first, t_second), timed_out = res[0]
    assert timed_out
    assert 0.2 < t_first < 30.0, t_first          # ~3 s at ~0.4 us per poll; generous bounds, but not minutes
    assert t_second < 0.5 * t_first and t_second < 1.0, (t_first, t_second)


def _steps_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from vae_training_amd.engine import Engine
        from vae_training_amd.parallel import GradExchange, shard_rows
        cfg = O.Config(12, 20, (), (), -1.0, True, "linear_gaussian")
        B, lr, steps = 1536, 1e-3, 7
        rng = np.random.default_rng(0)
        r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
        p = {k: r32(v) for k, v in O.init_params(cfg, seed=0).items()}
        xs = [r32(rng.standard_normal((B, 12))) for _ in range(steps)]
        zs = [r32(rng.standard_normal((B, 32))) for _ in range(steps)]
        lo, hi = shard_rows(B, world, rank)
        eng = Engine(hi - lo, 12, 20, (), (), -1.0, True, False, world=world, rank=rank, global_batch=B)
        assert not eng.supports_train_steps()              # no communicator yet: the moments cannot be exchanged
        ex = GradExchange(eng, dist, mode="p2p")
        assert ex.in_library and eng.supports_train_steps()
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()
        params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        ring = torch.zeros(steps + 4, dtype=torch.float32, device="cuda")
        eng.set_loss_history(ring)
        st, want = O.adam_init(p), []
        batches = []
        for s in range(steps):
            z1, z2 = O.split_latents(zs[s], 20)
            p, st, loss_ref = O.train_step(cfg, p, st, xs[s], z1, z2, lr)          # full batch on the oracle
            want.append(loss_ref)
            batches.append((dev(xs[s][lo:hi]), dev(z1[lo:hi]), dev(z2[lo:hi])))
        eng.train_steps(params, grads, m, v, step, batches[:4], lr)                # two launches: 4 + 3 steps
        eng.train_steps(params, grads, m, v, step, batches[4:], lr)
        torch.cuda.synchronize()
        # the stand-alone all-reduce keeps its epoch on the device: captured once, replayed three times, right every time
        import ctypes as C
        from vae_training_amd import _lib
        buf = torch.zeros(300, device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side):
                _lib.check(eng.lib.vaek_comm_allreduce(eng.h, C.c_void_p(buf.data_ptr()), 300, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.current_stream().wait_stream(side)
        for rep in range(3):
            buf.copy_(torch.arange(300, device="cuda") * 0.5 + (rank + 1) * (rep + 1))
            dist.barrier()
            gr.replay()
            torch.cuda.synchronize()
            want_buf = torch.arange(300, device="cuda") * 0.5 * world + sum(r + 1 for r in range(world)) * (rep + 1)
            assert torch.equal(buf, want_buf.to(torch.float32)), (rep, buf[:3], want_buf[:3])
        assert not ex.timed_out()
        got = ring.cpu().numpy()[:steps].astype(np.float64)
        worst = float(np.max(np.abs(got - np.array(want)) / np.abs(want)))
        perr = float(np.max(np.abs(params.cpu().numpy().astype(np.float64) - O.flatten(cfg, p))))
        digest = torch.tensor(params.cpu().numpy().view(np.int32).astype(np.int64).sum().reshape(1))
        allg = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(allg, digest)
        q.put((rank, worst, perr, all(int(a) == int(allg[0]) for a in allg), eng.train_steps_gave_up() or int(step.item()) != steps, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, 1.0, 1.0, False, True, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_steps_exchange_their_moments_across_ranks(world):
    """vaek_train_steps under data parallelism: each rank streams its shard, the reducers exchange the moment matrix over the
    P2P communicator inside the persistent launch, every rank applies the same update -- losses and parameters follow the
    oracle's FULL-batch training, replicas end bitwise identical, no bounded wait expires (ranks rehearsed on one GPU)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_steps_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, lerr, perr, same, bad, tb in res:
        assert tb is None, tb
        assert lerr <= 1e-5 and perr <= 0.02 * 1e-3 * 7 and same and not bad, (rank, lerr, perr, same, bad)


def _moments_rccl_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from vae_training_amd.engine import Engine
        from vae_training_amd.parallel import GradExchange, shard_rows
        cfg = O.Config(12, 20, (), (), -1.0, True, "linear_gaussian")
        B, lr, steps = 1536, 1e-3, 6
        rng = np.random.default_rng(0)
        r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
        p = {k: r32(v) for k, v in O.init_params(cfg, seed=0).items()}
        lo, hi = shard_rows(B, world, rank)
        eng = Engine(hi - lo, 12, 20, (), (), -1.0, True, False, world=world, rank=rank, global_batch=B)
        ex = GradExchange(eng, dist, mode="rccl")                  # no P2P communicator: the collective carries the moment matrix
        assert not ex.in_library and not eng.supports_train_steps() and eng.moment_len() == 6 * 256
        dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()
        params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        st, worst = O.adam_init(p), 0.0
        for s in range(steps):
            x = r32(rng.standard_normal((B, 12))); z1, z2 = O.split_latents(r32(rng.standard_normal((B, 32))), 20)
            p, st, loss_ref = O.train_step(cfg, p, st, x, z1, z2, lr)              # full batch on the oracle
            ex.moments_step(params, grads, m, v, step, dev(x[lo:hi]), dev(z1[lo:hi]), dev(z2[lo:hi]), lr)
            worst = max(worst, abs(float(grads[eng.P]) - loss_ref) / abs(loss_ref))
        perr = float(np.max(np.abs(params.cpu().numpy().astype(np.float64) - O.flatten(cfg, p))))
        digest = torch.tensor(params.cpu().numpy().view(np.int32).astype(np.int64).sum().reshape(1))
        allg = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(allg, digest)
        q.put((rank, worst, perr, all(int(a) == int(allg[0]) for a in allg), int(step.item()) != steps, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, 1.0, 1.0, False, True, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_moment_matrix_summed_by_the_collective_without_p2p(world):
    """Data parallel WITHOUT the P2P communicator (GradExchange mode "rccl"; gloo here): vaek_train_steps_moments on each shard, one
    all-reduce of the float64 moment image, vaek_train_steps_update everywhere -- losses and parameters follow the oracle's
    FULL-batch training (networks.py:87-101), replicas end bitwise identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_moments_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, lerr, perr, same, bad, tb in res:
        assert tb is None, tb
        assert lerr <= 1e-5 and perr <= 0.02 * 1e-3 * 6 and same and not bad, (rank, lerr, perr, same, bad)
