"""Worker of tests/test_gpu_run_dp.py::test_conv_vae_two_ranks_equal_one_rank_on_the_whole_batch (two ranks rehearsed on one GPU over
gloo): every rank takes its half of a batch through ConvVAE.train_step with the gradient all-reduce; rank 0 also runs ONE replica on
the whole batch.  The summed shard gradients must be the whole batch's gradient (the same per-sample arithmetic, the batch
reductions in a different order: 1e-4 of a leaf's max-abs), the replicas must end bitwise identical."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import conv_vae_oracle as CO                    # noqa: E402  (test infrastructure: parameter initialisation only)
from vae_training_amd.conv_vae import ConvVAE               # noqa: E402
from vae_training_amd.parallel import GradExchange          # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    size, widths, L, Bg = 32, (8, 16, 16, 32), 6, 16
    cfg = CO.ConvConfig(size, widths, L, -1.5, True)
    p = CO.init_params(cfg, seed=4)
    rng = np.random.default_rng(11)
    x = rng.random((Bg, size, size, 1)).astype(np.float32)
    z1 = rng.standard_normal((Bg, L)).astype(np.float32)
    z2 = rng.standard_normal((Bg, size, size, 1)).astype(np.float32)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda().contiguous()

    def replica(B, w):
        net = ConvVAE(B, size, widths, L, -1.5, True, world=w)
        bufs = [net.new_flat() for _ in range(4)]
        for name in net.leaves:
            net.view(bufs[0], name).copy_(dev(p[name]))
        return net, bufs, torch.zeros(1, dtype=torch.int32, device="cuda")

    per = Bg // world
    net, (params, grads, m, v), step = replica(per, world)
    exch = GradExchange(net.eng, dist, mode="rccl")
    sl = slice(rank * per, (rank + 1) * per)
    out4 = net.train_step(params, grads, m, v, step, dev(x[sl]), dev(z1[sl]), dev(z2[sl]), 1e-3, all_reduce=exch.all_reduce)
    g_dp, loss_dp = grads.clone(), float(out4[0])
    for _ in range(3):
        net.train_step(params, grads, m, v, step, dev(x[sl]), dev(z1[sl]), dev(z2[sl]), 1e-3, all_reduce=exch.all_reduce)
    chk = torch.stack([params.double().sum(), params.double().abs().sum()]).cpu()
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool(torch.equal(lo, hi))
    worst = 0.0
    if rank == 0:
        one, (p1, g1, m1, v1), s1 = replica(Bg, 1)
        o1 = one.train_step(p1, g1, m1, v1, s1, dev(x), dev(z1), dev(z2), 1e-3)
        for name in one.leaves:
            a, b = net.view(g_dp, name), one.view(g1, name)
            worst = max(worst, float((a - b).abs().max() / (b.abs().max() + 1e-30)))
        worst = max(worst, abs(loss_dp - float(o1[0])) / abs(float(o1[0])))
    print(f"RESULT rank={rank} replicas_identical={same} worst_rel={worst:.2e} steps={int(step[0])}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
