"""Worker of tests/test_gpu_run_dp.py::test_rank0_slow_plot_does_not_break_the_p2p_exchange: the CLI's data-parallel
model (in-kernel P2P exchange) with a plot/save block on rank 0 that takes longer than the exchange's spin bound
(~3 s, csrc/comm_dev.h).  Launched by torch.distributed.run; prints one RESULT line per rank."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from vae_training_amd import run as vrun  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = vrun.parse_arguments(["slow", "--dataset", "linear_gaussian", "--encoder_layer_sizes", "", "--layer_sizes", "",
                                 "--latent_dim", "20", "--padding_dim", "9", "-dd", "3", "--num_batches", "30", "--batch_size", "512",
                                 "--epsilon", "-1", "-tdv", "-ds", "2", "-lr", "1e-3", "--comm", "p2p"])
    args.device = 0
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    ds = vrun.get_dataset(args.dataset, args.dataset_seed, args.padding_dim, args.batch_size, args, world, rank)
    m = vrun.get_model(args, ds, out, dist)
    assert m.optimizer.exchange.in_library
    m.n_plot, m.n_print = 10, 10
    if rank == 0:
        slow = m.plot_epoch

        def plot_epoch():
            time.sleep(4.5)          # longer than the in-kernel exchange waits for a peer
            slow()
        m.plot_epoch = plot_epoch
    m.train()
    ok = m.check_replicas()
    print(f"RESULT rank={rank} replicas_identical={ok} timed_out={m.optimizer.exchange.timed_out()} steps={m.optimizer.state.step}",
          flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
