"""Command line of the reference (run.py:8-43, :350-369), flag for flag: `python run.py NAME
--dataset linear_gaussian ...` trains a VAE with the HIP kernels and leaves data/NAME/{args.json,
losses.npz, model.pkl, output_*.png}.  Flags the reference parses but never reads on the VAE path
(--num_epochs, --padding_type, -ii, -ufc, -wsl, -off, -ws) are accepted and inert, except -ws
which is rejected (out of scope).  Additions: --device, --force_generic."""
from __future__ import annotations

import argparse


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("name", help="The name of the experiment and output directory.")
    p.add_argument("--num_batches", dest="num_batches", type=int, default=15000, help="Number of batches to train on.")
    p.add_argument("--num_epochs", dest="num_epochs", type=int, default=10000)
    p.add_argument("--batch_size", dest="batch_size", type=int, default=100)
    p.add_argument("-lr", "--learning_rate", dest="learning_rate", type=float, default=0.0001)
    p.add_argument("--padding_dim", type=int, dest="padding_dim", default=0)
    p.add_argument("-ow", dest="overwrite", action="store_true")
    # the reference's default '4gaussian' is not one of its own choices (run.py:18-19): always pass --dataset
    p.add_argument("--dataset", dest="dataset", default="4gaussian", choices=["sphere", "linear_gaussian", "sigmoid"])
    p.add_argument("--layer_sizes", dest="layer_sizes", default="512|512",
                   help="Decoder MLP widths separated by pipes, e.g. 512|512|512; \"\" = linear")
    p.add_argument("--encoder_layer_sizes", dest="encoder_layer_sizes", default="512|512",
                   help="Encoder MLP widths separated by pipes; \"\" = linear")
    p.add_argument("--latent_dim", dest="latent_dimension", type=int, default=100)
    p.add_argument("-nojit", dest="nojit", action="store_true", help="accepted for compatibility (kernels are always eager launches)")
    p.add_argument("--padding_type", dest="padding_type", default="none", choices=["zero", "gaussian", "none"])
    p.add_argument("-ds", "--dataset_seed", dest="dataset_seed", type=int, default=69)
    p.add_argument("--state_dict", dest="state_dict", default=None)
    p.add_argument("--data_fn", dest="data_fn", default=None)
    p.add_argument("-ws", "--warm_start", action="store_true")
    p.add_argument("-ii", "--initialize_inverse", action="store_true")
    p.add_argument("-ufc", "--use_fred_covariance", action="store_true")
    p.add_argument("-e", "--epsilon", type=float, default=0.)
    p.add_argument("-tdv", dest="tunable_decoder_var", action="store_true")
    p.add_argument("-dn", "--dataset_noise", type=float, default=0.)
    p.add_argument("-dd", "--dataset_dimension", type=int, default=3)
    p.add_argument("-wsl", "--warm_start_linear", action="store_true")
    p.add_argument("-did", "--dataset_intrinsic_dimension", type=int, default=3)
    p.add_argument("-off", "--latent_off_dimension", type=int, default=1)
    p.add_argument("--device", type=int, default=None, help="HIP device ordinal (default: current)")
    p.add_argument("--force_generic", action="store_true", help="layer-by-layer kernels even where a fused path exists")
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="Dense GEMM arithmetic (bf16: wide layers on bf16 MFMA)")
    p.add_argument("--dist_backend", default="nccl", choices=["nccl", "gloo"],
                   help="process-group backend when launched with torch.distributed.run (gloo: one-GPU rehearsal)")
    p.add_argument("--comm", default="auto", choices=["auto", "rccl", "p2p"],
                   help="data-parallel gradient exchange: in-kernel peer-to-peer granules (small models) or RCCL all-reduce "
                        "of per-layer buckets overlapped with the backward pass")
    p.add_argument("--fast_loop", action="store_true", default=None,
                   help="run the steps between stats/plots with nothing per step on the host (trainer.py): linear VAEs through "
                        "vaek_train_steps_gen (up to 64 steps per persistent launch, batches drawn inside it), other models from a "
                        "hipGraph with on-device Philox batches.  Default: on for the models vaek_train_steps_gen covers")
    p.add_argument("--no_fast_loop", dest="fast_loop", action="store_false",
                   help="the reference's loop shape: one get_batch + one train_step call per iteration (model.py:221-222)")
    return p


def parse_arguments(argv=None):
    args = build_parser().parse_args(argv)
    args.model = "VAE"
    args.latent_distribution = "gaussian"
    args.tqdm = True
    return args


def get_dataset(name, seed, padding_dimension, batch_size, args, world=1, rank=0):
    from . import random as vrandom
    from .datasets import LinearGaussianDataset, SigmoidDataset, SphereDataset
    dev = None if getattr(args, "device", None) is None else f"cuda:{args.device}"
    ds = None
    if name == "sphere":
        ds = SphereDataset(seed, dimension=args.dataset_dimension, padding_dimension=args.padding_dim, device=dev)
    elif name == "linear_gaussian":
        ds = LinearGaussianDataset(seed, dimension=args.dataset_dimension,
                                   intrinsic_dimension=args.dataset_intrinsic_dimension,
                                   padding_dimension=args.padding_dim, var_added=args.dataset_noise, device=dev)
    elif name == "sigmoid":
        ds = SigmoidDataset(seed, dimension=args.dataset_dimension, padding_dimension=args.padding_dim, device=dev)
    if ds is not None and world > 1:
        # same manifold (A was drawn from the seed above) on every rank, a different sample stream per rank
        ds.key = vrandom.split(ds.key, world)[rank]
    return ds


def get_model(args, dataset, output_dir, dist=None):
    from . import random as vrandom
    from .vae import VAEModel
    if args.model != "VAE":
        raise NameError(f"model {args.model!r}: only the VAE branch exists (run.py:250-268)")
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    if world > 1:
        return _get_model_dp(args, dataset, output_dir, dist, world, rank)
    return VAEModel(dirname=output_dir, batch_size=args.batch_size, learning_rate=args.learning_rate, dataset=dataset,
                    num_batches=args.num_batches, num_epochs=args.num_epochs, layer_sizes=args.layer_sizes,
                    encoder_layer_sizes=args.encoder_layer_sizes, state_dict=args.state_dict, data_fn=args.data_fn,
                    epsilon=args.epsilon, tqdm=args.tqdm, latent_dimension=args.latent_dimension,
                    tunable_decoder_var=args.tunable_decoder_var, warm_start=args.warm_start,
                    dataset_name=args.dataset, latent_off_dimension=args.latent_off_dimension,
                    force_generic=getattr(args, "force_generic", False), fast_loop=getattr(args, "fast_loop", False),
                    dtype=getattr(args, "dtype", "f32"))


def _get_model_dp(args, dataset, output_dir, dist, world, rank):
    """Data parallel (this build's addition; the reference is single-device): --batch_size is the GLOBAL
    batch, every rank trains on batch_size / world rows of its own sample stream, gradients are summed
    over ranks (parallel.GradExchange: P2P inside the finalize kernel, or RCCL overlapped with the
    backward pass) and every replica applies the identical Adam update."""
    from . import random as vrandom
    from .parallel import GradExchange, shard_rows
    from .vae import VAEModel
    lo, hi = shard_rows(args.batch_size, world, rank)
    m = VAEModel(dirname=output_dir, batch_size=hi - lo, learning_rate=args.learning_rate, dataset=dataset,
                 num_batches=args.num_batches, num_epochs=args.num_epochs, layer_sizes=args.layer_sizes,
                 encoder_layer_sizes=args.encoder_layer_sizes, state_dict=args.state_dict, data_fn=args.data_fn,
                 epsilon=args.epsilon, tqdm=args.tqdm and rank == 0, latent_dimension=args.latent_dimension,
                 tunable_decoder_var=args.tunable_decoder_var, warm_start=args.warm_start, dataset_name=args.dataset,
                 latent_off_dimension=args.latent_off_dimension, force_generic=getattr(args, "force_generic", False),
                 dtype=getattr(args, "dtype", "f32"), world=world, rank=rank, global_batch=args.batch_size)
    eng = m.model.module.engine(hi - lo, args.batch_size)
    m.optimizer.exchange = GradExchange(eng, dist, mode=getattr(args, "comm", "auto"))
    if rank == 0:
        how = m.optimizer.exchange.mode + ("" if m.optimizer.exchange.in_library or eng.fused else ", per-layer buckets overlapped with the backward pass")
        print(f"Data parallel: world {world}, {hi - lo} rows per rank, gradient exchange {how}")
    m.key = vrandom.split(m.key, world)[rank]              # identical initial parameters, different latent draws
    m.rank = rank
    return m


def main(args):
    import os

    from .utils import get_output_dir, make_output_dir
    dist = None
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0")) if args.dist_backend == "nccl" else 0
        torch.cuda.set_device(local)
        args.device = local
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    if rank == 0:
        output_dir = make_output_dir(args.name, args.overwrite, args)
    if dist is not None:
        dist.barrier()
        output_dir = get_output_dir(args.name)
    dataset = get_dataset(args.dataset, args.dataset_seed, args.padding_dim, args.batch_size, args, world, rank)
    if dataset is None:
        raise ValueError("--dataset must be one of sphere, linear_gaussian, sigmoid")
    model = get_model(args, dataset, output_dir, dist)
    model.train()
    model.plot()
    model.save(final=True)
    if dist is not None:
        model.check_replicas()
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main(parse_arguments()))
