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
    p.add_argument("--fast_loop", action="store_true",
                   help="run the steps between stats/plots from a hipGraph with on-device Philox batches (trainer.py)")
    return p


def parse_arguments(argv=None):
    args = build_parser().parse_args(argv)
    args.model = "VAE"
    args.latent_distribution = "gaussian"
    args.tqdm = True
    return args


def get_dataset(name, seed, padding_dimension, batch_size, args):
    from .datasets import LinearGaussianDataset, SigmoidDataset, SphereDataset
    dev = None if getattr(args, "device", None) is None else f"cuda:{args.device}"
    if name == "sphere":
        return SphereDataset(seed, dimension=args.dataset_dimension, padding_dimension=args.padding_dim, device=dev)
    if name == "linear_gaussian":
        return LinearGaussianDataset(seed, dimension=args.dataset_dimension,
                                     intrinsic_dimension=args.dataset_intrinsic_dimension,
                                     padding_dimension=args.padding_dim, var_added=args.dataset_noise, device=dev)
    if name == "sigmoid":
        return SigmoidDataset(seed, dimension=args.dataset_dimension, padding_dimension=args.padding_dim, device=dev)
    return None


def get_model(args, dataset, output_dir):
    from .vae import VAEModel
    if args.model != "VAE":
        raise NameError(f"model {args.model!r}: only the VAE branch exists (run.py:250-268)")
    return VAEModel(dirname=output_dir, batch_size=args.batch_size, learning_rate=args.learning_rate, dataset=dataset,
                    num_batches=args.num_batches, num_epochs=args.num_epochs, layer_sizes=args.layer_sizes,
                    encoder_layer_sizes=args.encoder_layer_sizes, state_dict=args.state_dict, data_fn=args.data_fn,
                    epsilon=args.epsilon, tqdm=args.tqdm, latent_dimension=args.latent_dimension,
                    tunable_decoder_var=args.tunable_decoder_var, warm_start=args.warm_start,
                    dataset_name=args.dataset, latent_off_dimension=args.latent_off_dimension,
                    force_generic=getattr(args, "force_generic", False), fast_loop=getattr(args, "fast_loop", False))


def main(args):
    from .utils import make_output_dir
    output_dir = make_output_dir(args.name, args.overwrite, args)
    dataset = get_dataset(args.dataset, args.dataset_seed, args.padding_dim, args.batch_size, args)
    if dataset is None:
        raise ValueError("--dataset must be one of sphere, linear_gaussian, sigmoid")
    model = get_model(args, dataset, output_dir)
    model.train()
    model.plot()
    model.save(final=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main(parse_arguments()))
