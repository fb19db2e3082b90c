"""The two utils.py helpers on the VAE path: relu (utils.py:29-30; the kernels apply it, this is
for host-side checks) and the output-directory helper behind `run.py NAME -ow` (utils.py:46-65)."""
from __future__ import annotations

import json
import os

import torch

DATA_DIR = "data/"


def relu(x):
    return torch.clamp_min(x, 0)


def get_output_dir(name):
    return os.path.join(DATA_DIR, name)


def make_output_dir(name, overwrite, args):
    """data/NAME/ with args.json inside; -ow empties an existing directory, otherwise an existing
    directory is an error (same behaviour as the reference)."""
    dirname = get_output_dir(name)
    if os.path.exists(dirname):
        if not overwrite:
            raise ValueError(f"{dirname} already exists! Use a different name")
        for fn in os.listdir(dirname):
            os.remove(os.path.join(dirname, fn))
    else:
        os.makedirs(dirname)
    with open(os.path.join(dirname, "args.json"), "w") as f:
        json.dump(vars(args) if not isinstance(args, dict) else args, f)
    return dirname
