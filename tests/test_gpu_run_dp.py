"""-m gpu: `run.py` under torch.distributed.run -- the CLI's data-parallel mode (two ranks rehearsed on one
GPU over gloo): global batch sharded, gradients exchanged, replicas identical, rank 0 writes the outputs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_py_two_ranks(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "run.py"), "dp2", "--dataset", "linear_gaussian", "--encoder_layer_sizes", "",
           "--layer_sizes", "", "-ow", "--latent_dim", "20", "--padding_dim", "9", "-dd", "3", "--num_batches", "40", "--batch_size", "512",
           "--epsilon", "-1", "-tdv", "-ds", "2", "-lr", "1e-3", "--dist_backend", "gloo"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    os.makedirs(os.path.join(str(tmp_path), "data"), exist_ok=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = os.path.join(str(tmp_path), "data", "dp2")
    assert {"args.json", "losses.npz", "model.pkl"} <= set(os.listdir(d))
    assert r.stdout.count("Batch | 0 | VAE Loss") == 1            # only rank 0 reports
    z = np.load(os.path.join(d, "losses.npz"), allow_pickle=True)
    losses = np.asarray(z["VAE Loss"], dtype=np.float64)
    assert len(losses) == 41 and np.isfinite(losses).all() and losses[-1] < losses[1]
