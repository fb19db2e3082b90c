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


def test_run_py_two_ranks_hidden_layers_rccl_overlapped(tmp_path):
    """A layer-by-layer model forced onto the all-reduce transport (gloo stands in for RCCL on a one-GPU box): the CLI's
    train step takes vaek_train_step_grads_bucketed + one all-reduce per layer bucket + vaek_train_step_apply, and the
    replicas end bitwise identical (run.py checks and says so)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29518", os.path.join(ROOT, "run.py"), "dp2h", "--dataset", "sphere", "--encoder_layer_sizes", "48|32",
           "--layer_sizes", "32|48", "-ow", "--latent_dim", "6", "--padding_dim", "3", "-dd", "3", "--num_batches", "30",
           "--batch_size", "512", "--epsilon", "-3", "-tdv", "-lr", "1e-3", "--dist_backend", "gloo", "--comm", "rccl"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "gradient exchange rccl, per-layer buckets overlapped" in r.stdout, r.stdout[-1500:]
    assert "replicas identical on 2 ranks after 30 steps" in r.stdout, r.stdout[-1500:]
    z = np.load(os.path.join(str(tmp_path), "data", "dp2h", "losses.npz"), allow_pickle=True)
    losses = np.asarray(z["VAE Loss"], dtype=np.float64)
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[1:6])


def test_rank0_slow_plot_does_not_break_the_p2p_exchange(tmp_path):
    """Rank 0 spends 4.5 s in its plot/save block (longer than the in-kernel exchange's spin bound): the other rank must
    wait at the host barrier, not spin into a give-up -- no time-out recorded, replicas bitwise identical."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", os.path.join(ROOT, "tests", "dp_slow_rank0_worker.py"), str(tmp_path / "slow")]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for rank in (0, 1):
        assert f"RESULT rank={rank} replicas_identical=True timed_out=False steps=30" in r.stdout, r.stdout[-1500:]


@pytest.mark.parametrize("comm,port", [("p2p", 29521), ("rccl", 29522)])
def test_bench_py_two_ranks_rehearsed_on_one_gpu(comm, port):
    """The driver's multi-GPU bench command line, at two ranks sharing this box's one GPU (--rehearse-one-gpu: both ranks on
    cuda:0, gloo standing in for RCCL's rendezvous): rank 0 prints exactly one JSON line for n_gpus 2 with the whole-job rate,
    the loss is finite, and the in-kernel P2P exchange records no time-out (bench.py asserts both before it prints)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--steps", "20",
           "--warmup", "5", "--comm", comm, "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["unit"] == "samples/s" and out["config"]["parallelism"] == "dp2", out
    assert out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"]
    assert np.isfinite(out["value"]) and out["value"] > 0 and np.isfinite(out["config"]["final_loss"]), out
    assert out["config"]["grad_exchange"] == comm, out["config"]
    # either transport runs the step through the batch's second-moment matrix (vaek_train_steps): P2P exchanges it inside the
    # persistent launch, RCCL all-reduces it between the two halves of a launch-per-step step
    assert out["config"]["step_entry_point"].startswith("vaek_train_steps")
    assert (comm == "p2p") == ("persistent" in out["config"]["step_entry_point"])
    assert (comm == "rccl") == ("all-reduce of the 12 KB moment matrix" in out["config"]["step_entry_point"])


def test_conv_vae_two_ranks_equal_one_rank_on_the_whole_batch(tmp_path):
    """The convolutional VAE data parallel (conv_vae.py: world > 1): the all-reduced shard gradients are the whole batch's gradient,
    the replicas stay bitwise identical over 4 steps (tests/dp_conv_worker.py)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29523", os.path.join(ROOT, "tests", "dp_conv_worker.py")]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("RESULT"):
            kv = dict(t.split("=") for t in ln.split()[1:])
            res[int(kv["rank"])] = kv
    assert set(res) == {0, 1}, r.stdout[-1500:]
    assert all(kv["replicas_identical"] == "True" and kv["steps"] == "4" for kv in res.values()), res
    assert float(res[0]["worst_rel"]) <= 1e-4, res


def test_bench_py_conv_two_ranks_rehearsed_on_one_gpu():
    """bench.py --workload C5 at two ranks on this box's one GPU (gloo standing in for RCCL): one JSON line for n_gpus 2 with the
    whole-job rate; bench.py itself asserts that the replicas ended identical."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29524", os.path.join(ROOT, "bench.py"), "--workload", "C5", "--gpus", "2", "--rehearse-one-gpu", "--batch", "64",
           "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 128 and out["scaling"] == "weak"
    assert np.isfinite(out["value"]) and out["value"] > 0 and np.isfinite(out["config"]["final_loss"]), out
