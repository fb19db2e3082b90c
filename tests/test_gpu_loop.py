"""-m gpu: the reference's hot loop (model.py:207-222 -> vae.py:123-130 -> networks.py:87-101: get_batch, sample_latent,
train_step, N times) on the headline kernel -- vaek_train_steps_gen, where every step's batch is DRAWN inside the persistent
launch by the workgroups that multiply it (csrc/linear_moments.hip, GEN form; csrc/rng_dev.h) -- and its callers:
trainer.GraphLoop(moments=True), `run.py --fast_loop`.  Parity unpinned as everywhere (no JAX here): the checks are (i) against
vaek_make_batch + vaek_train_steps on the SAME Philox streams, bit for bit; (ii) against the per-sample loop (vaek_train_step_gen),
to summation order; (iii) against the float64 oracle on batches copied back from the device generator."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.gpu_util import dev, host

pytestmark = pytest.mark.gpu


def _spec(kind, dd, did, device="cuda"):
    g = torch.Generator().manual_seed(11)
    if kind == 0:
        return torch.randn(dd, did, generator=g).to(device).contiguous()
    return None


@pytest.mark.parametrize("kind,D,L,dd,pad,B,n,var", [
    (0, 12, 20, 3, 9, 1000, 5, 0.0),        # ragged batch, run-time shapes (T = 256: four tiles on four streamers)
    (0, 12, 20, 3, 9, 65536, 3, 0.0),       # the metric's shape and kernel instantiation (228 tiles of 288 rows, the last one ragged)
    (0, 12, 20, 3, 9, 3000, 70, 0.25),      # two persistent launches (64 + 6 steps), dataset noise (-dn)
    (2, 6, 6, 3, 3, 300, 4, 0.0),           # sphere dataset, L and D not multiples of 4 x 4 columns
    (0, 7, 10, 3, 4, 130, 3, 0.0),          # odd D: scalar LDS writes of the x rows
])
def test_in_launch_draw_equals_make_batch_plus_train_steps(kind, D, L, dd, pad, B, n, var):
    """vaek_train_steps_gen(n) == vaek_make_batch(step = t) for t = 0 .. n-1 followed by vaek_train_steps on those n batches:
    the same tiles reach the same matrix-core products, so parameters, moments, gradients and the loss ring are BITWISE equal."""
    from vae_training_amd.engine import Engine
    eng = Engine(B, D, L, (), (), -1.0, True, False)
    did = dd
    assert eng.supports_train_steps() and eng.supports_train_steps_gen(kind)
    A = _spec(kind, dd, did)
    torch.manual_seed(0)
    p0 = (torch.randn(eng.P, device="cuda") * 0.3).contiguous()
    seed, tag, row0 = 77, 5, 1000

    def state():
        return [p0.clone(), eng.new_flat(eng.grad_len), eng.new_flat(), eng.new_flat(), torch.zeros(1, dtype=torch.int32, device="cuda")]
    a, b = state(), state()
    ring_a = torch.zeros(n + 4, dtype=torch.float32, device="cuda"); ring_b = torch.zeros_like(ring_a)
    eng.set_loss_history(ring_a)
    eng.train_steps_gen(*a, n, 1e-3, kind, A, dd, did, pad, var, seed, tag=tag, row0=row0)
    torch.cuda.synchronize()
    assert not eng.train_steps_gave_up()
    batches = [eng.make_batch(kind, A, dd, did, pad, var, B, seed, step=t, tag=tag, row0=row0) for t in range(n)]
    eng.set_loss_history(ring_b)
    eng.train_steps(*b, batches, 1e-3)
    torch.cuda.synchronize()
    eng.set_loss_history(None)
    assert not eng.train_steps_gave_up()
    assert int(a[4].item()) == n == int(b[4].item())
    for x, y, what in zip(a[:4], b[:4], ("params", "grads", "m", "v")):
        assert torch.equal(x, y), what
    assert torch.equal(ring_a[:n], ring_b[:n]) and bool(torch.isfinite(ring_a[:n]).all())


def test_in_launch_draw_follows_the_oracle_on_the_generated_batches():
    """Three steps of vaek_train_steps_gen against three float64 oracle steps (networks.py:87-101) on the batches the device
    generator draws for those RNG steps (copied back): loss 1e-5 relative, parameters within 2 % of an Adam step per step."""
    from vae_training_amd.engine import Engine
    cfg = O.Config(12, 20, (), (), -1.0, True, "linear_gaussian")
    B, n, lr = 4096, 3, 1e-3
    eng = Engine(B, 12, 20, (), (), -1.0, True, False)
    A = _spec(0, 3, 3)
    r32 = lambda t: np.asarray(t, np.float32).astype(np.float64)
    p = {k: r32(v) for k, v in O.init_params(cfg, seed=3).items()}
    params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    ring = torch.zeros(8, dtype=torch.float32, device="cuda")
    eng.set_loss_history(ring)
    eng.train_steps_gen(params, grads, m, v, step, n, lr, 0, A, 3, 3, 9, 0.0, 123)
    torch.cuda.synchronize()
    eng.set_loss_history(None)
    assert not eng.train_steps_gave_up()
    st = O.adam_init(p)
    for t in range(n):
        x, z1, z2 = (r32(host(a)) for a in eng.make_batch(0, A, 3, 3, 9, 0.0, B, 123, step=t))
        p, st, loss = O.train_step(cfg, p, st, x, z1, z2, lr)
        assert abs(float(ring[t]) - loss) <= 1e-5 * abs(loss), (t, float(ring[t]), loss)
    assert np.max(np.abs(host(params) - O.flatten(cfg, p))) <= 0.02 * lr * n


def _cli_model(tmp_path, name, B, fast_loop):
    from vae_training_amd.run import get_dataset, parse_arguments
    from vae_training_amd.vae import VAEModel
    args = parse_arguments([name, "--dataset", "linear_gaussian", "--padding_dim", "9", "-dd", "3", "-ds", "2"])
    ds = get_dataset("linear_gaussian", 2, 9, B, args)
    return VAEModel(dirname=str(tmp_path), num_batches=10, num_epochs=1, batch_size=B, learning_rate=1e-3, layer_sizes="",
                    encoder_layer_sizes="", state_dict=None, data_fn=None, epsilon=-1.0, tqdm=False, dataset=ds,
                    latent_dimension=20, tunable_decoder_var=True, dataset_name="linear_gaussian", fast_loop=fast_loop)


@pytest.mark.parametrize("B,n", [(100, 150), (65536, 70)])
def test_graph_loop_on_the_moment_kernel_matches_the_per_sample_loop(tmp_path, B, n):
    """trainer.GraphLoop(moments=True) -- what run.py --fast_loop runs for a linear VAE -- against GraphLoop(moments=False)
    (vaek_train_step_gen: the per-sample kernels on the same Philox batches): every loss within 1e-5 relative, parameters within
    2 % of an Adam step per step taken."""
    from vae_training_amd.trainer import GraphLoop
    a, b = _cli_model(tmp_path, "a", B, True), _cli_model(tmp_path, "b", B, True)
    la, lb = GraphLoop(a, seed=9), GraphLoop(b, seed=9, steps_per_graph=16, moments=False)
    assert la.moments and not lb.moments and la.eng.supports_train_steps() and not la.bufs
    la.run(n // 2); la.run(n - n // 2)
    lb.run(n)
    la.check()
    assert a.optimizer.state.step == n == int(a.optimizer.state.step_dev.item()) == int(b.optimizer.state.step_dev.item())
    xa, xb = la.losses().double(), lb.losses().double()
    assert xa.numel() == n and float(((xa - xb).abs() / xb.abs()).max()) <= 1e-5
    assert float((a.model.flat - b.model.flat).abs().max()) <= 0.02 * 1e-3 * n
    assert float(xa[-1]) < float(xa[0])


def test_run_py_fast_loop_at_the_metric_batch_size(tmp_path, monkeypatch, capsys):
    """`python run.py NAME ... --batch_size 65536 --fast_loop`: the reference's CLI on the headline kernel -- same side effects
    and printed stats (model.py:195-205, 246-255), losses.npz holding one loss per step, and the same losses as the per-sample
    fast loop started from the same seeds."""
    from vae_training_amd import run, utils
    from vae_training_amd.trainer import GraphLoop
    monkeypatch.setattr(utils, "DATA_DIR", str(tmp_path) + "/")
    made = []
    orig = GraphLoop.__init__

    def spy(self, *a, **kw):
        orig(self, *a, **kw)
        made.append(self)
    monkeypatch.setattr(GraphLoop, "__init__", spy)
    base = ["--dataset", "linear_gaussian", "--encoder_layer_sizes", "", "--layer_sizes", "", "-ow", "--latent_dim", "20", "--padding_dim", "9",
            "-dd", "3", "--num_batches", "40", "--batch_size", "65536", "--epsilon", "-1", "-tdv", "-ds", "2", "-lr", "1e-3"]
    assert run.main(run.parse_arguments(["fast"] + base + ["--fast_loop"])) == 0
    out = capsys.readouterr().out
    assert "Batch | 0 | VAE Loss" in out and "Squared Norm of padding dimensions" in out
    assert len(made) == 1 and made[0].moments and made[0].eng.supports_train_steps()
    z = np.load(os.path.join(str(tmp_path), "fast", "losses.npz"), allow_pickle=True)
    fast = np.asarray(z["VAE Loss"], dtype=np.float64)
    assert {"args.json", "losses.npz", "model.pkl"} <= set(os.listdir(os.path.join(str(tmp_path), "fast")))
    # the same run with the fast loop on the per-sample kernels (same seeds -> same Philox batches)
    monkeypatch.setattr(GraphLoop, "__init__", lambda self, *a, **kw: (orig(self, *a, **dict(kw, moments=False)), made.append(self))[0])
    assert run.main(run.parse_arguments(["slow"] + base + ["--fast_loop"])) == 0
    capsys.readouterr()
    assert len(made) == 2 and not made[1].moments
    z2 = np.load(os.path.join(str(tmp_path), "slow", "losses.npz"), allow_pickle=True)
    slow = np.asarray(z2["VAE Loss"], dtype=np.float64)
    assert fast.shape == slow.shape and fast.size >= 40 and np.isfinite(fast).all()
    assert np.max(np.abs(fast - slow) / np.abs(slow)) <= 1e-5
    assert fast[-1] < fast[1]
    # default (neither --fast_loop nor --no_fast_loop): a linear VAE takes the moment loop by itself
    monkeypatch.setattr(GraphLoop, "__init__", spy)
    assert run.main(run.parse_arguments(["auto"] + base)) == 0
    assert len(made) == 3 and made[2].moments


def test_sticky_status_survives_later_launches():
    """a bounded wait that expires in an early launch must still be reported after later launches.  A subprocess
    runs 130 steps (three persistent launches) WITHOUT streamers (VAEK_LIN_ROLES=6): the first launch's reducers give up."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, json, torch
sys.path.insert(0, %r)
from vae_training_amd.engine import Engine
eng = Engine(3000, 12, 20, (), (), -1.0, True, False)
z = lambda *s: torch.zeros(*s, device="cuda")
st = [torch.randn(eng.P, device="cuda") * 0.1, eng.new_flat(eng.grad_len), eng.new_flat(), eng.new_flat(), torch.zeros(1, dtype=torch.int32, device="cuda")]
b = (z(3000, 12), z(3000, 20), z(3000, 12))
eng.train_steps(*st, [b] * 130, 1e-3)
torch.cuda.synchronize()
first = eng.train_steps_gave_up(); w = eng.train_steps_status_word
second = eng.train_steps_gave_up()
print(json.dumps({"first": first, "word": w, "second": second}))
""" % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VAEK_LIN_ROLES="6"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["first"] and got["word"] & 0x80000000 and not got["second"]        # reported after the LAST launch; read-and-clear
