"""-m gpu: vaek_train_steps -- N software-pipelined train steps of a linear VAE evaluated through the batch's second-moment
matrix (csrc/linear_moments.hip) -- against the float64 oracle's N sequential VAE.train_step's (networks.py:87-101) on the
same batches, and against the library's own step-by-step path (vaek_train_step).  Tolerances are the f32 path's: loss 1e-5
relative (BASELINE's contract; measured ~1e-7), parameters within 2 % of one Adam step per step taken."""
import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.cases import build
from tests.gpu_util import dev, engine_for, host

pytestmark = pytest.mark.gpu


def _problem(cfg, dk, B, n, seed=5):
    _, sampler = O.make_dataset(**dk)
    rng = np.random.default_rng(seed)
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    p = {k: r32(v) for k, v in O.init_params(cfg, seed=3).items()}
    for k in p:
        if not k.endswith("kernel"):
            p[k] = r32(p[k] + 0.2 * rng.standard_normal(p[k].shape))
    batches = []
    for _ in range(n):
        x = r32(sampler(rng, B))
        z1, z2 = O.split_latents(r32(rng.standard_normal((B, cfg.L + cfg.D))), cfg.L)
        batches.append((x, np.ascontiguousarray(z1), np.ascontiguousarray(z2)))
    return p, batches


def _run_pipelined(eng, cfg, p, batches, lr):
    params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    ring = torch.zeros(len(batches) + 8, dtype=torch.float32, device="cuda")
    eng.set_loss_history(ring)
    eng.train_steps(params, grads, m, v, step, [tuple(dev(a) for a in b) for b in batches], lr)
    torch.cuda.synchronize()
    eng.set_loss_history(None)
    assert not eng.train_steps_gave_up(), "a bounded in-launch wait of the persistent form expired"
    return params, grads, m, v, int(step.item()), host(ring)[:len(batches)]


def _leafwise_grads_of_the_first_step(cfg, eng, p0, batch, lr):
    """grads[:P] of ONE vaek_train_steps step from p0, leaf by leaf against the oracle's loss_and_grad (networks.py:99): a small
    leaf (epsilon, epsilon_p, a bias) must be right on its OWN scale, not on the scale of the whole vector."""
    params, grads, m, v, step, losses = _run_pipelined(eng, cfg, p0, [batch], lr)
    loss, g = O.loss_and_grad(cfg, p0, *batch)
    got, want = host(grads), O.flatten(cfg, g)
    assert abs(got[eng.P] - loss) <= 1e-5 * abs(loss)
    assert np.max(np.abs(got[:eng.P] - want)) <= 2e-5 * np.max(np.abs(want)), "gradient, whole vector"
    worst = {}
    for name, (off, shape) in eng.leaves.items():
        k = int(np.prod(shape))
        worst[name] = float(np.max(np.abs(got[off:off + k] - want[off:off + k])) / (np.max(np.abs(want[off:off + k])) + 1e-30))
    assert max(worst.values()) <= 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:3]


@pytest.mark.parametrize("name,B,n", [("c1_linear_L20", 16, 6), ("c1_linear_L2", 16, 6), ("linear_notdv", 8, 5),
                                      ("c1_linear_L20", 1000, 7), ("c1_linear_L20", 256, 1), ("c1_linear_L2", 777, 2)])
def test_pipelined_steps_match_the_oracle(name, B, n):
    cfg, dk, _, lr = build(name)
    p, batches = _problem(cfg, dk, B, n)
    eng = engine_for(cfg, B)
    assert eng.supports_train_steps()
    _leafwise_grads_of_the_first_step(cfg, eng, p, batches[0], lr)
    params, grads, m, v, step, losses = _run_pipelined(eng, cfg, p, batches, lr)
    st = O.adam_init(p)
    for i, (x, z1, z2) in enumerate(batches):
        p, st, loss = O.train_step(cfg, p, st, x, z1, z2, lr)
        assert abs(losses[i] - loss) <= 1e-5 * abs(loss), (i, losses[i], loss)
    assert step == n
    assert abs(float(grads[eng.P]) - loss) <= 1e-5 * abs(loss)
    assert np.max(np.abs(host(params) - O.flatten(cfg, p))) <= 0.02 * lr * n
    wm, wv = O.flatten(cfg, st["m"]), O.flatten(cfg, st["v"])
    assert np.max(np.abs(host(m) - wm)) <= 2e-5 * np.max(np.abs(wm)) + 1e-9
    assert np.max(np.abs(host(v) - wv)) <= 5e-5 * np.max(np.abs(wv)) + 1e-12


def test_three_hundred_pipelined_steps_stay_on_the_oracle_trajectory():
    cfg, dk, _, lr = build("c1_linear_L20")
    B, n = 1024, 300
    p, batches = _problem(cfg, dk, B, n)
    eng = engine_for(cfg, B)
    params, grads, m, v, step, losses = _run_pipelined(eng, cfg, p, batches, lr)
    st = O.adam_init(p)
    want = []
    for x, z1, z2 in batches:
        p, st, loss = O.train_step(cfg, p, st, x, z1, z2, lr)
        want.append(loss)
    want = np.array(want)
    assert step == n and np.max(np.abs(losses - want) / np.abs(want)) <= 1e-5
    assert np.max(np.abs(host(params) - O.flatten(cfg, p))) <= 2e-5          # 300 steps of 1e-3: within 2 % of ONE step
    assert want[-1] < want[0]


def test_metric_size_and_agreement_with_the_step_by_step_kernels():
    """B = 65 536 (the metric's batch): 4 pipelined steps vs the oracle, and vs vaek_train_step on the same batches (the two
    library paths differ only in summation order)."""
    cfg, dk, _, lr = build("c1_linear_L20")
    B, n = 65536, 4
    p0, batches = _problem(cfg, dk, B, n)
    eng = engine_for(cfg, B)
    _leafwise_grads_of_the_first_step(cfg, eng, p0, batches[0], lr)
    params, grads, m, v, step, losses = _run_pipelined(eng, cfg, p0, batches, lr)
    p, st = dict(p0), O.adam_init(p0)
    for i, (x, z1, z2) in enumerate(batches):
        p, st, loss = O.train_step(cfg, p, st, x, z1, z2, lr)
        assert abs(losses[i] - loss) <= 1e-5 * abs(loss), (i, losses[i], loss)
    assert np.max(np.abs(host(params) - O.flatten(cfg, p))) <= 0.02 * lr * n
    p2 = dev(O.flatten(cfg, p0)); g2 = eng.new_flat(eng.grad_len); m2 = eng.new_flat(); v2 = eng.new_flat()
    s2 = torch.zeros(1, dtype=torch.int32, device="cuda")
    for x, z1, z2 in batches:
        eng.train_step(p2, g2, m2, v2, s2, dev(x), dev(z1), dev(z2), lr)
    assert float((params - p2).abs().max()) <= 0.02 * lr and abs(float(grads[eng.P]) - float(g2[eng.P])) <= 1e-5 * abs(float(g2[eng.P]))


def test_pipelined_steps_refuse_what_they_cannot_do():
    from vae_training_amd._lib import VaekError
    cfg = O.Config(7, 6, (), (), -3.0, True, "sigmoid")           # two decoders with a sigmoid head: not linear in the inputs
    eng = engine_for(cfg, 64)
    assert not eng.supports_train_steps()
    z = lambda *s: torch.zeros(*s, device="cuda")
    with pytest.raises(VaekError):
        eng.train_steps(eng.new_flat(), eng.new_flat(eng.grad_len), eng.new_flat(), eng.new_flat(),
                        torch.zeros(1, dtype=torch.int32, device="cuda"), [(z(64, 7), z(64, 6), z(64, 7))], 1e-3)
    mlp = engine_for(O.Config(12, 20, (32,), (32,), -1.0, True, "linear_gaussian"), 64)
    assert not mlp.supports_train_steps()


def test_persistent_and_launch_per_step_forms_agree(monkeypatch):
    """The two schedules of vaek_train_steps (one persistent launch per 64 steps with in-launch hand-offs; n + 2 launches ordered
    by the stream) are the same train steps in two summation orders -- the persistent form sums a tile over 4 multiplying waves
    and updates on the float64 matrix cores, the launch-per-step form sums over 8 waves and updates on the vector units: after 70
    steps every loss within 2e-6 relative, parameters within 2e-6, Adam moments within 1e-5 of their scale.  A subprocess takes
    the launch-per-step form (the choice is read once per process)."""
    import os, subprocess, sys, json
    code = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from oracle import elbo_oracle as O
from tests.cases import build
from tests.gpu_util import dev, engine_for
from tests.test_gpu_steps import _problem, _run_pipelined
cfg, dk, _, lr = build("c1_linear_L20")
p, batches = _problem(cfg, dk, 3000, 70)
eng = engine_for(cfg, 3000)
params, grads, m, v, step, losses = _run_pipelined(eng, cfg, p, batches, lr)
print(json.dumps({"p": params.cpu().numpy().astype(np.float64).tolist(), "m": m.cpu().numpy().astype(np.float64).tolist(), "l": np.asarray(losses, np.float64).tolist()}))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for persist in ("1", "0"):
        env = dict(os.environ, VAEK_LIN_PERSIST=persist)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append({k: np.asarray(v) for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items()})
    a, b = outs
    assert np.max(np.abs(a["l"] - b["l"]) / np.abs(b["l"])) <= 2e-6
    assert np.max(np.abs(a["p"] - b["p"])) <= 2e-6
    assert np.max(np.abs(a["m"] - b["m"])) <= 1e-5 * np.max(np.abs(b["m"]))


def test_graph_replay_of_pipelined_steps_equals_the_eager_call():
    """vaek_train_steps captured into a hipGraph (bench.py's steady-state loop) and replayed: bitwise the parameters, moments and
    losses of the same call made eagerly -- the zeroing of the arrival counters and every hand-off of the persistent launch must
    survive capture -- and no bounded wait expires."""
    cfg, dk, _, lr = build("c1_linear_L20")
    B, n = 70000, 130                                  # 274 tiles; two full persistent launches and a short one per call
    p, batches = _problem(cfg, dk, B, 6)
    dbat = [tuple(dev(a) for a in b) for b in batches]
    seq = [dbat[i % len(dbat)] for i in range(n)]
    eng = engine_for(cfg, B)

    def fresh():
        return (dev(O.flatten(cfg, p)), eng.new_flat(eng.grad_len), eng.new_flat(), eng.new_flat(),
                torch.zeros(1, dtype=torch.int32, device="cuda"))

    pe, ge, me, ve, se = fresh()
    ring_e = torch.zeros(2 * n + 8, dtype=torch.float32, device="cuda")
    eng.set_loss_history(ring_e)
    eng.train_steps(pe, ge, me, ve, se, seq, lr)
    eng.train_steps(pe, ge, me, ve, se, seq, lr)
    torch.cuda.synchronize()
    assert not eng.train_steps_gave_up(), hex(eng.train_steps_status_word)

    pg, gg, mg, vg, sg = fresh()
    ring_g = torch.zeros(2 * n + 8, dtype=torch.float32, device="cuda")
    eng.set_loss_history(ring_g)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            eng.train_steps(pg, gg, mg, vg, sg, seq, lr)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    eng.set_loss_history(None)
    assert not eng.train_steps_gave_up(), hex(eng.train_steps_status_word)
    assert int(sg.item()) == int(se.item()) == 2 * n
    assert torch.equal(pg, pe) and torch.equal(mg, me) and torch.equal(vg, ve) and torch.equal(gg, ge)
    assert torch.equal(ring_g, ring_e) and float(ring_e[2 * n - 1]) < float(ring_e[0])


@pytest.mark.parametrize("seed", range(14))
def test_pipelined_steps_random_shapes(seed):
    """Random (D, L, B, with / without the tunable decoder variance): odd row lengths (16-byte LDS-DMA pieces straddling rows and
    tensor ends), fewer rows than a tile, three- and four-block feature counts (persistent and launch-per-step forms), batches
    that need taller tiles -- three steps against the oracle each."""
    rng = np.random.default_rng(1000 + seed)
    while True:
        D, L = int(rng.integers(1, 22)), int(rng.integers(1, 22))
        if L + 2 * D + 1 <= 63:
            break
    B = int(rng.choice([1, 2, 7, 31, 255, 256, 257, 1000, 4097, 70001])) if seed < 10 else int(rng.integers(1, 3000))
    if B * min(D, L) < 8:
        B = 8
    tdv = bool(rng.integers(0, 2))
    cfg = O.Config(D, L, (), (), float(rng.uniform(-3.0, 0.5)), tdv, "linear_gaussian")
    r32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    p = {k: r32(v) for k, v in O.init_params(cfg, seed=seed).items()}
    for k in p:
        if not k.endswith("kernel"):
            p[k] = r32(p[k] + 0.2 * rng.standard_normal(p[k].shape))
    n, lr = 3, 1e-3
    A = rng.standard_normal((D, D)) / np.sqrt(D)
    batches = []
    for _ in range(n):
        x = r32(rng.standard_normal((B, D)) @ A)
        batches.append((x, r32(rng.standard_normal((B, L))), r32(rng.standard_normal((B, D)))))
    eng = engine_for(cfg, B)
    assert eng.supports_train_steps()
    params, grads, m, v, step, losses = _run_pipelined(eng, cfg, p, batches, lr)
    st = O.adam_init(p)
    for i, (x, z1, z2) in enumerate(batches):
        p, st, loss = O.train_step(cfg, p, st, x, z1, z2, lr)
        assert abs(losses[i] - loss) <= 1e-5 * abs(loss) + 1e-6, (D, L, B, tdv, i, losses[i], loss)
    assert step == n
    assert np.max(np.abs(host(params) - O.flatten(cfg, p))) <= 0.02 * lr * n, (D, L, B, tdv)
