"""-m gpu: the whole-network kernel for one-hidden-layer MLP VAEs (csrc/fused_mlp1.hip: forward, ELBO and backward of
networks.py:61-99 in one launch + the fused finalize) against the float64 oracle, gradient leaf by gradient leaf, and against
the layer-by-layer kernels (force_generic) on the same inputs.  BASELINE config 2's shape at full width and batch, ragged
batches, one and two decoders, unequal hidden widths, the 16-wide feature variant."""
import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.gpu_util import dev, engine_for, host, random_problem, rel_err

pytestmark = pytest.mark.gpu

SIG = dict(name="sigmoid", seed=69, dd=3, pad=3)            # D = 7 (dd + pad + 1)
SPH = dict(name="sphere", seed=69, dd=3, pad=3)             # D = 6
LIN = dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=9)   # D = 12


def _grads(eng, cfg, p, x, z1, z2):
    grads = eng.new_flat(eng.grad_len)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.profile_begin(64)
    eng.grads_only(dev(O.flatten(cfg, p)), grads, step, dev(x), dev(z1), dev(z2))
    torch.cuda.synchronize()
    return host(grads), eng.profile_report()


@pytest.mark.parametrize("cfg,dk,B", [
    (O.Config(7, 6, (256,), (256,), -3.0, True, "sigmoid"), SIG, 8192),        # BASELINE config 2 as bench.py runs it
    (O.Config(7, 6, (256,), (256,), -3.0, True, "sigmoid"), SIG, 1000),        # ragged last tile (1000 = 31 x 32 + 8)
    (O.Config(7, 6, (256,), (256,), -3.0, True, "sigmoid"), SIG, 5),           # less than one tile
    (O.Config(6, 6, (200,), (200,), -3.0, True, "sphere"), SPH, 4096),         # one decoder, 200 units (the scripts' width)
    (O.Config(7, 6, (48,), (33,), -2.0, False, "sigmoid"), SIG, 777),          # unequal widths, fixed decoder variance
    (O.Config(12, 10, (64,), (128,), -1.0, True, "linear_gaussian"), LIN, 2048),   # D, L > 8: the 16-wide variant
    (O.Config(7, 6, (256,), (256,), -3.0, True, "sigmoid"), SIG, 40000),       # more tiles than workgroups: several per workgroup
])
def test_mlp1_every_gradient_leaf_matches_the_oracle(cfg, dk, B):
    p, x, z1, z2 = random_problem(cfg, dk, B)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    eng = engine_for(cfg, B)
    assert eng.fused
    got, rep = _grads(eng, cfg, p, x, z1, z2)
    assert "fused_mlp1_fwd_bwd" in rep and not any(k.startswith("gemm") for k in rep), sorted(rep)
    assert abs(got[eng.P] - loss) <= 1e-5 * abs(loss), (got[eng.P], loss)
    want = O.flatten(cfg, g)
    assert rel_err(got[:eng.P], want) <= 2e-5
    for name, (off, shape) in eng.leaves.items():
        n = int(np.prod(shape))
        assert np.max(np.abs(got[off:off + n] - want[off:off + n])) <= 1e-4 * (np.max(np.abs(want[off:off + n])) + 1e-30), name
    # and the layer-by-layer kernels on the same inputs
    gen, rep2 = _grads(engine_for(cfg, B, force_generic=True), cfg, p, x, z1, z2)
    assert "fused_mlp1_fwd_bwd" not in rep2
    assert rel_err(got[:eng.P], gen[:eng.P]) <= 2e-5 and abs(got[eng.P] - gen[eng.P]) <= 2e-6 * abs(loss)


def test_mlp1_train_steps_follow_the_oracle_and_are_repeatable():
    cfg = O.Config(7, 6, (256,), (256,), -3.0, True, "sigmoid")
    B, lr = 8192, 1e-4
    p, x, z1, z2 = random_problem(cfg, SIG, B)
    eng = engine_for(cfg, B)
    runs = []
    for _ in range(2):
        params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        q, st = dict(p), O.adam_init(p)
        for k in range(3):
            q, st, loss = O.train_step(cfg, q, st, x, z1, z2, lr)
            eng.train_step(params, grads, m, v, step, dev(x), dev(z1), dev(z2), lr)
            assert abs(float(grads[eng.P]) - loss) <= 1e-5 * abs(loss), (k, float(grads[eng.P]), loss)
        assert np.max(np.abs(host(params) - O.flatten(cfg, q))) <= 0.02 * lr * 3
        wm = O.flatten(cfg, st["m"])
        assert np.max(np.abs(host(m) - wm)) <= 5e-5 * np.max(np.abs(wm))
        runs.append((params.clone(), grads.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])       # no atomics anywhere: bitwise


def test_mlp1_shards_sum_to_the_full_batch():
    """data parallel: two half-batch contexts dividing by the GLOBAL batch sum to the full-batch gradient."""
    cfg = O.Config(7, 6, (256,), (256,), -3.0, True, "sigmoid")
    B = 2048
    p, x, z1, z2 = random_problem(cfg, SIG, B)
    full, _ = _grads(engine_for(cfg, B), cfg, p, x, z1, z2)
    parts = []
    for r in range(2):
        sl = slice(r * B // 2, (r + 1) * B // 2)
        e = engine_for(cfg, B // 2, world=2, rank=r, global_batch=B)
        assert e.fused
        parts.append(_grads(e, cfg, p, x[sl], z1[sl], z2[sl])[0])
    tot = parts[0] + parts[1]
    assert rel_err(tot[:e.P], full[:e.P]) <= 2e-6
