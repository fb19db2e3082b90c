"""-m gpu: the fused forward / ELBO / backward kernel of a WIDE linear decoder (csrc/linear_wide.hip; BASELINE config 4: linear-padding
at ambient dimension 4096, networks.py:61-84 and :94-99 for one Dense L -> D) against the float64 oracle, leaf by leaf, and against
the layer-by-layer kernels it replaces (force_generic).  Parity unpinned as everywhere (no JAX here)."""
import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.gpu_util import dev, engine_for, host, random_problem

pytestmark = pytest.mark.gpu


def _grads(eng, cfg, p, x, z1, z2):
    grads = eng.new_flat(eng.grad_len)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.profile_begin(256)
    eng.grads_only(dev(O.flatten(cfg, p)), grads, step, dev(x), dev(z1), dev(z2))
    torch.cuda.synchronize()
    return grads, eng.profile_report()


@pytest.mark.parametrize("D,L,B,tdv,hidden", [(1024, 20, 1000, True, ()), (2048, 7, 300, False, ()), (4096, 20, 4096, True, ()),
                                              (1024, 31, 513, True, ()), (1024, 6, 640, True, (48,))])
def test_every_gradient_leaf_against_the_oracle(D, L, B, tdv, hidden):
    cfg = O.Config(D, L, hidden, (), -1.0, tdv, "linear_gaussian")
    p, x, z1, z2 = random_problem(cfg, dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=D - 3), B)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    eng = engine_for(cfg, B)
    grads, rep = _grads(eng, cfg, p, x, z1, z2)
    assert "lwd_decoder_fwd_bwd" in rep and not any(k.startswith("gemm_f32_fwd_elbo") for k in rep), sorted(rep)
    got, want = host(grads), O.flatten(cfg, g)
    assert abs(got[eng.P] - loss) <= 1e-5 * abs(loss), (got[eng.P], loss)
    assert np.max(np.abs(got[:eng.P] - want)) <= 2e-5 * np.max(np.abs(want))
    worst = {}
    for name, (off, shape) in eng.leaves.items():
        k = int(np.prod(shape))
        worst[name] = float(np.max(np.abs(got[off:off + k] - want[off:off + k])) / (np.max(np.abs(want[off:off + k])) + 1e-30))
    assert max(worst.values()) <= 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    # bitwise repeatable; and the same numbers (to summation order) as the layer-by-layer kernels it replaces
    again, _ = _grads(eng, cfg, p, x, z1, z2)
    assert torch.equal(grads, again)
    ref, rep2 = _grads(engine_for(cfg, B, force_generic=True), cfg, p, x, z1, z2)
    assert "lwd_decoder_fwd_bwd" not in rep2
    assert float((grads - ref).abs().max()) <= 2e-5 * float(ref[:eng.P].abs().max())


def test_three_train_steps_and_shard_additivity():
    cfg = O.Config(1024, 20, (), (), -1.0, True, "linear_gaussian")
    B, lr = 2048, 1e-3
    p, x, z1, z2 = random_problem(cfg, dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=1021), B)
    eng = engine_for(cfg, B)
    params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    q, st = dict(p), O.adam_init(p)
    for _ in range(3):
        eng.train_step(params, grads, m, v, step, dev(x), dev(z1), dev(z2), lr)
        q, st, loss = O.train_step(cfg, q, st, x, z1, z2, lr)
        assert abs(float(grads[eng.P]) - loss) <= 1e-5 * abs(loss)
    assert np.max(np.abs(host(params) - O.flatten(cfg, q))) <= 0.02 * lr * 3
    # two half-batch shards with the global divisor sum to the full-batch gradient
    full, _ = _grads(eng, cfg, p, x, z1, z2)
    acc = torch.zeros_like(full, dtype=torch.float64)
    for w in range(2):
        s = slice(w * B // 2, (w + 1) * B // 2)
        e = engine_for(cfg, B // 2, world=2, rank=w, global_batch=B)
        acc += _grads(e, cfg, p, x[s], z1[s], z2[s])[0].double()
    assert float((acc - full.double()).abs().max()) <= 2e-5 * float(full[:eng.P].abs().max())
