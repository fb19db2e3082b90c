"""-m gpu: the BASELINE.json configurations at their FULL sizes.  Where the float64 oracle finishes in
seconds (M, C2, C4: linear or narrow) the comparison is direct; for C3 (65 536 x 512-wide MLP) the
oracle runs the forward/loss only and the backward is checked through size-independent properties:
bitwise repeatability and shard additivity (the sum of two half-batch gradients taken with the global
divisor equals the full-batch gradient)."""
import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.gpu_util import dev, engine_for, host, random_problem, rel_err

pytestmark = pytest.mark.gpu


def _grads(eng, cfg, p, x, z1, z2):
    grads = eng.new_flat(eng.grad_len)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.grads_only(dev(O.flatten(cfg, p)), grads, step, dev(x), dev(z1), dev(z2))
    return grads


def test_M_metric_config_batch_65536():
    cfg = O.Config(12, 20, (), (), -1.0, True, "linear_gaussian")            # seed_linpadding_expts.sh:1
    p, x, z1, z2 = random_problem(cfg, dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=9), 65536)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    eng = engine_for(cfg, 65536)
    assert eng.fused
    got = host(_grads(eng, cfg, p, x, z1, z2))
    assert abs(got[eng.P] - loss) <= 1e-5 * abs(loss) and rel_err(got[:eng.P], O.flatten(cfg, g)) <= 2e-5
    # three Adam steps against the oracle at full size
    st = O.adam_init(p)
    params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(3):
        p, st, l = O.train_step(cfg, p, st, x, z1, z2, 1e-3)
        eng.train_step(params, grads, m, v, step, dev(x), dev(z1), dev(z2), 1e-3)
        assert abs(float(grads[eng.P]) - l) <= 1e-5 * abs(l)
    assert np.max(np.abs(host(params) - O.flatten(cfg, p))) <= 0.02 * 1e-3


def test_C2_sigmoid_width256_batch_8192():
    cfg = O.Config(7, 6, (256,), (256,), -3.0, True, "sigmoid")
    assert cfg.n_params() == 10779
    p, x, z1, z2 = random_problem(cfg, dict(name="sigmoid", seed=69, dd=3, pad=3), 8192)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    got = host(_grads(engine_for(cfg, 8192), cfg, p, x, z1, z2))
    assert abs(got[cfg.n_params()] - loss) <= 1e-5 * abs(loss) and rel_err(got[:cfg.n_params()], O.flatten(cfg, g)) <= 2e-5


def test_C4_linear_ambient_4096_batch_32768():
    cfg = O.Config(4096, 20, (), (), -1.0, True, "linear_gaussian")
    assert cfg.n_params() == 167977
    p, x, z1, z2 = random_problem(cfg, dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=4093), 32768)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    got = host(_grads(engine_for(cfg, 32768), cfg, p, x, z1, z2))
    assert abs(got[cfg.n_params()] - loss) <= 1e-5 * abs(loss) and rel_err(got[:cfg.n_params()], O.flatten(cfg, g)) <= 2e-5


@pytest.mark.parametrize("dtype,ltol", [("f32", 1e-5), ("bf16", 2e-3)])
def test_C3_sphere_512x3_batch_65536_properties(dtype, ltol):
    cfg = O.Config(6, 6, (512, 512, 512), (512, 512, 512), -3.0, True, "sphere")
    assert cfg.n_params() == 1063955
    B = 65536
    p, x, z1, z2 = random_problem(cfg, dict(name="sphere", seed=69, dd=3, pad=3), B)
    ev = O.loss_eval(cfg, p, x, z1, z2)                                     # oracle forward + loss at full size
    eng = engine_for(cfg, B, dtype=dtype)
    g1 = _grads(eng, cfg, p, x, z1, z2)
    assert abs(float(g1[eng.P]) - ev[0]) <= ltol * abs(ev[0])
    assert torch.equal(g1, _grads(eng, cfg, p, x, z1, z2))                 # bitwise repeatable
    acc = torch.zeros_like(g1, dtype=torch.float64)
    for w in range(2):                                                      # shard additivity, global divisor
        s = slice(w * B // 2, (w + 1) * B // 2)
        e = engine_for(cfg, B // 2, dtype=dtype, world=2, rank=w, global_batch=B)
        acc += _grads(e, cfg, p, x[s], z1[s], z2[s]).double()
    assert rel_err(acc.cpu().numpy(), host(g1)) <= (5e-6 if dtype == "f32" else 5e-3)
