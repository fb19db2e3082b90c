"""-m gpu: value checks of the kernels that carry the wide configurations (BASELINE config 3, C3) -- the
128 x 128 register-blocked f32 Dense variant for forward, dX and dW|db, the streaming bulk finalize with its
fused Adam branch, and the bf16 matrix-core path at width 512 -- against the float64 oracle, gradient leaf by
gradient leaf (networks.py:99-100).  The in-process profiler's labels prove that the kernel under test is the
one that ran."""
import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.gpu_util import dev, engine_for, host, random_problem

pytestmark = pytest.mark.gpu

SPHERE = dict(name="sphere", seed=69, dd=3, pad=3)


def _leafwise(cfg, eng, got, want_tree):
    """max-abs error of each leaf relative to that leaf's max-abs, and of the whole set relative to the set's."""
    want = O.flatten(cfg, want_tree)
    per_leaf = {}
    for name, (off, shape) in eng.leaves.items():
        n = int(np.prod(shape))
        w, g = want[off:off + n], got[off:off + n]
        per_leaf[name] = float(np.max(np.abs(g - w)) / (np.max(np.abs(w)) + 1e-30))
    overall = float(np.max(np.abs(got[:eng.P] - want)) / np.max(np.abs(want)))
    return per_leaf, overall


def _profiled_grads(eng, params, x, z1, z2):
    grads = eng.new_flat(eng.grad_len)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.profile_begin(256)
    eng.grads_only(params, grads, step, x, z1, z2)
    torch.cuda.synchronize()
    return grads, eng.profile_report()


def test_c3_width512_f32_every_gradient_leaf_on_the_128x128_kernels():
    cfg = O.Config(6, 6, (512, 512, 512), (512, 512, 512), -3.0, True, "sphere")
    B = 16384              # forward / dX: 128 x 4 = 512 tiles of 128 x 128 -> the register-blocked variant fires
    p, x, z1, z2 = random_problem(cfg, SPHERE, B)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    eng = engine_for(cfg, B)
    grads, rep = _profiled_grads(eng, dev(O.flatten(cfg, p)), dev(x), dev(z1), dev(z2))
    for label in ("gemm_f32_fwd_128x128", "gemm_f32_dx_128x128", "gemm_f32_dw_128x128", "bulk_finalize"):
        assert label in rep, (label, sorted(rep))
    got = host(grads)
    assert abs(got[eng.P] - loss) <= 1e-5 * abs(loss)
    per_leaf, overall = _leafwise(cfg, eng, got, g)
    assert overall <= 2e-5, overall
    assert max(per_leaf.values()) <= 1e-4, sorted(per_leaf.items(), key=lambda kv: -kv[1])[:3]


def test_c3_width512_bf16_every_gradient_leaf_in_the_bf16_envelope():
    cfg = O.Config(6, 6, (512, 512, 512), (512, 512, 512), -3.0, True, "sphere")
    B = 16384
    p, x, z1, z2 = random_problem(cfg, SPHERE, B)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    eng = engine_for(cfg, B, dtype="bf16")
    grads, rep = _profiled_grads(eng, dev(O.flatten(cfg, p)), dev(x), dev(z1), dev(z2))
    assert any(k.startswith("gemm_bf16") for k in rep), sorted(rep)
    got = host(grads)
    assert abs(got[eng.P] - loss) <= 2e-3 * abs(loss), (got[eng.P], loss)
    per_leaf, overall = _leafwise(cfg, eng, got, g)
    assert overall <= 2e-2, overall
    assert max(per_leaf.values()) <= 5e-2, sorted(per_leaf.items(), key=lambda kv: -kv[1])[:3]


@pytest.mark.parametrize("latent,hidden,dtype", [
    (6, (256, 256), "f32"),      # tail (epsilon_p, epsilon, loss) finalized + Adam'd in the fused 64-output block
    (64, (256, 256), "f32"),     # L + 5 > 64: the tail takes the separate Adam launch from off_epsp
    (6, (255, 257), "f32"),      # odd layer sizes: segment boundaries off the 16-byte grid -> bulk finalize's scalar tail
    (6, (256, 256), "bf16"),
])
def test_train_step_through_bulk_finalize_adam_matches_oracle(latent, hidden, dtype):
    """vaek_train_step on a model with off_epsp >= 65 536: the weights' slab sum AND their Adam update run in
    bulk_finalize_kernel (csrc/elbo.hip), three consecutive steps against O.train_step (networks.py:100)."""
    cfg = O.Config(6, latent, hidden, hidden, -3.0, True, "sphere")
    B, lr = 256, 1e-3
    assert cfg.n_params() - latent - 1 >= 65536
    p, x, z1, z2 = random_problem(cfg, SPHERE, B)
    eng = engine_for(cfg, B, dtype=dtype)
    assert not eng.fused
    params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    st = O.adam_init(p)
    xd, z1d, z2d = dev(x), dev(z1), dev(z2)
    ltol, ptol, mtol = (1e-5, 0.02, 5e-5) if dtype == "f32" else (2e-3, None, 2e-2)
    for k in range(3):
        p, st, loss = O.train_step(cfg, p, st, x, z1, z2, lr)
        eng.profile_begin(256)
        eng.train_step(params, grads, m, v, step, xd, z1d, z2d, lr)
        torch.cuda.synchronize()
        rep = eng.profile_report()
        assert "bulk_finalize_adam" in rep, sorted(rep)
        assert ("adam" in rep) == (latent + 5 > 64), sorted(rep)
        assert abs(float(grads[eng.P]) - loss) <= ltol * abs(loss), (k, float(grads[eng.P]), loss)
    assert int(step.item()) == 3
    want_p = O.flatten(cfg, p)
    dp = np.abs(host(params) - want_p)
    if dtype == "f32":
        assert np.max(dp) <= ptol * lr
    else:
        # bf16 products perturb every gradient by ~1e-2 relative: where a gradient is near zero Adam's m / sqrt(v) can flip
        # sign, so single parameters may sit up to 2 lr per step off; the bulk must still track the oracle closely
        assert np.max(dp) <= 2.0 * 3 * lr * 1.01 and np.mean(dp) <= 0.05 * lr, (np.max(dp) / lr, np.mean(dp) / lr)
    want_m = O.flatten(cfg, st["m"]); want_v = O.flatten(cfg, st["v"])
    assert np.max(np.abs(host(m) - want_m)) <= mtol * np.max(np.abs(want_m))
    assert np.max(np.abs(host(v) - want_v)) <= mtol * np.max(np.abs(want_v))
