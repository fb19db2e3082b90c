"""-m gpu: the bf16 matrix-core Dense path (vaek_config.dtype = VAEK_BF16).  bf16 products carry 2^-9
relative rounding per operand, so this path is NOT held to the 1e-5 ELBO contract (that is the f32
path's, tests/test_gpu_parity.py); the tolerances below are its measured envelope against the float64
oracle on the same inputs: loss 2e-3 relative, gradient 2e-2 of the leaf-set max-abs."""
import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.gpu_util import dev, engine_for, host, random_problem, rel_err

pytestmark = pytest.mark.gpu


# (128,): one hidden layer -> only skinny layers, exact f32 kernels.  (256, 128), (192, 64, 192): every hidden width a multiple
# of 64 -> the bf16-STORAGE mode (gemm_bf16s.hip: bf16 activations / gradients in HBM, direct-to-LDS loads, transposed LDS
# reads for dW).  (200, 200): widths off the 64 grid -> f32 storage with the round-1 kernels (gemm_bf16.hip).
# (512, 512) at 2 048 and 2 000 rows: C3's widths -- the decoder's last layer backward on the matrix cores (sk_last_bwd_mfma_kernel:
# H = 512, whole 16-row blocks per workgroup) and, with a ragged workgroup split, the per-lane form beside it.
@pytest.mark.parametrize("hidden,B", [((128,), 700), ((256, 128), 1000), ((192, 64, 192), 333), ((200, 200), 500), ((128, 128), 64 * 9 + 1),
                                      ((512, 512), 2048), ((512, 512), 2000)])
@pytest.mark.parametrize("dataset", ["sphere", "sigmoid"])
def test_bf16_dense_path_tracks_oracle(hidden, B, dataset):
    D = 7 if dataset == "sigmoid" else 6
    cfg = O.Config(D, 6, hidden, hidden, -3.0, True, dataset)
    dk = dict(name=dataset, seed=69, dd=3, pad=3)
    p, x, z1, z2 = random_problem(cfg, dk, B)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    eng = engine_for(cfg, B, dtype="bf16")
    assert not eng.fused
    params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.profile_begin(256)
    eng.grads_only(params, grads, step, dev(x), dev(z1), dev(z2))
    torch.cuda.synchronize()
    rep = eng.profile_report()
    storage_mode = len(hidden) >= 2 and all(h % 64 == 0 for h in hidden)
    assert any(k.startswith("gemm_bf16s_") for k in rep) == storage_mode, sorted(rep)
    if len(hidden) >= 2 and not storage_mode:
        assert any(k.startswith("gemm_bf16_") for k in rep), sorted(rep)
    got = host(grads)
    assert abs(got[eng.P] - loss) <= 2e-3 * abs(loss), (got[eng.P], loss)
    assert rel_err(got[:eng.P], O.flatten(cfg, g)) <= 2e-2
    # and it is genuinely a different arithmetic from the f32 path (the bf16 kernels ran)
    e32 = engine_for(cfg, B, force_generic=True)          # (one-hidden-layer f32 models otherwise take the whole-network kernel)
    g32 = e32.new_flat(e32.grad_len)
    e32.grads_only(params, g32, step, dev(x), dev(z1), dev(z2))
    if len(hidden) >= 2:                       # a layer with both dims >= 64 exists -> the bf16 kernels ran
        assert not torch.equal(g32, grads)
    else:                                      # only skinny layers: they stay on the exact f32 kernels by design
        assert torch.equal(g32, grads)
    assert abs(host(g32)[eng.P] - loss) <= 1e-5 * abs(loss)


def test_bf16_train_steps_reduce_loss():
    cfg = O.Config(6, 6, (256, 256), (256, 256), -3.0, True, "sphere")
    B, lr = 4096, 1e-3
    _, sampler = O.make_dataset("sphere", 69, dd=3, pad=3)
    rng = np.random.default_rng(0)
    eng = engine_for(cfg, B, dtype="bf16")
    p = O.init_params(cfg, seed=0)
    params = dev(O.flatten(cfg, p)); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    losses = []
    for s in range(40):
        z = rng.standard_normal((B, 12))
        eng.train_step(params, grads, m, v, step, dev(sampler(rng, B)), dev(z[:, :6]), dev(z[:, 6:]), lr)
        losses.append(float(grads[eng.P]))
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5])


def test_bf16s_variants_are_bitwise_identical():
    """Every tile / ring variant of the bf16-storage GEMMs (csrc/gemm_bf16s.hip: 128- and 256-wide tiles, 2- to 6-deep LDS
    rings, the persistent staggered form) accumulates the same k order, so the whole gradient must be BITWISE the default's --
    which the tests above hold to the oracle.  4 096 rows for all of them; 65 536 rows (tiles > workgroups: the persistent
    kernel crosses tile seams with loads in flight) for the persistent form."""
    import ctypes as C
    cfg = O.Config(6, 6, (256, 512, 256), (256, 256), -3.0, True, "sphere")
    n_nt, n_tn = C.c_int(), C.c_int()
    for B, nts in ((4096, None), (65536, (7, 11))):
        p, x, z1, z2 = random_problem(cfg, dict(name="sphere", seed=69, dd=3, pad=3), B)
        eng = engine_for(cfg, B, dtype="bf16")
        lib = eng.lib
        assert lib.vaek_debug_hs_variant(-2, 0, C.byref(n_nt), C.byref(n_tn)) == 0
        params, xd, z1d, z2d = dev(O.flatten(cfg, p)), dev(x), dev(z1), dev(z2)
        step = torch.zeros(1, dtype=torch.int32, device="cuda")

        def grads():
            g = eng.new_flat(eng.grad_len)
            eng.grads_only(params, g, step, xd, z1d, z2d)
            torch.cuda.synchronize()
            return g
        try:
            ref = grads()
            assert torch.isfinite(ref).all()
            for nt in (range(n_nt.value) if nts is None else nts):
                assert lib.vaek_debug_hs_variant(nt, 0, None, None) == 0
                assert torch.equal(grads(), ref), f"NT variant {nt} differs at B={B}"
            assert lib.vaek_debug_hs_variant(-2, 0, None, None) == 0
            if nts is None:
                for tn in range(n_tn.value):
                    assert lib.vaek_debug_hs_variant(-1, tn, None, None) == 0
                    assert torch.equal(grads(), ref), f"TN variant {tn} differs"
        finally:
            lib.vaek_debug_hs_variant(-2, 0, None, None)
