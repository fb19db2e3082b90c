"""-m gpu: edge shapes and error paths of the C ABI (the reference has no tests of its own; these are
the degenerate cases its code would accept: one-row batches, one-dimensional data/latents, the
largest fused shape, the first shape past it, epsilon pinned at 0, fixed decoder variance)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import elbo_oracle as O
from tests.gpu_util import dev, engine_for, host, random_problem, rel_err

pytestmark = pytest.mark.gpu


def _check(cfg, dk, B, expect_fused=None, **kw):
    p, x, z1, z2 = random_problem(cfg, dk, B)
    loss, g = O.loss_and_grad(cfg, p, x, z1, z2)
    eng = engine_for(cfg, B, **kw)
    if expect_fused is not None:
        assert eng.fused == expect_fused
    grads = eng.new_flat(eng.grad_len)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng.grads_only(dev(O.flatten(cfg, p)), grads, step, dev(x), dev(z1), dev(z2))
    got = host(grads)
    assert abs(got[eng.P] - loss) <= 1e-5 * abs(loss)
    want = O.flatten(cfg, g)
    assert rel_err(got[:eng.P], want) <= 2e-5
    return got, want


@pytest.mark.parametrize("impl", ["mfma", "valu"])
@pytest.mark.parametrize("B", [1, 2, 63, 65])
def test_tiny_batches(B, impl):
    _check(O.Config(12, 20, (), (), -1.0, True, "linear_gaussian"), dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=9), B,
           True, fused_impl=impl)
    _check(O.Config(7, 6, (), (), -3.0, True, "sigmoid"), dict(name="sigmoid", seed=69, dd=3, pad=3), B, True, fused_impl=impl)


@pytest.mark.parametrize("impl", ["mfma", "valu", "generic"])
def test_one_dimensional_and_largest_fused_shapes(impl):
    kw = dict(force_generic=True) if impl == "generic" else dict(fused_impl=impl)
    fused = impl != "generic"
    _check(O.Config(1, 1, (), (), 0.5, True, "linear_gaussian"), dict(name="linear_gaussian", seed=1, dd=1, did=1, pad=0), 130, fused, **kw)
    _check(O.Config(32, 32, (), (), -1.0, True, "linear_gaussian"), dict(name="linear_gaussian", seed=1, dd=3, did=3, pad=29), 300, fused, **kw)
    _check(O.Config(32, 32, (), (), -2.0, True, "sigmoid"), dict(name="sigmoid", seed=1, dd=3, pad=28), 300, fused, **kw)
    _check(O.Config(31, 5, (), (), -1.0, False, "linear_gaussian"), dict(name="linear_gaussian", seed=1, dd=3, did=2, pad=28), 77, fused, **kw)


def test_first_shape_past_the_fused_table_runs_layer_by_layer():
    _check(O.Config(33, 8, (), (), -1.0, True, "linear_gaussian"), dict(name="linear_gaussian", seed=1, dd=3, did=3, pad=30), 200, False)
    _check(O.Config(12, 40, (), (), -1.0, True, "linear_gaussian"), dict(name="linear_gaussian", seed=1, dd=3, did=3, pad=9), 200, False)
    _check(O.Config(6, 6, (5,), (3, 2), -3.0, True, "sphere"), dict(name="sphere", seed=1, dd=3, pad=3), 200, False)   # odd tiny widths


@pytest.mark.parametrize("impl", ["mfma", "generic"])
def test_epsilon_zero_and_fixed_decoder_variance(impl):
    kw = dict(force_generic=True) if impl == "generic" else dict(fused_impl=impl)
    cfg = O.Config(12, 20, (), (), 0.0, True, "linear_gaussian")          # CLI default epsilon 0 with -tdv: d/d epsilon == 0
    got, _ = _check(cfg, dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=9), 256, **kw)
    assert got[cfg.n_params() - 1] == 0.0
    _check(O.Config(12, 20, (), (), 0.0, False, "linear_gaussian"), dict(name="linear_gaussian", seed=2, dd=3, did=3, pad=9), 256, **kw)


def test_error_codes_not_aborts():
    from vae_training_amd import _lib
    from vae_training_amd.engine import Engine
    eng = Engine(64, 12, 20)
    lib = eng.lib
    null = C.c_void_p(None)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = eng.new_flat(eng.grad_len)
    assert lib.vaek_train_step(eng.h, null, null, null, null, null, null, null, null, 1e-3, null, st) == -1
    ws_bad = C.c_void_p(eng.workspace.data_ptr() + 4)                      # misaligned workspace
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    x = torch.zeros(64, 12, device="cuda"); z1 = torch.zeros(64, 20, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.vaek_train_step(eng.h, p(f), p(f), p(f), p(f), p(step), p(x), p(z1), p(x), 1e-3, ws_bad, st) == -4
    assert b"workspace" in lib.vaek_last_error()
    assert lib.vaek_adam_step(eng.h, p(f), p(f), p(f), p(f), 10, 1e-3, 0, null, 1.0, st) == -1      # step < 1 without a device counter
    assert lib.vaek_comm_allreduce(eng.h, p(f), 10, st) == -5                                    # no communicator
    assert lib.vaek_make_batch(eng.h, 0, null, 3, 3, 8, 0.0, p(x), p(z1), p(x), 64, 0, 1, null, 0, 0, st) == -1   # D mismatch / null A
    eng2 = Engine(64, 12, 20, world=2, rank=1, global_batch=128)
    with pytest.raises(_lib.VaekError, match="communicator"):
        eng2.train_step(f[:eng.P].clone(), f, f[:eng.P].clone(), f[:eng.P].clone(), step, x, z1, x, 1e-3)


def test_launch_floor_probe():
    """vaek_microbench_launch: the empty kernel and the one-dependent-load kernel run; the graph interval is microseconds."""
    import ctypes as C
    from vae_training_amd import _lib
    from vae_training_amd.engine import Engine
    eng = Engine(64, 12, 20)
    p = torch.zeros(128, dtype=torch.int32, device="cuda")
    out = torch.zeros(16, dtype=torch.int32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(eng.lib.vaek_microbench_launch(eng.h, 1, 9, 3, C.c_void_p(p.data_ptr()), C.c_void_p(out.data_ptr()), st))
    torch.cuda.synchronize()
    assert int(out.abs().sum()) == 0
    assert eng.lib.vaek_microbench_launch(eng.h, 1, 9, 3, None, None, st) != 0          # kind 1 needs its buffers
    us = eng.measure_launch_floor(n=50, reps=5)
    assert 0.2 < us < 50.0, us
