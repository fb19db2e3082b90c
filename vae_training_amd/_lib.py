"""ctypes binding of include/vaek.h (the drop-in boundary).  Loads the in-tree libvaek.so."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VAEK_LIB_PATH") or os.path.join(_HERE, "csrc", "libvaek.so")      # override: A/B builds, diagnostics
VAEK_MAX_HIDDEN = 8
VAEK_F32, VAEK_BF16 = 0, 1
VAEK_ACT_NONE, VAEK_ACT_RELU = 0, 1


class VaekError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvaek error {code}: {msg}")
        self.code = code


class VaekConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("batch", C.c_int32), ("data_dim", C.c_int32), ("latent_dim", C.c_int32),
        ("n_enc_hidden", C.c_int32), ("enc_hidden", C.c_int32 * VAEK_MAX_HIDDEN),
        ("n_dec_hidden", C.c_int32), ("dec_hidden", C.c_int32 * VAEK_MAX_HIDDEN),
        ("sigmoid_decoder", C.c_int32), ("tunable_eps", C.c_int32), ("eps_cli", C.c_float),
        ("dtype", C.c_int32), ("device", C.c_int32), ("world", C.c_int32), ("rank", C.c_int32),
        ("global_batch", C.c_int64), ("force_generic", C.c_int32), ("reserved", C.c_int32 * 7),
    ]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); exactly the symbols include/vaek.h declares
SIGNATURES = {
    "vaek_version": (C.c_int, []),
    "vaek_last_error": (C.c_char_p, []),
    "vaek_ctx_create": (C.c_int, [C.POINTER(VaekConfig), C.POINTER(_vp)]),
    "vaek_ctx_destroy": (C.c_int, [_vp]),
    "vaek_param_count": (C.c_int, [_vp, C.POINTER(_i64)]),
    "vaek_grad_len": (C.c_int, [_vp, C.POINTER(_i64)]),
    "vaek_leaf_count": (C.c_int, [_vp, C.POINTER(_i32)]),
    "vaek_leaf_info": (C.c_int, [_vp, _i32, C.c_char_p, _i32, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32)]),
    "vaek_workspace_bytes": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "vaek_uses_fused_path": (C.c_int, [_vp, C.POINTER(_i32)]),
    "vaek_dense_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vaek_dense_bwd_dx": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vaek_dense_bwd_dw": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "vaek_elbo_fwd_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "vaek_elbo_fwd_bwd_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "vaek_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _i32, _vp, _f32, _vp]),
    "vaek_train_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp]),
    "vaek_train_step_grads_only": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vaek_train_step_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp]),
    "vaek_bucket_count": (C.c_int, [_vp, C.POINTER(_i32)]),
    "vaek_bucket_info": (C.c_int, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "vaek_train_step_grads_bucketed": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vaek_loss_eval": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vaek_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp, _i32, _vp, _vp]),
    "vaek_comm_buffer_bytes": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "vaek_comm_create": (C.c_int, [_vp, _vp]),
    "vaek_comm_init": (C.c_int, [_vp, _vp]),
    "vaek_comm_status": (C.c_int, [_vp, C.POINTER(_i32)]),
    "vaek_comm_destroy": (C.c_int, [_vp]),
    "vaek_comm_allreduce": (C.c_int, [_vp, _vp, _i64, _vp]),
    "vaek_make_batch": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _i32, _i64, C.c_uint64, _vp,
                                  C.c_uint32, C.c_uint32, _vp]),
    "vaek_make_batch_next": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _i32, _i64, C.c_uint64, _vp,
                                       _i32, C.c_uint32, _vp]),
    "vaek_train_step_gen": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _vp, _i32, _i32, _i32, _f32,
                                      _vp, _vp, _vp, _i64, C.c_uint64, _vp, _i32, C.c_uint32, _vp]),
    "vaek_supports_train_steps": (C.c_int, [_vp, C.POINTER(_i32)]),
    "vaek_train_steps": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp]),
    "vaek_train_steps_status": (C.c_int, [_vp, _vp, C.POINTER(_i32)]),
    "vaek_supports_train_steps_gen": (C.c_int, [_vp, _i32, C.POINTER(_i32)]),
    "vaek_train_steps_moment_len": (C.c_int, [_vp, C.POINTER(_i64)]),
    "vaek_train_steps_moments": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vaek_train_steps_update": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp]),
    "vaek_train_steps_gen": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _f32, C.c_int64, C.c_uint64, C.c_uint32, _i32,
                                       _f32, _vp, _vp]),
    "vaek_conv2d_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "vaek_to_bf16": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "vaek_conv2d_forward_workspace": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_size_t)]),
    "vaek_conv2d_bias_grad": (C.c_int, [_vp, _vp, _vp, C.c_int64, _i32, _vp]),
    "vaek_conv2d_bias_grad_bf16": (C.c_int, [_vp, _vp, _vp, C.c_int64, _i32, _vp]),
    "vaek_dense_fwd_reparam": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "vaek_reparam_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_int64, _vp, _vp]),
    "vaek_conv2d_transpose_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "vaek_conv2d_weight_grad_workspace": (C.c_int, [_i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_size_t)]),
    "vaek_conv2d_weight_grad": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vaek_rng_fill": (C.c_int, [_vp, _vp, _vp, _i64, C.c_uint64, C.c_uint32, C.c_uint32, _vp]),
    "vaek_set_loss_history": (C.c_int, [_vp, _vp, _i64]),
    "vaek_microbench_copy": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "vaek_microbench_launch": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vaek_microbench_mfma": (C.c_int, [_vp, _i32, _i32, _i32, _vp, C.POINTER(C.c_double), _vp]),
    "vaek_profile_begin": (C.c_int, [_vp, _i32]),
    "vaek_profile_report": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
}

_lib = None


def load():
    """Load libvaek.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. vae_training_amd has no CPU/PyTorch fallback.")
    # torch first: libvaek shares device pointers and streams with PyTorch-ROCm, so both must use
    # the one HIP runtime torch ships (same SONAME libamdhip64.so.7 -- whoever loads first wins)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise VaekError(rc, load().vaek_last_error().decode("utf-8", "replace"))
