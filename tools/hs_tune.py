"""Time the tile / ring variants of the bf16-storage GEMMs (csrc/gemm_bf16s.hip) on the C3 step and check that every variant
produces bitwise the gradient of variant 0 (same k order, so nothing may differ).  Usage: python tools/hs_tune.py [B]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import WORKLOADS, init_params_flat, make_batches, data_dim  # noqa: E402
from vae_training_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
SINGLE = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else None      # profile one (nt, tn) pair only
w = WORKLOADS["C3"]
D, L = data_dim(w), w["L"]
eng = Engine(B, D, L, w["enc"], w["dec"], w["eps"], w["tdv"], False, dtype="bf16")
lib = eng.lib
n_nt, n_tn = C.c_int(), C.c_int()
lib.vaek_debug_hs_variant(-1, -1, C.byref(n_nt), C.byref(n_tn))
params = init_params_flat(eng, 0)
x, z1, z2 = make_batches(w, B, eng.device, 1, seed=5)[0]
step = torch.zeros(1, dtype=torch.int32, device="cuda")


def run(nt, tn, reps=6):
    assert lib.vaek_debug_hs_variant(nt, tn, None, None) == 0
    g = eng.new_flat(eng.grad_len)
    for _ in range(2):
        eng.grads_only(params, g, step, x, z1, z2)
    torch.cuda.synchronize()
    eng.profile_begin(4096)
    for _ in range(reps):
        eng.grads_only(params, g, step, x, z1, z2)
    torch.cuda.synchronize()
    rep = eng.profile_report()
    return g, {k: v["total_ms"] / v["count"] * 1e3 for k, v in rep.items()}, sum(v["total_ms"] for v in rep.values()) / reps * 1e3


if SINGLE is not None:
    _, t0, tot0 = run(*SINGLE, reps=3)
    print(f"variant nt={SINGLE[0]} tn={SINGLE[1]}: step kernels {tot0:.0f} us; per launch:", {k: round(v, 1) for k, v in t0.items()})
    sys.exit(0)
ref, t0, tot0 = run(0, 0)
print(f"variant nt=0 tn=0: step kernels {tot0:.0f} us; per launch:", {k: round(v, 1) for k, v in t0.items() if "bf16s" in k})
for nt in ([int(a) for a in os.environ["HS_NT"].split(",")] if os.environ.get("HS_NT") else range(1, n_nt.value)):
    g, t, tot = run(nt, 0)
    print(f"nt={nt}: fwd {t['gemm_bf16s_fwd']:.1f} dx {t['gemm_bf16s_dx']:.1f} us   identical={torch.equal(g, ref)}", flush=True)
for tn in ([int(a) for a in os.environ["HS_TN"].split(",")] if os.environ.get("HS_TN") is not None and os.environ.get("HS_TN") != "" else (range(1, n_tn.value) if os.environ.get("HS_TN") is None else [])):
    g, t, tot = run(0, tn)
    print(f"tn={tn}: dw {t['gemm_bf16s_dw']:.1f} us   identical={torch.equal(g, ref)}", flush=True)
