# Diagnostic build (-DVAEK_LIN_STAMPS): where the updater workgroup of vaek_train_steps spends its cycles.
set -e
cd $GRAFT_REPO_ROOT/vae_training_amd/csrc
mkdir -p /tmp/linst && for f in api gemm_f32 gemm_bf16 gemm_bf16s gemm_skinny16 linear_moments elbo fused_small fused_mfma comm rng microbench; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DVAEK_LIN_STAMPS -c $f.hip -o /tmp/linst/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/linst/libvaek.so /tmp/linst/*.o
cd $GRAFT_REPO_ROOT
VAEK_LIB_PATH=/tmp/linst/libvaek.so VAEK_LIN_ROLES=${ROLES:-4} python3 - <<'PY'
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from bench import WORKLOADS, data_dim, init_params_flat, make_batches
from vae_training_amd.engine import Engine
w = WORKLOADS["M"]; B = 65536
eng = Engine(B, data_dim(w), w["L"], (), (), w["eps"], w["tdv"], False)
params = init_params_flat(eng, 0); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
step = torch.zeros(1, dtype=torch.int32, device="cuda")
batches = make_batches(w, B, eng.device, 4, seed=1)
buf = torch.zeros(16, dtype=torch.int64, device="cuda")
assert eng.lib.vaek_debug_lin_stamps(C.c_void_p(buf.data_ptr())) == 0
for _ in range(3):
    eng.train_steps(params, grads, m, v, step, batches, 1e-3)
torch.cuda.synchronize()
t = buf.cpu().numpy()
names = ["load params + expand M", "e^{lv/2}", "build R, E", "P1 = R M, Q = E M", "G, dWd, partial sums", "tree reduce", "outputs + Adam"]
print("updater phases (cycles):")
for i, n in enumerate(names):
    print(f"   {n:28s} {t[i + 1] - t[i]:8d}")
print("   total", t[7] - t[0])
PY
