# Diagnostic build (-DVAEK_M1_STAMPS): where one workgroup of the one-hidden-layer whole-network kernel spends its time (C2).
set -e
cd $GRAFT_REPO_ROOT/vae_training_amd/csrc
mkdir -p /tmp/m1st && for f in $(ls *.hip | sed "s/\.hip$//"); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DVAEK_M1_STAMPS -c $f.hip -o /tmp/m1st/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/m1st/libvaek.so /tmp/m1st/*.o
cd $GRAFT_REPO_ROOT
VAEK_LIB_PATH=/tmp/m1st/libvaek.so python3 - <<'PY'
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from bench import WORKLOADS, data_dim, init_params_flat, make_batches
from vae_training_amd.engine import Engine
w = WORKLOADS["C2"]; B = w["batch"]
eng = Engine(B, data_dim(w), w["L"], w["enc"], w["dec"], w["eps"], w["tdv"], w["dataset"] == "sigmoid")
params = init_params_flat(eng, 0); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
step = torch.zeros(1, dtype=torch.int32, device="cuda")
batches = make_batches(w, B, eng.device, 4, seed=1)
buf = torch.zeros(32, dtype=torch.int64, device="cuda")
eng.lib.vaek_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert eng.lib.vaek_debug_set_stamps(eng.h, C.c_void_p(buf.data_ptr())) == 0
for i in range(6):
    eng.train_step(params, grads, m, v, step, *batches[i % 4], 1e-4)
torch.cuda.synchronize()
t = buf.cpu().numpy()
names = ["weights, small vectors, exp -> registers", "inputs -> LDS", "encoder layer 1", "mu (MFMA) + reparam", "decoder(s) layer 1",
         "x_hat (MFMA passes) + sigmoid", "ELBO elementwise", "decoder(s) backward", "d samples (MFMA passes)", "encoder backward",
         "small vectors", "partial row + scalar sums"]
print("workgroup 7 of fused_mlp1 (us, s_memrealtime; each stamp drains the wave's memory operations first):")
for i, n in enumerate(names):
    print(f"   {n:42s} {(t[i + 1] - t[i]) / 100.0:7.2f}")
print(f"   total {(t[12] - t[0]) / 100.0:.2f}")
PY
