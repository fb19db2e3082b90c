"""Phase breakdown of fused_linear_kernel from in-kernel s_memtime stamps.
Build the diagnostic library first (tools/stamps.sh); never quote this build's run time."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vae_training_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvaek_stamps.so")
from vae_training_amd.engine import Engine
import bench

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "M"]
B = w["batch"]
eng = Engine(B, bench.data_dim(w), w["L"], w["enc"], w["dec"], w["eps"], w["tdv"], w["dataset"] == "sigmoid")
lib = eng.lib
lib.vaek_debug_set_stamps.argtypes = [C.c_void_p, C.c_void_p]
nblk = 512
stamps = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device="cuda")
lib.vaek_debug_set_stamps(eng.h, C.c_void_p(stamps.data_ptr()))
params = bench.init_params_flat(eng)
grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
step = torch.zeros(1, dtype=torch.int32, device="cuda")
batches = bench.make_batches(w, B, eng.device, 2, 1)
for i in range(10):
    eng.train_step(params, grads, m, v, step, *batches[i % 2], w["lr"])
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(nblk, 4, 8)[:256]
names = ["prologue", "inputs landed", "mu+samples", "y+dy", "g+dmu+image", "MFMA loop", "epilogue"]
d = np.diff(s[:, :, :7], axis=2).astype(np.float64)
print("cycles per phase (median over 1024 waves; s_memtime ticks = shader cycles... 100MHz if constant):")
for i in range(6):
    print(f"  {names[i]:>14s} -> {names[i+1]:<14s} median {np.median(d[:,:,i]):9.0f}  p10 {np.percentile(d[:,:,i],10):9.0f}  p90 {np.percentile(d[:,:,i],90):9.0f}")
print("  total", np.median(s[:, :, 6] - s[:, :, 0]))
print("  first-start to last-end across grid:", s[:, :, 6].max() - s[:, :, 0].min())
