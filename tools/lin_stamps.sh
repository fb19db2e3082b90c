# Diagnostic build (-DVAEK_LIN_STAMPS): where the updater workgroup of vaek_train_steps spends its cycles.
set -e
cd $GRAFT_REPO_ROOT/vae_training_amd/csrc
mkdir -p /tmp/linst && for f in $(ls *.hip | sed "s/\.hip$//"); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DVAEK_LIN_STAMPS=${STAMPS:-1} -c $f.hip -o /tmp/linst/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/linst/libvaek.so /tmp/linst/*.o
cd $GRAFT_REPO_ROOT
# PERSIST=1: the persistent form with all roles (stamps of the LAST step of the launch; 9 = params published, 8 = wait over)
if [ "${PERSIST:-0}" = "1" ]; then export VAEK_LIN_PERSIST=1; else export VAEK_LIN_PERSIST=0 VAEK_LIN_ROLES=${ROLES:-4}; fi
VAEK_LIB_PATH=/tmp/linst/libvaek.so python3 - <<'PY'
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from bench import WORKLOADS, data_dim, init_params_flat, make_batches
from vae_training_amd.engine import Engine
w = WORKLOADS["M"]; B = 65536
persist = os.environ.get("VAEK_LIN_PERSIST") == "1"
eng = Engine(B, data_dim(w), w["L"], (), (), w["eps"], w["tdv"], False)
params = init_params_flat(eng, 0); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
step = torch.zeros(1, dtype=torch.int32, device="cuda")
batches = make_batches(w, B, eng.device, 48 if persist else 4, seed=1)
buf = torch.zeros(256, dtype=torch.int64, device="cuda")
assert eng.lib.vaek_debug_lin_stamps(C.c_void_p(buf.data_ptr())) == 0
NS = int(os.environ.get("LIN_NSTEPS", 64))
off = 0
for nsteps in ((40, NS, NS) if persist else (4, 4, 4)):
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if os.environ.get("LIN_GEN") == "1":       # the drawing form (vaek_train_steps_gen)
        A = torch.randn(3, 3, generator=torch.Generator().manual_seed(2)).cuda().contiguous()
        eng.train_steps_gen(params, grads, m, v, step, nsteps, 1e-3, 0, A, 3, 3, 9, 0.0, 7)
    else:
        eng.train_steps(params, grads, m, v, step, [batches[(off + i) % len(batches)] for i in range(nsteps)], 1e-3)      # (the rotation keeps a short call's inputs out of the Infinity Cache)
    off += nsteps
    e1.record()
    torch.cuda.synchronize()
    print(f"{nsteps} steps: {e0.elapsed_time(e1) * 1e3 / nsteps:.2f} us/step (events around the call)")
t = buf.cpu().numpy()
if persist:
    assert not eng.train_steps_gave_up()
    print(f"   loop entry -> end of last step: {(t[16 + 7] - t[16 + 10]) / 100.0:.1f} us (s_memrealtime, 100 MHz) = {(t[7] - t[10])} s_memtime ticks")
    print(f"   last step: {(t[16 + 7] - t[16 + 0]) / 100.0:.2f} us")
    print(f"   waiting for the reducers: total {t[11] / 100.0:.1f} us, first step {t[13] / 100.0:.1f} us, longest {t[12] / 100.0:.1f} us")
    t0 = t[16 + 10]
    print(f"   reducer workgroup 5 (32 of the launch's batches): waiting {t[40] / 100.0:.1f} us, reducing {t[41] / 100.0:.1f} us")
    print(f"   streamer 7 ({t[46]} work items), us per item: counted wait {t[42] / 100.0 / t[46]:.2f}, barrier {t[43] / 100.0 / t[46]:.2f}, ragged fix {t[44] / 100.0 / t[46]:.2f}, "
          f"products (pieces of a later tile issued between the k-steps) {t[47] / 100.0 / t[46]:.2f}, barrier {t[48] / 100.0 / t[46]:.2f}, cross-wave sum + store {(t[45] - t[47] - t[48]) / 100.0 / t[46]:.2f}")
    print(f"   streamers 0 / S/2 / S-1 entered the kernel at {(t[50] - t0) / 100.0:.1f} / {(t[51] - t0) / 100.0:.1f} / {(t[52] - t0) / 100.0:.1f} us, signalled batch 0 at {(t[53] - t0) / 100.0:.1f} / {(t[54] - t0) / 100.0:.1f} / {(t[55] - t0) / 100.0:.1f} us")
    print("   reducer 5, batches 0..2: woke at / done at (us): " + "  ".join(f"{(t[56 + 2 * n] - t0) / 100.0:.1f} / {(t[57 + 2 * n] - t0) / 100.0:.1f}" for n in range(3)))
    print("   batch: reduced at | updated at  (us after the updater entered its loop; last arrival of each role)")
    for n in range(0, nsteps, 1 if nsteps <= 24 else 3):
        print(f"   {n:3d}   {(t[128 + n] - t0) / 100.0:8.1f}   {(t[64 + n] - t0) / 100.0:8.1f}")
    print("updater, last step of the launch (s_memtime ticks):")
    print(f"   publish params               {t[9] - t[0]:8d}")
    print(f"   wait for the reducers        {t[8] - t[9]:8d}")
    print(f"   M into registers             {t[1] - t[8]:8d}")
else:
    print("updater phases (ticks):")
    print(f"   load params + expand M       {t[1] - t[0]:8d}")
names = (["SM, P1 (wave 0)", "barrier", "G, dWd, gradients, partial sums", "barrier", "outputs + Adam"] if persist else
         ["e^{lv/2}", "SM", "P1", "G, dWd, partial sums", "tree reduce", "outputs + Adam"])
for i, n in enumerate(names):
    print(f"   {n:28s} {t[i + 2] - t[i + 1]:8d}")
print("   total", t[7] - t[0])
PY
