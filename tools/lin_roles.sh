# Diagnostic: the persistent form of vaek_train_steps with some roles left out (VAEK_LIN_ROLES: 1 streamers only, 3 + reducers,
# 7 everything), 64 steps of the metric's shape per call, arguments marshalled once (Engine.plan_train_steps), events around ten
# calls: us per step, median and best.  Results are garbage unless ROLES=7.
cd $GRAFT_REPO_ROOT
for roles in ${ROLES_LIST:-1 3 7}; do
VAEK_LIN_ROLES=$roles python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import WORKLOADS, data_dim, init_params_flat, make_batches
from vae_training_amd.engine import Engine
w = WORKLOADS["M"]; B = int(os.environ.get("LIN_B", 65536)); N = int(os.environ.get("LIN_NSTEPS", 64))
eng = Engine(B, data_dim(w), w["L"], (), (), w["eps"], w["tdv"], False)
params = init_params_flat(eng, 0); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
step = torch.zeros(1, dtype=torch.int32, device="cuda")
batches = make_batches(w, B, eng.device, 48, seed=1)
# (several plans walking the 48-batch rotation: a short call's inputs must not sit in the Infinity Cache from the call before)
plans = [eng.plan_train_steps(params, grads, m, v, step, [batches[(o + i) % len(batches)] for i in range(N)], 1e-3) for o in range(0, 48, 16)]
for it in range(3):
    plans[it % 3]()
torch.cuda.synchronize()
out = []
for it in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plans[it % 3]()
    e1.record()
    torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) * 1e3 / N)
print("roles", os.environ["VAEK_LIN_ROLES"], f"{N} steps per call, us/step: median {np.median(out):.2f} best {min(out):.2f}")
PY
done
