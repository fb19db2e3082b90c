# Diagnostic: metric kernel with everything but dispatch + loads + partial-row store removed (-DVAEK_ABLATE).
set -e
cd $(dirname $0)/../vae_training_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DVAEK_FUSED_ONLY_M"
rm -rf /tmp/vaek_ab && mkdir -p /tmp/vaek_ab          # a private directory: every object of csrc/, the ablated fused_mfma.o compiled last
for f in $(ls *.hip | sed "s/\.hip$//" | grep -v "^fused_mfma$"); do /opt/rocm/bin/hipcc $F -c $f.hip -o /tmp/vaek_ab/$f.o & done; wait
/opt/rocm/bin/hipcc $F -DVAEK_ABLATE=${ABL:-1} -c fused_mfma.hip -o /tmp/vaek_ab/fused_mfma.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvaek_ablate.so /tmp/vaek_ab/*.o
cd ../..
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from vae_training_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvaek_ablate.so")
import torch, bench
from vae_training_amd.engine import Engine
w = bench.WORKLOADS["M"]; B = w["batch"]
for waves in (4,):
    eng = Engine(B, 12, 20, (), (), -1.0, True, False)
    params = bench.init_params_flat(eng); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    batches = bench.make_batches(w, B, eng.device, 4, 1)
    for i in range(20): eng.grads_only(params, grads, step, *batches[i % 4])
    eng.profile_begin(4096)
    for i in range(200): eng.grads_only(params, grads, step, *batches[i % 4])
    torch.cuda.synchronize()
    rep = eng.profile_report()
    print("ablated (loads + store only), waves", waves, {k: round(r["total_ms"] / r["count"] * 1e3, 2) for k, r in rep.items()})
PY
