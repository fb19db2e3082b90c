"""Staged smoke scenarios for vaek_train_steps' persistent launch, each in its own process (a GPU fault ends only that one):
    python3 tools/lin_debug.py            # runs all scenarios, prints one line each
    python3 tools/lin_debug.py <name>     # one scenario in this process"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCEN = ["eager5", "eager20x2", "eager64", "graph20", "profile20", "small16", "small1000x7"]


def run(name):
    import numpy as np
    import torch
    from bench import WORKLOADS, data_dim, init_params_flat, make_batches
    from vae_training_amd.engine import Engine
    w = WORKLOADS["M"]
    B = {"small16": 16, "small1000x7": 1000}.get(name, 65536)
    eng = Engine(B, data_dim(w), w["L"], (), (), w["eps"], w["tdv"], False)
    params = init_params_flat(eng, 0); grads = eng.new_flat(eng.grad_len); m = eng.new_flat(); v = eng.new_flat()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    batches = make_batches(w, B, eng.device, 24, seed=1)
    grp = lambda n: [batches[i % len(batches)] for i in range(n)]
    if name == "eager5":
        eng.train_steps(params, grads, m, v, step, grp(5), 1e-3)
    elif name == "eager20x2":
        eng.train_steps(params, grads, m, v, step, grp(20), 1e-3)
        eng.train_steps(params, grads, m, v, step, grp(20), 1e-3)
    elif name == "eager64":
        eng.train_steps(params, grads, m, v, step, grp(64), 1e-3)
    elif name == "graph20":
        eng.train_steps(params, grads, m, v, step, grp(5), 1e-3)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                eng.train_steps(params, grads, m, v, step, grp(20), 1e-3)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        print("captured", flush=True)
        g.replay(); torch.cuda.synchronize()
        print("replayed once", flush=True)
        g.replay(); g.replay()
    elif name == "profile20":
        eng.train_steps(params, grads, m, v, step, grp(5), 1e-3)
        torch.cuda.synchronize()
        eng.profile_begin(max_records=64)
        eng.train_steps(params, grads, m, v, step, grp(20), 1e-3)
        torch.cuda.synchronize()
        print(eng.profile_report())
    elif name == "small16":
        eng.train_steps(params, grads, m, v, step, grp(1), 1e-3)
    elif name == "small1000x7":
        eng.train_steps(params, grads, m, v, step, grp(7), 1e-3)
    torch.cuda.synchronize()
    gave_up = eng.train_steps_gave_up()
    print(f"{name}: ok step={int(step.item())} loss={float(grads[eng.P]):.5f} gave_up={gave_up} status={eng.train_steps_status_word:#x} "
          f"finite={bool(torch.isfinite(params).all())}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for s in SCEN:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=240)
            tail = [l for l in (r.stdout + r.stderr).strip().splitlines() if "amdgpu.ids" not in l][-3:]
            print(f"[{s}] rc={r.returncode} :: " + " | ".join(tail), flush=True)
