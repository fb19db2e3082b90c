"""Per-kernel time of vaek_train_step_gen (finalize + next-batch draw in one launch) beside the separate launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_training_amd.engine import Engine
from vae_training_amd.datasets import LinearGaussianDataset
ds = LinearGaussianDataset(2, 3, 3, 9)
k, A, dd, did, pad, var = ds.device_spec()
for B in (100, 65536):
    eng = Engine(B, 12, 20, epsilon=-1.0, tunable_decoder_var=True)
    st = [torch.randn(eng.P, device="cuda") * 0.3, eng.new_flat(eng.grad_len), eng.new_flat(), eng.new_flat(),
          torch.zeros(1, dtype=torch.int32, device="cuda")]
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    bufs = [eng.make_batch(k, A, dd, did, pad, var, B, 1, counter=counter, which=0),
            eng.make_batch(k, A, dd, did, pad, var, B, 1, step=0)]
    for mode in ("separate", "gen"):
        for rep in range(2):
            if rep == 1:
                eng.profile_begin(2048)
            for n in range(200):
                if mode == "gen":
                    eng.train_step_gen(*st, bufs[n % 2], 1e-3, k, A, dd, did, pad, var, bufs[(n + 1) % 2], 1, counter, (n + 1) % 2)
                else:
                    eng.make_batch(k, A, dd, did, pad, var, B, 1, step_dev=st[4], out=bufs[0])
                    eng.train_step(*st, *bufs[0], 1e-3)
            torch.cuda.synchronize()
        r = eng.profile_report()
        print(B, mode, {kk: round(v["total_ms"] / v["count"] * 1e3, 2) for kk, v in r.items()})
