import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_training_amd.engine import Engine
from vae_training_amd.datasets import LinearGaussianDataset
ds = LinearGaussianDataset(2, 3, 3, 9)
for B in (100, 65536):
    eng = Engine(B, 12, 20)
    k, A, dd, did, pad, var = ds.device_spec()
    out = eng.make_batch(k, A, dd, did, pad, var, B, 1, step=1)
    for name, o in (("x+z", out), ("z only", (None, out[1], out[2])), ("x only", (out[0], None, None))):
        eng.profile_begin(1024)
        for i in range(100):
            eng.make_batch(k, A, dd, did, pad, var, B, 1, step=i, out=o)
        torch.cuda.synchronize()
        r = eng.profile_report()["make_batch"]
        print(B, name, round(r["total_ms"] / r["count"] * 1e3, 2), "us")
