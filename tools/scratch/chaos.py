import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_conv as T
from oracle import conv_vae_oracle as CO
size, widths, L, B, tdv = 64, (32, 64, 128, 256), 32, 2, True
cfg = CO.ConvConfig(size, widths, L, -1.5, tdv)
p, x, z1, z2 = T._conv_problem(cfg, B)
_, e1 = T._bf16_emulation_grads(cfg, p, x, z1, z2)
rng = np.random.default_rng(1)
for amp in (1e-7, 1e-6):
    p2 = {k: v * (1 + amp * rng.standard_normal(v.shape)) if k == "Encoder/Conv0/bias" or k == "Encoder/Conv0/kernel" else v for k, v in p.items()}
    _, e2 = T._bf16_emulation_grads(cfg, p2, x, z1, z2)
    print(amp, " ".join(f"{np.max(np.abs(e1[n]-e2[n]))/np.max(np.abs(e1[n])):.1e}" for n in ["Encoder/Conv0/kernel", "Encoder/Conv1/kernel", "Decoder/FC/kernel", "Decoder/ConvT2/kernel", "Decoder/ConvT3/kernel"]))
    _, e3 = T._bf16_emulation_grads(cfg, p2, x, z1, z2, (False,)*5)
    _, e4 = T._bf16_emulation_grads(cfg, p, x, z1, z2, (False,)*5)
    print(" all-rounded:", " ".join(f"{np.max(np.abs(e3[n]-e4[n]))/np.max(np.abs(e4[n])):.1e}" for n in ["Encoder/Conv0/kernel", "Encoder/Conv1/kernel", "Decoder/FC/kernel", "Decoder/ConvT2/kernel", "Decoder/ConvT3/kernel"]))
