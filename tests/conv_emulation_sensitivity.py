"""How sharp can an END-TO-END comparison of the convolutional VAE with its bf16-operand emulation be?  (Not a test: run by hand,
`python tests/conv_emulation_sensitivity.py`, CPU only, about a minute.)

The emulation of tests/test_gpu_conv.py (_bf16_emulation_grads) is run twice at BASELINE config 5's widths, the second time with
the first convolution's kernel and bias multiplied by (1 + 1e-7 * noise) -- the size of a float32 accumulation-order difference.
Output of the run recorded in profiles/r02_conv_emulation_sensitivity.txt: the emulation's OWN gradients move by 1e-2 .. 5e-2 of
their max-abs, because a 1e-7 change of an activation flips its bf16 rounding in the next layer with probability ~1e-7 / 2^-9, a
flip is a 4e-3 relative change, and the flips multiply from layer to layer.  Hence test_conv_vae_loss_and_every_gradient_leaf checks
every convolution call on the inputs it actually received (2e-6) and keeps only a loose end-to-end bound."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import test_gpu_conv as T                               # noqa: E402
from oracle import conv_vae_oracle as CO                # noqa: E402

size, widths, L, B, tdv = 64, (32, 64, 128, 256), 32, 2, True
cfg = CO.ConvConfig(size, widths, L, -1.5, tdv)
p, x, z1, z2 = T._conv_problem(cfg, B)
_, e1 = T._bf16_emulation_grads(cfg, p, x, z1, z2)
rng = np.random.default_rng(1)
names = ["Encoder/Conv0/kernel", "Encoder/Conv1/kernel", "Decoder/FC/kernel", "Decoder/ConvT2/kernel", "Decoder/ConvT3/kernel"]
print("perturbation of Encoder/Conv0   " + "  ".join(n.split("/")[1] for n in names))
for amp in (1e-7, 1e-6):
    p2 = {k: v * (1 + amp * rng.standard_normal(v.shape)) if k.startswith("Encoder/Conv0/") else v for k, v in p.items()}
    _, e2 = T._bf16_emulation_grads(cfg, p2, x, z1, z2)
    print(f"{amp:8.0e}                       " + "  ".join(f"{np.max(np.abs(e1[n] - e2[n])) / np.max(np.abs(e1[n])):.1e}" for n in names))
