"""Steps/s of the reference-shaped Python loop vs the graph-captured loop at the reference's own
batch size (run.py default 100; seed_linpadding_expts.sh:1 model)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_training_amd.run import get_dataset, parse_arguments
from vae_training_amd.trainer import GraphLoop
from vae_training_amd.vae import VAEModel

def build(B):
    args = parse_arguments(["t", "--dataset", "linear_gaussian", "--padding_dim", "9", "-dd", "3", "-ds", "2"])
    ds = get_dataset("linear_gaussian", 2, 9, B, args)
    return VAEModel(dirname=tempfile.mkdtemp(), num_batches=10, num_epochs=1, batch_size=B, learning_rate=1e-3, layer_sizes="",
                    encoder_layer_sizes="", state_dict=None, data_fn=None, epsilon=-1.0, tqdm=False, dataset=ds,
                    latent_dimension=20, tunable_decoder_var=True, dataset_name="linear_gaussian")

for B in (100, 65536):
    m = build(B)
    for _ in range(20):
        m.train_one_batch(m.dataset.get_batch(B))
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 2000
    for _ in range(n):
        m.train_one_batch(m.dataset.get_batch(B))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B:6d} drop-in python loop : {dt / n * 1e6:8.1f} us/step  {B * n / dt:14.0f} samples/s")
    for pipe in (False, True):
        m = build(B); loop = GraphLoop(m, pipeline=pipe)
        loop.run(200); torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20000
        loop.run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"B={B:6d} hipGraph loop (K7 rng, pipeline={pipe!s:5}): {dt / n * 1e6:8.1f} us/step  {B * n / dt:14.0f} samples/s  "
              f"final loss {float(loop.losses()[-1]):.4f}")
