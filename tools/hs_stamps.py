"""Phase shares of the bf16-storage forward GEMM's main loop from the -DVAEK_HS_STAMPS build (tools/hs_stamps.sh)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import WORKLOADS, data_dim, init_params_flat, make_batches  # noqa: E402
from vae_training_amd.engine import Engine  # noqa: E402

B = 65536
w = WORKLOADS["C3"]
eng = Engine(B, data_dim(w), w["L"], w["enc"], w["dec"], w["eps"], w["tdv"], False, dtype="bf16")
lib = eng.lib
params = init_params_flat(eng, 0)
x, z1, z2 = make_batches(w, B, eng.device, 1, seed=5)[0]
for nt in [int(a) for a in sys.argv[1:]] or [0]:
    lib.vaek_debug_hs_variant(nt, -1, None, None)
    buf = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device="cuda")
    assert lib.vaek_debug_hs_stamps(C.c_void_p(buf.data_ptr())) == 0
    for _ in range(2):
        out = eng.loss_eval(params, x, z1, z2)           # forward only: the LAST hs_nt launch (a forward GEMM) leaves its stamps
    torch.cuda.synchronize()
    buf.zero_()
    out = eng.loss_eval(params, x, z1, z2)
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(-1, 8)
    nw = {3: (B // 256) * 4 * 8, 4: (B // 256) * 4 * 8, 10: (B // 256) * 4 * 8, 7: (B // 256) * 2 * 8, 8: (B // 256) * 2 * 8,
          9: (B // 256) * 2 * 8}.get(nt, (B // 128) * 4 * 4)
    a, ends = raw[:nw], raw[nw:2 * nw, 0]
    tot = a[:, 5].astype(np.float64)
    names = ["prologue issue", "vmcnt wait", "barrier", "stage issue", "multiply"]
    span = ends.max() - a[:, 6].min()
    print(f"nt variant {nt}: {nw} waves; kernel span {span} ticks; per wave: entry->loop end median {np.median(a[:, 7] - a[:, 6]):.0f}, "
          f"main loop {np.median(tot):.0f}, epilogue (loop end -> stores done) {np.median(ends - a[:, 7]):.0f}")
    # residency: how many waves are alive at the kernel's midpoint
    mid = a[:, 6].min() + span // 2
    print(f"   waves alive at the midpoint: {int(((a[:, 6] <= mid) & (ends >= mid)).sum())}; first entry -> last entry {a[:, 6].max() - a[:, 6].min()} ticks; "
          f"average waves alive {float((ends - a[:, 6]).sum()) / span:.0f}")
    order = np.argsort(a[:, 6])
    ent = (a[order, 6] - a[:, 6].min())
    print("   entry time percentiles (ticks):", [int(np.percentile(ent, q)) for q in (1, 10, 25, 50, 75, 90, 99)])
    for i, n in enumerate(names):
        print(f"   {n:15s} median {np.median(a[:, i]):8.0f}  share {np.median(a[:, i] / tot) * 100:5.1f} %")
