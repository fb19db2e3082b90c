"""Launch floor of the box: per-kernel duration (kernel timestamps) and launch-to-launch interval inside a hipGraph for an
empty kernel and for a kernel with one dependent pair of loads, at 1, 9 and 256 workgroups."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from vae_training_amd import _lib
from vae_training_amd.engine import Engine
eng = Engine(256, 12, 20)
p = torch.zeros(128, dtype=torch.int32, device="cuda"); out = torch.zeros(4096, dtype=torch.int32, device="cuda")
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def launch(kind, blocks, n):
    _lib.check(eng.lib.vaek_microbench_launch(eng.h, kind, blocks, n, C.c_void_p(p.data_ptr()), C.c_void_p(out.data_ptr()), st()))
for kind in (0, 1):
    for blocks in (1, 9, 256, 2048):
        launch(kind, blocks, 10); torch.cuda.synchronize()
        eng.profile_begin(512); launch(kind, blocks, 200); torch.cuda.synchronize()
        r = list(eng.profile_report().values())[0]
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                launch(kind, blocks, 100)
        torch.cuda.current_stream().wait_stream(side)
        g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"kind {kind} blocks {blocks:5d}: kernel {r['total_ms'] / r['count'] * 1e3:6.2f} us   graph interval {dt / 2000 * 1e6:6.2f} us")
