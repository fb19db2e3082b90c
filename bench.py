#!/usr/bin/env python3
"""bench.py -- ELBO train-step samples/sec (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one VAE.train_step (networks.py:87-101: forward, ELBO, backward, Adam) over one batch of
synthetic linear-padding data already resident in HBM.  Default workload "M" = the configuration
the metric is quoted on: linear_gaussian dd=3 pad=9 (D=12), latent 20, linear encoder/decoder,
-tdv, epsilon=-1, lr=1e-3 (seed_linpadding_expts.sh:1) at batch 65 536 PER GPU (weak scaling;
--scaling strong divides a global 65 536 over the ranks instead).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (dataset, dd, did, pad, D, L, enc_hidden, dec_hidden, eps, tdv, lr, default batch)
    "M": dict(dataset="linear_gaussian", dd=3, did=3, pad=9, L=20, enc=(), dec=(), eps=-1.0, tdv=True, lr=1e-3, batch=65536,
              desc="linear-padding dd3 pad9 (D=12) L=20 linear enc/dec -tdv eps=-1 lr=1e-3 (seed_linpadding_expts.sh:1)"),
    "M20": dict(dataset="linear_gaussian", dd=3, did=3, pad=17, L=20, enc=(), dec=(), eps=-1.0, tdv=True, lr=1e-3, batch=65536,
                desc="linear-padding dd3 pad17 (D=20) L=20 linear (seed_linpadding_expts.sh:2)"),
    "C2": dict(dataset="sigmoid", dd=3, did=3, pad=3, L=6, enc=(256,), dec=(256,), eps=-3.0, tdv=True, lr=1e-4, batch=8192,
               desc="sigmoid dd3 pad3 (D=7) L=6 width-256 MLP, two decoders"),
    "C3": dict(dataset="sphere", dd=3, did=3, pad=3, L=6, enc=(512, 512, 512), dec=(512, 512, 512), eps=-3.0, tdv=True,
               lr=1e-4, batch=65536, desc="sphere dd3 pad3 (D=6) L=6 512|512|512 MLP"),
    "C4": dict(dataset="linear_gaussian", dd=3, did=3, pad=4093, L=20, enc=(), dec=(), eps=-1.0, tdv=True, lr=1e-3,
               batch=32768, desc="linear-padding ambient D=4096 L=20 linear"),
}


def data_dim(w):
    return w["dd"] + w["pad"] + (1 if w["dataset"] == "sigmoid" else 0)


def n_params(w):
    D, L = data_dim(w), w["L"]
    def net(k, hs, last):
        n = 0
        for h in list(hs) + [last]:
            n += k * h + h
            k = h
        return n
    P = net(D, w["enc"], L) + net(L, w["dec"], D) * (2 if w["dataset"] == "sigmoid" else 1) + L + (1 if w["tdv"] else 0)
    return P


def flops_per_sample(w):
    D, L = data_dim(w), w["L"]
    def pw(k, hs, last):
        n, first = 0, None
        for h in list(hs) + [last]:
            if first is None:
                first = k * h
            n += k * h
            k = h
        return n, first
    e, e1 = pw(D, w["enc"], L)
    d, _ = pw(L, w["dec"], D)
    nd = 2 if w["dataset"] == "sigmoid" else 1
    return 2 * (e + nd * d) * 2 + 2 * (e - e1 + nd * d)     # fwd + dW + dX (no dX for the encoder input)


def init_params_flat(eng, seed=0):
    """lecun-normal kernels (truncated normal / sqrt(fan_in)), zero biases, epsilon_p = epsilon = 1."""
    g = torch.Generator().manual_seed(seed)
    flat = torch.zeros(eng.P, dtype=torch.float32)
    for name, (off, shape) in eng.leaves.items():
        n = int(np.prod(shape))
        if name.endswith("kernel"):
            w = torch.empty(shape, dtype=torch.float32)
            torch.nn.init.trunc_normal_(w, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=g)
            flat[off:off + n] = (w * (math.sqrt(1.0 / shape[0]) / 0.87962566103423978)).reshape(-1)
        elif name.startswith("epsilon"):
            flat[off:off + n] = 1.0
    return flat.to(eng.device)


def make_batches(w, B, device, nbuf, seed):
    """Synthetic inputs restating datasets.py (linear :183-195, sigmoid :240-249, sphere :75-84) plus
    z ~ N(0,1)^(L+D) (model.py:227), generated on the device; `nbuf` rotating batches."""
    g = torch.Generator(device=device).manual_seed(seed)
    gc = torch.Generator().manual_seed(2)                   # dataset seed (-ds 2)
    D, L, dd = data_dim(w), w["L"], w["dd"]
    A = torch.randn(dd, w["did"] if w["dataset"] == "linear_gaussian" else 1, generator=gc).to(device)
    out = []
    for _ in range(nbuf):
        x = torch.zeros(B, D, dtype=torch.float32, device=device)
        if w["dataset"] == "linear_gaussian":
            x[:, :dd] = torch.randn(B, w["did"], device=device, generator=g) @ A.T
        elif w["dataset"] == "sigmoid":
            z = torch.randn(B, dd, device=device, generator=g)
            x[:, :dd] = z
            x[:, dd] = torch.sigmoid(z @ A).squeeze(1)
        else:
            s = torch.randn(B, dd, device=device, generator=g)
            x[:, :dd] = s / s.norm(dim=1, keepdim=True)
        z = torch.randn(B, L + D, device=device, generator=g)
        out.append((x.contiguous(), z[:, :L].contiguous(), z[:, L:].contiguous()))
    return out


def usable_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box
    hands one job a share of a many-core host; oversubscribing it makes an OpenMP loop crawl)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(w, B, seconds):
    """The same train step on the host cores: the C + OpenMP port (oracle/elbo_ref.c, rebuilt here with
    -march=native, all cores), plus the torch float32 restatement (oracle/elbo_torch.py) for reference --
    "CPU restatement, not JAX" (the JAX reference cannot run here).  Checker code,
    timed only in this leg."""
    from oracle import elbo_oracle as O
    from oracle import elbo_ref as R
    from oracle import elbo_torch as T
    D, L = data_dim(w), w["L"]
    cfg = O.Config(D, L, w["enc"], w["dec"], w["eps"], w["tdv"], w["dataset"])
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, D)).astype(np.float32); z1 = rng.standard_normal((B, L)).astype(np.float32)
    z2 = rng.standard_normal((B, D)).astype(np.float32)
    p0 = O.init_params(cfg, seed=0)
    # C + OpenMP port
    lib = R.load(native=True)
    c = R.make_cfg(D, L, w["enc"], w["dec"], w["eps"], w["tdv"], w["dataset"] == "sigmoid")
    P = cfg.n_params()
    params = np.ascontiguousarray(O.flatten(cfg, p0), dtype=np.float32); m = np.zeros(P, np.float32); v = np.zeros(P, np.float32)
    cores = usable_cores()
    torch.set_num_threads(cores)
    for t in (1, 2):
        R.step(lib, c, params, m, v, t, x, z1, z2, w["lr"], nthreads=cores)
    n, t0 = 0, time.perf_counter()
    while True:
        R.step(lib, c, params, m, v, 3 + n, x, z1, z2, w["lr"], nthreads=cores)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds * 0.7 or n >= 20000:
            break
    out = {"value": B * n / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": f"{n} train steps at batch {B} of the same workload, C + OpenMP float32 port (oracle/elbo_ref.c, "
                     f"-O3 -march=native, {cores} threads), {dt:.1f} s wall"}
    # torch restatement, a few steps only
    model = T.TorchVAE(cfg, p0, dtype=torch.float32)
    opt = T.make_adam(model, w["lr"])
    xt, z1t, z2t = torch.from_numpy(x), torch.from_numpy(z1), torch.from_numpy(z2)
    T.train_step(model, opt, xt, z1t, z2t)
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds * 0.3 and k < 200:
        T.train_step(model, opt, xt, z1t, z2t)
        k += 1
    out["torch_restatement_samples_per_s"] = B * max(k, 1) / (time.perf_counter() - t0)
    out["torch_threads"] = torch.get_num_threads()
    return out


def cpu_baseline_c1(seconds_each=1.5):
    """BASELINE config 1 (the reference's own CPU-runnable case, seed_linpadding_expts.sh:1 at run.py:14's batch sizes) on the
    host cores: the C port at B = 128 (BASELINE.json) and 100 (the script default), latent 2 (BASELINE.json) and 20 (the
    script).  One thread: at these sizes a step is ~100 kflop and OpenMP's fork/join costs more than it."""
    from oracle import elbo_oracle as O
    from oracle import elbo_ref as R
    lib = R.load(native=True)
    legs = []
    for B in (128, 100):
        for L in (2, 20):
            cfg = O.Config(12, L, (), (), -1.0, True, "linear_gaussian")
            rng = np.random.default_rng(1)
            x = rng.standard_normal((B, 12)).astype(np.float32); z1 = rng.standard_normal((B, L)).astype(np.float32)
            z2 = rng.standard_normal((B, 12)).astype(np.float32)
            c = R.make_cfg(12, L, (), (), -1.0, True, False)
            P = cfg.n_params()
            params = np.ascontiguousarray(O.flatten(cfg, O.init_params(cfg, seed=0)), dtype=np.float32)
            m = np.zeros(P, np.float32); v = np.zeros(P, np.float32)
            for t in (1, 2):
                R.step(lib, c, params, m, v, t, x, z1, z2, 1e-3, nthreads=1)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds_each and n < 2000000:
                for _ in range(200):
                    R.step(lib, c, params, m, v, 3 + n, x, z1, z2, 1e-3, nthreads=1)
                    n += 1
            dt = time.perf_counter() - t0
            legs.append({"batch": B, "latent_dim": L, "value": B * n / dt, "unit": "samples/s", "us_per_step": dt / n * 1e6,
                         "cores": 1, "kind": "port", "sample": f"{n} train steps, C float32 port, 1 thread, {dt:.1f} s wall "
                                                               "(includes ~1 us of ctypes call overhead per step)"})
    return legs


def bench_conv(args):
    """Workload C5 (BASELINE config 5; no reference counterpart): the convolutional VAE's train step assembled from the
    library's blocks (vae_training_amd/conv_vae.py).  One GPU: the step captured into a hipGraph.  N GPUs (torch.distributed.run):
    data parallel, weak scaling -- every rank its own batch, one RCCL all-reduce of the flat gradient per step, eager.  Its own
    line: same metric and unit."""
    import torch

    from vae_training_amd.conv_vae import ConvVAE
    from vae_training_amd.parallel import GradExchange
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    S, widths, L = 64, (32, 64, 128, 256), 32
    B = args.batch or 4096                                   # config 5 names 32 768 over 8 GPUs
    steps, warm = min(args.steps, 20), min(max(args.warmup, 2), 5)
    net = ConvVAE(B, S, widths, L, -3.0, True, device=local_rank, world=world)
    g = torch.Generator(device="cpu").manual_seed(0)
    params, grads, m, v = net.new_flat(), net.new_flat(), net.new_flat(), net.new_flat()
    for name, (off, shape) in net.leaves.items():
        if name.endswith("kernel"):
            fan = 16 * shape[2] if "Conv" in name and "ConvT" not in name else (4 * shape[3] if "ConvT" in name else shape[0])
            net.view(params, name).copy_((torch.randn(*shape, generator=g) / math.sqrt(fan)).to(net.device))
        elif name == "epsilon":
            net.view(params, name).fill_(1.0)
    g = torch.Generator(device="cpu").manual_seed(1 + rank)  # every rank its own shard of the global batch
    x = torch.rand(B, S, S, 1, generator=g).to(net.device)
    z1 = torch.randn(B, L, generator=g).to(net.device)
    z2 = torch.randn(B, S, S, 1, generator=g).to(net.device)
    step = torch.zeros(1, dtype=torch.int32, device=net.device)
    if world == 1 and not args.no_graph:
        # the step as one hipGraph (nothing in it touches the host): ~95 launches whose 4-6 us gaps are 8 % of an eager step
        run, out4 = net.capture(params, grads, m, v, step, x, z1, z2, 1e-4, warmup=warm)
        run()
    else:
        exch = GradExchange(net.eng, dist, mode="rccl") if world > 1 else None
        box = {}

        def run():
            box["out4"] = net.train_step(params, grads, m, v, step, x, z1, z2, 1e-4, all_reduce=exch.all_reduce if exch else None)
        for _ in range(warm):
            run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if box_used := (world > 1 or args.no_graph):
        out4 = box["out4"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)             # the slowest rank's clock
        elapsed = float(t[0])
        chk = torch.stack([params.double().sum(), params.double().abs().sum()]).cpu()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "replicas diverged"
    loss = float(out4[0])
    assert math.isfinite(loss)
    if rank != 0:
        return
    chans = [1, *widths]
    macs = sum((S >> (i + 1)) ** 2 * chans[i + 1] * 16 * chans[i] for i in range(4))       # one conv stack, forward, per sample
    flops = 3 * 2 * (2 * macs + 2 * net.bott * L)                                           # both stacks + the two Dense; fwd + 2 x bwd
    tf = world * B * flops / (elapsed / steps) / 1e12
    out = {"metric": "ELBO train-step samples/sec", "value": world * B * steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": steps,
           "warmup": warm, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "C5: conv VAE 64x64x1, 4x4/s2 convs 32|64|128|256, L=32 (BASELINE config 5; no reference counterpart)",
                      "batch_per_gpu": B, "global_batch": B * world, "params": net.n_params, "parallelism": f"dp{world}",
                      "path": "blocks, launch per layer, one hipGraph per step" if world == 1 and not args.no_graph else
                              "blocks, launch per layer, eager" if world == 1 else
                              "blocks, launch per layer, eager + one all-reduce of the flat gradient per step",
                      "grad_exchange": "none" if world == 1 else "rccl", "final_loss": loss},
           "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0 * world, "unit": "TFLOP/s", "frac": tf / (2500.0 * world), "traffic": None,
                        "kernel": "all launches of a step", "kernel_avg_us": 0.0, "algorithmic_per_launch": world * B * flops,
                        "step_level": {"achieved": tf, "frac": tf / (2500.0 * world)}, "step_kernels_us": {}},
           "cpu_baseline": None}
    print(json.dumps(out), flush=True)


def main():
    if "--workload" in sys.argv and sys.argv[sys.argv.index("--workload") + 1] == "C5":
        ap = argparse.ArgumentParser()
        ap.add_argument("--workload"); ap.add_argument("--gpus", type=int, default=1); ap.add_argument("--steps", type=int, default=10)
        ap.add_argument("--warmup", type=int, default=3); ap.add_argument("--batch", type=int, default=0)
        ap.add_argument("--no-cpu-baseline", action="store_true"); ap.add_argument("--rehearse-one-gpu", action="store_true")
        ap.add_argument("--no-graph", action="store_true", help="eager launches on one GPU too (counter profiling)")
        return bench_conv(ap.parse_args())
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=960)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--workload", default="M", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (weak) / global batch (strong); 0 = workload default")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--comm", default="auto", choices=["auto", "rccl", "p2p"])
    ap.add_argument("--nbuf", type=int, default=0,
                    help="rotating synthetic input batches resident in HBM; 0 = auto: enough that one rotation (>= 600 MB, at most 48 "
                         "batches) cannot stay in the 256 MiB Infinity Cache")
    ap.add_argument("--keep-mall", action="store_true",
                    help="do not sweep the Infinity Cache before the timed region (default: a 1 GiB fill evicts the inputs the warm-up "
                         "steps left there, so the timed steps read them from HBM)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--measure-peaks", action="store_true",
                    help="also run the library's micro-benchmarks (copy, MFMA loops, empty launches) and report them as roofline.peak_measured; "
                         "they reach only ~0.55 of the HBM and ~0.66 of the f32 MFMA peak, so `peak` stays the guide's figure")
    ap.add_argument("--force-generic", action="store_true", help="layer-by-layer kernels even where a fused path exists")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a ONE-GPU box: every rank uses cuda:0 and the process group is gloo")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="Dense GEMM arithmetic (bf16: wide layers on bf16 MFMA)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="linear VAEs on one GPU: the step-by-step kernels (vaek_train_step) instead of vaek_train_steps, the N-step "
                         "entry point whose launches overlap the pass over batch n + 2 with the Adam update of batch n")
    ap.add_argument("--graph", type=int, default=-1,
                    help="steps captured per hipGraph (0 = eager launches; -1 = auto: the largest even size <= 200 dividing --steps)")
    ap.add_argument("--only-main", action="store_true", help="skip the A/B legs (per-sample path, fresh inputs): profiler passes")
    ap.add_argument("--repeat", type=int, default=20,
                    help="how many times the timed region of exactly --steps steps is run (each time behind its own Infinity Cache sweep, "
                         "barrier and synchronize); `value` is the MEDIAN region, min / max ride along")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph", "direct"],
                    help="vaek_train_steps only: replay a hipGraph of the K steps, or call the library directly with pre-marshalled "
                         "arguments (one persistent launch per 64 steps either way; auto = direct: a graph replay costs the host more "
                         "than the one or two launches it replaces)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no MI355X visible; the HIP path has no CPU fallback")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from vae_training_amd.engine import Engine
    from vae_training_amd.parallel import GradExchange

    w = WORKLOADS[args.workload]
    B0 = args.batch or w["batch"]
    if args.scaling == "weak":
        B_local, B_global = B0, B0 * world
    else:
        assert B0 % world == 0, "strong scaling needs batch divisible by the rank count"
        B_local, B_global = B0 // world, B0
    D, L = data_dim(w), w["L"]
    eng = Engine(B_local, D, L, w["enc"], w["dec"], w["eps"], w["tdv"], w["dataset"] == "sigmoid", device=local_rank,
                 world=world, rank=rank, global_batch=B_global, force_generic=args.force_generic, dtype=args.dtype)
    params = init_params_flat(eng, seed=0)
    grads = eng.new_flat(eng.grad_len)
    m, v = eng.new_flat(), eng.new_flat()
    step_dev = torch.zeros(1, dtype=torch.int32, device=device)
    batch_bytes = B_local * 4 * (2 * D + L)
    nbuf = args.nbuf if args.nbuf > 0 else int(min(48, max(4, math.ceil(600e6 / batch_bytes))))
    batches = make_batches(w, B_local, device, nbuf, seed=1000 + rank)
    mall_sweep = None if args.keep_mall else torch.zeros(1 << 28, dtype=torch.float32, device=device)     # 1 GiB
    sweep_sink = torch.zeros((), dtype=torch.float32, device=device)
    exch = GradExchange(eng, dist, mode=args.comm) if world > 1 else None
    lr = w["lr"]

    # data parallel over RCCL (no P2P communicator): a linear VAE's step goes through its second-moment matrix too, summed by the
    # collective between vaek_train_steps_moments and vaek_train_steps_update (launch per step)
    moments_rccl = exch is not None and not exch.in_library and eng.moment_len() > 0 and not args.no_pipeline

    def one_step(i):
        x, z1, z2 = batches[i % len(batches)]
        if exch is None or exch.in_library:
            eng.train_step(params, grads, m, v, step_dev, x, z1, z2, lr)
        elif moments_rccl:
            exch.moments_step(params, grads, m, v, step_dev, x, z1, z2, lr)
        elif eng.fused:
            eng.grads_only(params, grads, step_dev, x, z1, z2)
            exch.all_reduce(grads)
            eng.apply(params, grads, m, v, step_dev, lr)
        else:       # layer-by-layer model: per-layer buckets all-reduced on a side stream under the rest of the backward
            exch.overlapped_grads(params, grads, step_dev, x, z1, z2)
            eng.apply(params, grads, m, v, step_dev, lr)

    # Linear VAE on one GPU (the metric's workload): K consecutive steps go through vaek_train_steps -- every step still
    # reads its own batch once, computes loss and gradients at the CURRENT parameters and applies Adam, but evaluated through
    # the batch's second-moment matrix, so that launch n streams batch n while it sums batch n - 1 and updates with batch n - 2
    # (csrc/linear_moments.hip).  Everything else: one vaek_train_step per step.
    # Data parallel: the moment matrix is additive over ranks and is exchanged inside the persistent launch over the P2P
    # communicator (same kernel at every N); without that communicator (RCCL transport) N > 1 takes the per-sample step.
    use_pipe = (exch is None or exch.in_library) and not args.no_pipeline and eng.supports_train_steps()

    launched = [0]              # train steps issued on the main entry point so far (profiler tools divide counters by it)

    def run_group(i0, n):
        """n consecutive train steps on batches i0, i0 + 1, ... (mod the rotation)."""
        if n <= 0:
            return
        launched[0] += n
        if use_pipe:
            eng.train_steps(params, grads, m, v, step_dev, [batches[(i0 + k) % len(batches)] for k in range(n)], lr)
        else:
            for k in range(n):
                one_step(i0 + k)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The steady-state loop is launch-bound (two ~5-10 us kernels per step), so G consecutive steps
    # are captured once into a hipGraph and replayed; every step inside still runs the full forward,
    # backward and Adam on its own (rotating) batch, and K timed steps are exactly K train steps.
    gsteps = args.graph
    if gsteps < 0:
        if exch is not None and not exch.in_library:
            gsteps = 0
        else:
            # a graph launch costs ~6 us of its own (13.40 us/step at 20 steps per graph, 13.15 at 200): capture the largest
            # size <= 200 that divides K (or leaves the fewest eager steps over); step i of a replay reads batch i % nbuf
            cands = [g for g in range(200, 0, -2) if g <= max(args.steps, 2)]      # even: the fresh-input leg alternates two buffers
            gsteps = min(cands, key=lambda g: (args.steps % g, -g))          # fewest eager left-over steps, then the largest
    graph = None
    # vaek_train_steps needs no graph: ONE library call covers all K steps (a persistent launch per 64 of them)
    use_plan = use_pipe and args.launch != "graph" and 0 < args.steps <= 4096
    n_warm_eager = max(args.warmup, 3)
    run_group(0, n_warm_eager)
    if use_pipe and dist is not None:
        # the in-launch exchange has never met this topology before: all ranks agree that no bounded wait expired, or all fall
        # back to the per-sample step with the gradient exchange
        torch.cuda.synchronize()
        bad = torch.tensor([1.0 if eng.train_steps_gave_up() else 0.0], device="cpu" if args.rehearse_one_gpu else device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if float(bad.item()) != 0.0:
            use_pipe = False
            run_group(0, n_warm_eager)
    if use_plan and not use_pipe:           # (the warm-up found an exchange time-out: back on the per-sample step)
        use_plan = False
    if gsteps > 0 and not use_plan:
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                run_group(0, gsteps)
        torch.cuda.current_stream().wait_stream(side)
        fence()                             # ranks leave capture at different times: line them up first
        launched[0] -= gsteps               # (capture recorded, did not run, that group)
        graph.replay()                      # one untimed replay (graph upload)
        launched[0] += gsteps

    plan = None
    if use_plan:
        gsteps = args.steps
        plan = eng.plan_train_steps(params, grads, m, v, step_dev, [batches[k % len(batches)] for k in range(gsteps)], lr)
        plan()
        launched[0] += gsteps

    def run_steps(n):
        done = 0
        if plan is not None:
            while n - done >= gsteps:
                plan()
                done += gsteps
                launched[0] += gsteps
        elif graph is not None:
            while n - done >= gsteps:
                graph.replay()
                done += gsteps
                launched[0] += gsteps
        run_group(0, n - done)

    def sweep():
        """Evict whatever the untimed steps left in the 256 MiB Infinity Cache, so that the timed steps read their inputs from
        HBM (a short run would otherwise re-read the very batches its warm-up just touched), then bring everything ELSE back to
        its steady state -- code, parameters, Adam moments, workspace -- with a few untimed steps on batches from the far end of
        the rotation, which a short timed run does not reach."""
        if mall_sweep is None:
            return
        # a READ sweep: reading 1 GiB leaves the cache full of clean lines.  (Round 2 swept with fill_(): the 256 MiB of dirty lines
        # it left were written back to HBM while the timed steps streamed their inputs -- 35 us of a 20-step region, an artefact of
        # the sweep, not of the steps.)
        sweep_sink.copy_(mall_sweep.sum())
        run_group(len(batches) - 4, 4)

    regions = []
    for _ in range(max(1, args.repeat)):
        sweep()
        fence()
        t0 = time.perf_counter()
        run_steps(args.steps)
        fence()
        regions.append(time.perf_counter() - t0)
    if dist is not None:
        t = torch.tensor(regions, dtype=torch.float64, device="cpu" if args.rehearse_one_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # per region: the slowest rank's clock
        regions = [float(x) for x in t.tolist()]
    elapsed = float(np.median(regions))
    loss = float(grads[eng.P].item())
    assert math.isfinite(loss), "train step produced a non-finite loss"
    if use_pipe:
        assert not eng.train_steps_gave_up(), (f"a bounded in-launch wait of vaek_train_steps expired (status "
                                               f"{eng.train_steps_status_word:#x}, {elapsed:.3f} s for {args.steps} steps): results invalid")
    if exch is not None and exch.in_library:
        assert not exch.timed_out(), "p2p gradient exchange gave up waiting for a peer"

    # ---- A/B (never `value`): the same workload through the drop-in per-sample step, vaek_train_step (two launches per step:
    # the fused forward/backward chain + finalize/Adam), captured and timed the same way -- what `value` was in round 1
    per_sample = None
    if use_pipe and world == 1 and not args.only_main:
        gs2 = max(2, min(gsteps if gsteps > 0 else 20, 192))
        p2, g2, m2, v2 = params.clone(), eng.new_flat(eng.grad_len), m.clone(), v.clone()
        s2 = step_dev.clone()

        def ps_steps(n):
            for k in range(n):
                x, z1, z2 = batches[k % len(batches)]
                eng.train_step(p2, g2, m2, v2, s2, x, z1, z2, lr)
        ps_steps(3)
        torch.cuda.synchronize()
        side2 = torch.cuda.Stream()
        side2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side2):
            graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph2, stream=side2):
                ps_steps(gs2)
        torch.cuda.current_stream().wait_stream(side2)
        graph2.replay()
        if mall_sweep is not None:
            sweep_sink.copy_(mall_sweep.sum())
            ps_steps(4)
        torch.cuda.synchronize()
        reps = max(1, min(args.steps, 400) // gs2)
        t1 = time.perf_counter()
        for _ in range(reps):
            graph2.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        per_sample = {"entry_point": "vaek_train_step (per-sample forward/backward chain + finalize: two launches per step)",
                      "samples_per_s": B_local * reps * gs2 / dt, "us_per_step": dt / (reps * gs2) * 1e6, "steps": reps * gs2,
                      "launch": f"hipGraph x{gs2} steps",
                      "note": "A/B leg: at most 400 steps, started right behind an Infinity Cache sweep -- colder than a long run of "
                              "the same kernels (round 1's driver figure for them, 13.3 us, came from the main timed region)"}

    # ---- roofline leg: the same steps again with every library launch bracketed by its own begin/end timestamps ----
    roofline = None
    if not args.no_roofline:
        rsteps = min(args.steps, 192)                       # kernel-timestamp leg (one event pair per launch)
        sweep()
        eng.profile_begin(max_records=rsteps * 64)
        run_group(0, rsteps)
        torch.cuda.synchronize()
        rep = eng.profile_report()
        if rep:
            P = eng.P
            dom = max(rep, key=lambda k: rep[k]["total_ms"])
            dom_avg_s = rep[dom]["total_ms"] / rep[dom]["count"] * 1e-3
            step_kernels_s = sum(r["total_ms"] for r in rep.values()) / rsteps * 1e-3      # all launches of one step
            step_wall_s = elapsed / args.steps
            mfma_bound = flops_per_sample(w) / (4.0 * (2 * D + L)) > 300.0             # above the ridge: C3
            if mfma_bound:
                # layer-by-layer path: the Dense kernels run once per layer; the step's flops are priced against the time
                # of all Dense launches of a step
                alg = B_local * flops_per_sample(w)
                if eng.fused:       # whole-network kernel (one-hidden-layer MLPs): every flop of the step is in that one launch
                    kernel_s, kernel_name = dom_avg_s, dom
                else:
                    kernel_s = sum(r["total_ms"] for k, r in rep.items() if k.startswith("gemm")) / rsteps * 1e-3
                    kernel_name = "all Dense (gemm_*) launches of a step"
                peak, unit, bound, scale = (2500.0 if args.dtype == "bf16" else 157.3), "TFLOP/s", "mfma", 1e12
                alg_step = alg
            else:
                alg = B_local * 4 * (2 * D + L)              # x, z1, z2 read once; + 32 P per step
                alg_step = alg + 32 * P
                peak, unit, bound, scale = 8000.0, "GB/s", "hbm", 1e9
                if eng.fused:
                    # one kernel reads every algorithmic input byte: price THAT kernel by its own duration.  A persistent launch of
                    # vaek_train_steps covers several steps: its algorithmic bytes are those of all its steps.
                    steps_per_launch = rsteps / rep[dom]["count"] if dom.startswith("lin_moments") else 1.0
                    alg *= steps_per_launch
                    kernel_s, kernel_name = dom_avg_s, dom
                else:
                    # multi-kernel path: the algorithmic bytes are spread over all launches of the step, so the
                    # denominator is their summed duration (one kernel's duration would overstate the rate)
                    alg = alg_step
                    kernel_s, kernel_name = step_kernels_s, "all launches of a step (layer-by-layer path)"
            achieved = alg / kernel_s / scale
            # HBM bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh): a constant read from profiles/, NOT
            # measured in this run -- taken from the newest round's file that lists the dominant kernel under its current name
            # (none: null), and labelled with the file it came from
            traffic, traffic_source = None, None
            for rnd in ("r03", "r02", "r01"):
                tfile = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{args.workload}{'_bf16' if args.dtype == 'bf16' else ''}.json")
                if not (os.path.exists(tfile) and B_local == w["batch"]):
                    continue
                meta = json.load(open(tfile))
                kern = meta["kernels"]
                if eng.fused:
                    if dom not in kern:
                        continue
                    if "traffic_bytes" in kern[dom]:
                        traffic = kern[dom]["traffic_bytes"]
                    elif "traffic_bytes_per_step" in kern[dom]:       # persistent launch: per step x its steps
                        traffic = kern[dom]["traffic_bytes_per_step"] * steps_per_launch
                else:
                    names = " ".join(v.get("rocprof_name", k) for k, v in kern.items())
                    if any(k.split("<")[0] not in names and k not in kern for k in rep if k.startswith(("gemm", "ts_gemm", "hs_", "sk_"))):
                        continue            # a kernel of today's step is missing from that file: its total is stale
                    traffic = sum(v.get("traffic_bytes_per_step", 0) for v in kern.values()) or None
                if traffic is not None:
                    traffic_source = f"profiles/{os.path.basename(tfile)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {meta.get('commit', 'commit not recorded')}; not measured in this run)"
                    break
            roofline = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                        "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel_name, "kernel_avg_us": kernel_s * 1e6,
                        "dominant_kernel": dom, "dominant_kernel_avg_us": dom_avg_s * 1e6,
                        "steps_per_dominant_launch": (rsteps / rep[dom]["count"] if dom.startswith("lin_moments") else 1.0),
                        "dominant_kernel_us_per_step": dom_avg_s * 1e6 / (rsteps / rep[dom]["count"] if dom.startswith("lin_moments") else 1.0),
                        "launches_per_step": sum(r["count"] for r in rep.values()) / rsteps,
                        "algorithmic_per_launch": alg,
                        # the same algorithmic work priced against the whole step's wall time (launch gaps, finalize, Adam
                        # included): what `value` corresponds to
                        "step_level": {"algorithmic_per_step": alg_step, "step_us": step_wall_s * 1e6,
                                       "achieved": alg_step / step_wall_s / scale, "frac": alg_step / step_wall_s / scale / peak},
                        "step_kernels_us": {k: r["total_ms"] / rsteps * 1e3 for k, r in rep.items()},
                        "param_bytes_per_step": 32 * P,
                        "inputs": f"{nbuf} rotating batches = {nbuf * batch_bytes / 1e6:.0f} MB per rotation"
                                  + ("" if args.keep_mall else ", Infinity Cache swept (1 GiB read) before the timed steps")}
            if rank == 0 and args.measure_peaks:       # optional: the library's own copy / MFMA-loop / empty-launch micro-benchmarks
                roofline["peak_measured"] = eng.measure_peaks()
                roofline["peak_measured"]["empty_kernel_launch_interval_us_in_hipGraph"] = eng.measure_launch_floor()

    # ---- auxiliary (never `value`): the same step with FRESH inputs drawn on the device every step, as the
    # reference's loop does (dataset.get_batch + sample_latent, model.py:221 / vae.py:125-128): the Philox draw of
    # batch n+1 rides in the finalize launch of step n (vaek_train_step_gen), all inside the hipGraph
    fresh = None
    kind = {"linear_gaussian": 0, "sigmoid": 1, "sphere": 2}[w["dataset"]]
    if args.only_main:
        pass
    elif world == 1 and use_pipe and eng.supports_train_steps_gen(kind):
        # the loop body of model.py:221-222 as run.py --fast_loop runs it (trainer.GraphLoop): vaek_train_steps_gen, every step's
        # batch drawn inside the persistent launch (same Philox streams as vaek_make_batch), nothing read from HBM, no graph
        A = torch.randn(w["dd"], w["did"], generator=torch.Generator().manual_seed(2)).to(device).contiguous() if kind == 0 else None
        p3, g3, m3, v3, s3 = params.clone(), eng.new_flat(eng.grad_len), m.clone(), v.clone(), step_dev.clone()
        gen = lambda n: eng.train_steps_gen(p3, g3, m3, v3, s3, n, lr, kind, A, w["dd"], w["did"], w["pad"], 0.0, 7)
        gen(64)
        torch.cuda.synchronize()
        times = []
        for _ in range(max(1, min(args.repeat, 10))):
            t1 = time.perf_counter()
            gen(args.steps)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t1)
        dt = float(np.median(times))
        assert not eng.train_steps_gave_up()
        fresh = {"samples_per_s": B_local * args.steps / dt, "us_per_step": dt / args.steps * 1e6, "steps": args.steps,
                 "entry_point": "vaek_train_steps_gen (trainer.GraphLoop, run.py --fast_loop)",
                 "note": "inputs drawn inside the persistent launch each step (Philox work items in the streamer workgroups, tile by "
                         "tile into LDS) instead of pre-resident batches: what the reference's loop does per iteration "
                         "(dataset.get_batch + sample_latent + train_step)"}
    elif world == 1 and graph is not None and w["dataset"] == "linear_gaussian":
        A = torch.randn(w["dd"], w["did"], generator=torch.Generator().manual_seed(2)).to(device).contiguous()
        assert gsteps % 2 == 0
        bufs = [tuple(torch.empty_like(t) for t in batches[0]) for _ in range(2)]
        counter = torch.zeros(2, dtype=torch.int32, device=device)
        eng.make_batch(0, A, w["dd"], w["did"], w["pad"], 0.0, B_local, seed=7, counter=counter, which=0, out=bufs[0])
        fresh_n = [0]

        def fresh_step():       # train on batch n, draw batch n+1 in the finalize launch (vaek_train_step_gen, trainer.GraphLoop)
            n = fresh_n[0]
            eng.train_step_gen(params, grads, m, v, step_dev, bufs[n % 2], lr, 0, A, w["dd"], w["did"], w["pad"], 0.0,
                               bufs[(n + 1) % 2], 7, counter, (n + 1) % 2)
            fresh_n[0] = n + 1
        for _ in range(4):
            fresh_step()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=side):
                for _ in range(gsteps):
                    fresh_step()
        torch.cuda.current_stream().wait_stream(side)
        g2.replay(); torch.cuda.synchronize()
        nrep = max(1, args.steps // gsteps)
        t1 = time.perf_counter()
        for _ in range(nrep):
            g2.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        fresh = {"samples_per_s": B_local * nrep * gsteps / dt, "us_per_step": dt / (nrep * gsteps) * 1e6,
                 "note": "inputs drawn on the device each step by vaek_train_step_gen (Philox work items riding in the launches of "
                         "the step before) instead of pre-resident batches"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w, B_local, args.cpu_seconds)
        if args.workload == "M":
            cpu["c1"] = cpu_baseline_c1()

    if rank == 0:
        value = B_global * args.steps / elapsed
        out = {
            "metric": "ELBO train-step samples/sec", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "steps_launched_total": launched[0],
            "timed_regions": {"count": len(regions), "steps_each": args.steps, "median_ms_per_step": elapsed / args.steps * 1e3,
                              "min_ms_per_step": min(regions) / args.steps * 1e3, "max_ms_per_step": max(regions) / args.steps * 1e3,
                              "note": "value = the median region; each region = exactly --steps train steps between barrier + synchronize"},
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['desc']}", "batch_per_gpu": B_local, "global_batch": B_global,
                       "data_dim": D, "latent_dim": L, "params": eng.P, "path": "fused" if eng.fused else "layer-by-layer",
                       "step_entry_point": ("vaek_train_steps (up to 64 steps per persistent launch: streamers | reducers | updater)" if use_pipe else
                                            "vaek_train_steps_moments + all-reduce of the 12 KB moment matrix + vaek_train_steps_update (launch per step)"
                                            if moments_rccl else "vaek_train_step"),
                       "parallelism": f"dp{world}", "grad_exchange": (exch.mode if exch else "none"),
                       "launch": (f"direct library call x{gsteps} steps (arguments marshalled once)" if plan is not None else
                                  f"hipGraph x{gsteps} steps" if graph is not None else "eager"), "input_batches": nbuf,
                       "final_loss": loss},
            "roofline": roofline, "cpu_baseline": cpu, "per_sample_path": per_sample, "fresh_inputs_each_step": fresh,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
