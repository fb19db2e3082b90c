"""Philox / Box-Muller: the NumPy oracle against Random123's published known answers (CPU), and the
HIP generator (csrc/rng.hip) against that oracle + distributional checks (GPU)."""
import numpy as np
import pytest
import torch

from oracle import philox as P

# Random123 kat_vectors, philox4x32 with 10 rounds: counter[4] key[2] -> expected[4]
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_oracle_known_answers(ctr, key, want):
    assert tuple(int(v) for v in P.philox4x32(ctr, key)) == want


def test_box_muller_oracle_moments():
    ctr = np.zeros((50000, 4), dtype=np.uint32); ctr[:, 0] = np.arange(50000)
    n = P.normals_from_bits(P.philox4x32(ctr, (7, 9))).reshape(-1)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01 and abs(((n - n.mean()) ** 4).mean() - 3) < 0.1


@pytest.mark.gpu
def test_device_philox_bits_and_normals_match_oracle():
    from vae_training_amd.engine import Engine
    eng = Engine(64, 12, 20)
    n, seed, step, tag = 4099, 0x123456789ABCDEF, 17, 3
    normals, bits = eng.rng_fill(n, seed, step, tag, bits=True)
    nb = (n + 3) // 4
    ctr = np.zeros((nb, 4), dtype=np.uint32); ctr[:, 0] = np.arange(nb); ctr[:, 2] = step; ctr[:, 3] = tag
    want = P.philox4x32(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), want.reshape(-1)[:n])        # bit exact
    assert np.max(np.abs(normals.cpu().numpy() - P.normals_from_bits(want).reshape(-1)[:n])) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["linear_gaussian", "sigmoid", "sphere"])
def test_make_batch_reproduces_datasets_py(kind):
    """Rows are a pure function of (seed, step, tag, global row): equal to the oracle's stream pushed through
    the dataset formulas of datasets.py, independent of how rows are sharded."""
    from vae_training_amd.datasets import LinearGaussianDataset, SigmoidDataset, SphereDataset
    from vae_training_amd.engine import Engine
    ds = {"linear_gaussian": lambda: LinearGaussianDataset(2, 3, 3, 9, var_added=0.04), "sigmoid": lambda: SigmoidDataset(69, 3, 3),
          "sphere": lambda: SphereDataset(69, 3, 3)}[kind]()
    D, L, rows, seed, step = ds.dimension, 5, 300, 99, 12
    eng = Engine(rows, D, L, sigmoid_decoder=(kind == "sigmoid"))
    k, A, dd, did, pad, var = ds.device_spec()
    x, z1, z2 = eng.make_batch(k, A, dd, did, pad, var, rows, seed, step=step, tag=0)
    xa, z1a, z2a = (t.cpu().numpy().astype(np.float64) for t in (x, z1, z2))
    Ah = None if A is None else A.cpu().numpy().astype(np.float64)
    for i in (0, 1, 157, 299):
        q0 = P.noise_block_offset(did)
        n = P.sample_normals(seed, step, 0, i, 4 * q0 + D)
        if kind == "linear_gaussian":
            want = np.concatenate([Ah.reshape(dd, did) @ n[:did], np.zeros(pad)]) + np.sqrt(var) * n[4 * q0:4 * q0 + D]
        elif kind == "sigmoid":
            want = np.concatenate([n[:dd], [1 / (1 + np.exp(-(n[:dd] @ Ah)))], np.zeros(pad)])
        else:
            want = np.concatenate([n[:dd] / np.linalg.norm(n[:dd]), np.zeros(pad)])
        assert np.max(np.abs(xa[i] - want)) < 5e-5, (kind, i)
        z = P.sample_normals(seed, step, 0x40000000, i, L + D)           # model.py:227 column order
        assert np.max(np.abs(np.concatenate([z1a[i], z2a[i]]) - z)) < 5e-5
    # sharding: rows 100..199 generated alone equal rows 100..199 of the full batch, bit for bit
    eng2 = Engine(100, D, L, sigmoid_decoder=(kind == "sigmoid"))
    x2, z12, z22 = eng2.make_batch(k, A, dd, did, pad, var, 100, seed, step=step, tag=0, row0=100)
    assert torch.equal(x2, x[100:200]) and torch.equal(z12, z1[100:200]) and torch.equal(z22, z2[100:200])
    # a different step or tag gives a different batch; moments are right
    x3, _, _ = eng.make_batch(k, A, dd, did, pad, var, rows, seed, step=step + 1, tag=0)
    assert not torch.equal(x3, x)
    big = Engine(65536, D, L, sigmoid_decoder=(kind == "sigmoid"))
    xb, z1b, z2b = big.make_batch(k, A, dd, did, pad, var, 65536, seed, step=1)
    for z in (z1b, z2b):
        assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1) < 0.01
    if kind == "sphere":
        assert torch.allclose(xb[:, :dd].norm(dim=1), torch.ones(65536, device="cuda"), atol=1e-5)


@pytest.mark.gpu
def test_graph_loop_equals_eager_steps(tmp_path):
    """trainer.GraphLoop (hipGraph replay, device RNG, loss ring) == the same steps issued one by one."""
    from vae_training_amd.run import get_dataset, parse_arguments
    from vae_training_amd.trainer import GraphLoop
    from vae_training_amd.vae import VAEModel

    def build():
        args = parse_arguments(["t", "--dataset", "linear_gaussian", "--padding_dim", "9", "-dd", "3", "-ds", "2"])
        ds = get_dataset("linear_gaussian", 2, 9, 100, args)
        return VAEModel(dirname=str(tmp_path), num_batches=10, num_epochs=1, batch_size=100, learning_rate=1e-3, layer_sizes="",
                        encoder_layer_sizes="", state_dict=None, data_fn=None, epsilon=-1.0, tqdm=False, dataset=ds,
                        latent_dimension=20, tunable_decoder_var=True, dataset_name="linear_gaussian")
    a, b, c = build(), build(), build()
    la, lb = GraphLoop(a, steps_per_graph=8, seed=5, moments=False), GraphLoop(b, steps_per_graph=8, seed=5, pipeline=False, moments=False)
    lc = GraphLoop(c, steps_per_graph=8, seed=5, pipeline=False, moments=False)
    assert la.pipeline and len(la.bufs) == 2 and len(lb.bufs) == 1
    la.run(31)                       # pipelined: 2 eager warm-ups + 3 replays of 8 + 5 eager
    la.run(12)                       # odd step count so far: one eager step to regain buffer parity, a replay, 3 eager
    lc.run(43)                       # unpipelined graph
    for _ in range(43):
        lb._one()                    # unpipelined, step by step: make_batch(step_dev) + train_step
    torch.cuda.synchronize()
    assert a.optimizer.state.step == 43 == int(a.optimizer.state.step_dev.item()) == int(b.optimizer.state.step_dev.item())
    assert la.counter.tolist() == [44, 43]         # the last two draws (batches 42, 43) each left "my step + 1" in the other slot
    for m_, l_ in ((a, la), (c, lc)):
        assert torch.equal(m_.model.flat, b.model.flat) and torch.equal(l_.losses(), lb.losses())
    assert la.losses().numel() == 43 and bool(torch.isfinite(la.losses()).all())


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 100, 65536])
def test_make_batch_next_is_make_batch_with_a_self_advancing_step(rows):
    """vaek_make_batch_next draws from counter[which] and stores counter[which ^ 1] = step + 1 -- any grid size."""
    from vae_training_amd.engine import Engine
    eng = Engine(rows, 12, 20)
    A = torch.randn(3, 3, device="cuda")
    counter = torch.tensor([-1, 7], dtype=torch.int32, device="cuda")
    for k in range(4):
        got = eng.make_batch(0, A, 3, 3, 9, 0.25, rows, 11, counter=counter, which=(k + 1) % 2, tag=3, row0=5)
        want = eng.make_batch(0, A, 3, 3, 9, 0.25, rows, 11, step=7 + k, tag=3, row0=5)
        assert all(torch.equal(g, w) for g, w in zip(got, want))
        assert int(counter[k % 2]) == 8 + k and int(counter[(k + 1) % 2]) == 7 + k
    with pytest.raises(Exception, match="invalid"):
        eng.make_batch(0, A, 3, 3, 9, 0.25, rows, 11, counter=counter, which=2)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,hidden,B", [("linear_gaussian", (), 100), ("linear_gaussian", (), 256), ("linear_gaussian", (), 1000), ("linear_gaussian", (), 65536), ("sigmoid", (), 300),
                                            ("sphere", (64,), 500), ("sphere", (64, 32), 500)])
def test_train_step_gen_is_make_batch_next_plus_train_step(kind, hidden, B):
    """vaek_train_step_gen (next batch drawn by spare blocks of the finalize launch on the fused path; two launches
    back to back on the layer-by-layer path) == vaek_make_batch + vaek_train_step, bit for bit, over several steps."""
    from vae_training_amd.engine import Engine
    dd, pad = 3, (9 if kind == "linear_gaussian" else 3)
    D = dd + pad + (1 if kind == "sigmoid" else 0)
    L = 20 if kind == "linear_gaussian" else 6
    k = {"linear_gaussian": 0, "sigmoid": 1, "sphere": 2}[kind]
    A = torch.randn(3, 3, device="cuda") if k == 0 else torch.randn(3, device="cuda")
    eng = Engine(B, D, L, hidden, hidden, -1.0, True, kind == "sigmoid")
    assert bool(eng.fused) == (len(hidden) <= 1)        # linear and one-hidden-layer models have whole-network kernels
    torch.manual_seed(0)
    p0 = torch.randn(eng.P, device="cuda") * 0.3

    def state():
        return [p0.clone(), eng.new_flat(eng.grad_len), eng.new_flat(), eng.new_flat(), torch.zeros(1, dtype=torch.int32, device="cuda")]
    seed, var = 77, 0.25
    # (a) separate calls, host step
    sa = state()
    losses_a = []
    for n in range(5):
        x, z1, z2 = eng.make_batch(k, A, dd, 3, pad, var, B, seed, step=n, tag=1, row0=10)
        eng.train_step(*sa, x, z1, z2, 1e-3)
        losses_a.append(sa[1][eng.P].clone())
    # (b) fused generator
    sb = state()
    counter = torch.tensor([0, 0], dtype=torch.int32, device="cuda")
    bufs = [eng.make_batch(k, A, dd, 3, pad, var, B, seed, counter=counter, which=0, tag=1, row0=10),
            tuple(torch.empty_like(t) for t in (x, z1, z2))]
    losses_b = []
    for n in range(5):
        eng.train_step_gen(*sb, bufs[n % 2], 1e-3, k, A, dd, 3, pad, var, bufs[(n + 1) % 2], seed, counter, (n + 1) % 2, tag=1, row0=10)
        losses_b.append(sb[1][eng.P].clone())
    torch.cuda.synchronize()
    assert counter.tolist() == [6, 5] and int(sb[4]) == 5
    assert torch.equal(torch.stack(losses_a), torch.stack(losses_b))
    for ta, tb in zip(sa[:4], sb[:4]):
        assert torch.equal(ta, tb)
    with pytest.raises(Exception, match="alias"):
        eng.train_step_gen(*sb, bufs[0], 1e-3, k, A, dd, 3, pad, var, bufs[0], seed, counter, 0, tag=1, row0=10)
