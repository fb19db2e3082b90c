"""Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123)
in NumPy -- TEST INFRASTRUCTURE for csrc/rng.hip (the on-device generator of datasets.py:75-84,
183-195, 240-249 draws and model.py:225-228 latents).  Pinned by Random123's published known-answer
vectors (tests/test_rng.py).  jax.random (threefry) streams are NOT reproduced: only the
distributions are part of the reference's contract."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(counter, key, rounds=10):
    """counter: (..., 4) uint32, key: (..., 2) uint32 -> (..., 4) uint32."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.array(np.broadcast_to(np.asarray(key, dtype=np.uint32), c.shape[:-1] + (2,)), dtype=np.uint32, copy=True)
    for _ in range(rounds):
        p0 = c[..., 0].astype(np.uint64) * M0
        p1 = c[..., 2].astype(np.uint64) * M1
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c = np.stack([hi1 ^ c[..., 1] ^ k[..., 0], lo1, hi0 ^ c[..., 3] ^ k[..., 1], lo0], axis=-1)
        k = np.stack([k[..., 0] + W0, k[..., 1] + W1], axis=-1)
    return c


def normals_from_bits(bits):
    """The kernel's Box-Muller: per Philox block (4 words) two pairs; u1 = ((w >> 8) + 0.5) 2^-24 in
    (0,1), u2 = w 2^-32; (r cos t, r sin t) with r = sqrt(-2 ln u1), t = 2 pi u2.  float64 here."""
    b = np.asarray(bits, dtype=np.uint32)
    u1 = ((b[..., 0::2] >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
    u2 = b[..., 1::2].astype(np.float64) * 2.0 ** -32
    r = np.sqrt(-2.0 * np.log(u1))
    out = np.empty(b.shape, dtype=np.float64)
    out[..., 0::2] = r * np.cos(2 * np.pi * u2)
    out[..., 1::2] = r * np.sin(2 * np.pi * u2)
    return out


def sample_normals(seed, step, tag, sample_index, n):
    """n normals of one sample row exactly as csrc/rng.hip draws them: block q uses
    counter = (sample_lo, q, step, tag), key = (seed_lo, seed_hi)."""
    nblk = (n + 3) // 4
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = np.uint32(sample_index & 0xFFFFFFFF)
    ctr[:, 1] = np.arange(nblk, dtype=np.uint32)
    ctr[:, 2] = np.uint32(step & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32(tag)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return normals_from_bits(philox4x32(ctr, key)).reshape(-1)[:n]


def noise_block_offset(did):
    """Dataset noise normals (linear_gaussian var_added > 0) start at Philox block ceil(did / 4) of the row's
    dataset stream, one block per 4 data dims -- exactly as csrc/rng.hip draws them."""
    return (did + 3) // 4
