// Device side of K7 (rng.hip): Philox4x32-10, Box-Muller, and the batch-drawing work items -- shared by the
// stand-alone generator kernel and by the finalize kernel of the fused path, whose spare blocks draw the NEXT
// step's batch (vaek_train_step_gen).
#pragma once
#include "vaek_internal.h"

namespace vaek {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)c.x * 0xD2511F53ull;
        const unsigned long long p1 = (unsigned long long)c.z * 0xCD9E8D57ull;
        c = make_uint4((unsigned)(p1 >> 32) ^ c.y ^ k.x, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k.y, (unsigned)p0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}

// Box-Muller on one Philox block: 4 words -> 4 normals.  u1 in (0,1) from 24 bits, u2 in [0,1) from 32.
// On the hardware transcendentals: r = sqrt(-2 ln u1) = v_sqrt(-2 ln 2 * v_log(u1)) (v_log_f32 is log2), and v_sin_f32 / v_cos_f32 take
// their argument in REVOLUTIONS, so cos(2 pi u2) is one instruction on u2 -- 6 quarter-rate instructions per block where the
// library's logf / sincospif spent ~200 (the draw inside vaek_train_steps_gen's streamers runs at two waves per SIMD: nothing hides
// a long dependent chain there).  Within 5e-6 of the float64 Box-Muller of oracle/philox.py (tests/test_rng.py); every generator
// entry point shares this function, so they stay bit-identical to each other.
__device__ __forceinline__ void normals4(uint4 b, float (&n)[4]) {
    const float u1a = ((float)(b.x >> 8) + 0.5f) * 5.9604644775390625e-08f, u1b = ((float)(b.z >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1a));
    const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1b));
    const float ta = (float)b.y * 2.3283064365386963e-10f, tb = (float)b.w * 2.3283064365386963e-10f;
    n[0] = ra * __builtin_amdgcn_cosf(ta); n[1] = ra * __builtin_amdgcn_sinf(ta);
    n[2] = rb * __builtin_amdgcn_cosf(tb); n[3] = rb * __builtin_amdgcn_sinf(tb);
}

struct NormalStream {           // sequential normals of one row
    uint2 key; unsigned row, step, tag, q; int have; float buf[4];
    __device__ __forceinline__ float next() {
        if (have == 0) { normals4(philox4x32_10(make_uint4(row, q++, step, tag), key), buf); have = 4; }
        return buf[4 - have--];
    }
};


// The dataset stream of global row `grow` at RNG step `step`: up to 16 normals (blocks 0 .. 3 under tag a.tag), kept in
// registers -- every index below is static (a runtime-indexed array would live in scratch memory).
__device__ __forceinline__ void dataset_normals(const BatchArgs& a, unsigned step, long long grow, uint2 key, float (&nrm)[16]) {
    const int nn = a.kind == 0 ? a.did : a.dd;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float n4[4] = {0.f, 0.f, 0.f, 0.f};
        if (4 * q < nn) normals4(philox4x32_10(make_uint4((unsigned)grow, (unsigned)q, step, a.tag), key), n4);
        nrm[4 * q] = n4[0]; nrm[4 * q + 1] = n4[1]; nrm[4 * q + 2] = n4[2]; nrm[4 * q + 3] = n4[3];
    }
}
// columns c0 .. c0 + 3 of that row (datasets.py:183-195 linear_gaussian, :240-249 sigmoid, :75-84 sphere) from its normals
__device__ __forceinline__ void dataset_cols4(const BatchArgs& a, unsigned step, long long grow, uint2 key, const float (&nrm)[16], int c0, float (&o)[4]) {
    auto pick = [&](int d) { float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v = (k == d) ? nrm[k] : v;
        return v; };
    if (a.kind == 0) {                                       // Y = (A X^T)^T, zero padding, optional noise
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = c0 + c;
            float v = 0.f;
            if (d < a.dd) {
#pragma unroll
                for (int k = 0; k < 16; ++k) if (k < a.did) v = fmaf(a.A[d * a.did + k], nrm[k], v);
            }
            o[c] = v;
        }
        if (a.noise_std > 0.f) {                             // noise normals: blocks (did+3)/4 .. of the same stream
            float n4[4];
            normals4(philox4x32_10(make_uint4((unsigned)grow, (unsigned)((a.did + 3) / 4 + c0 / 4), step, a.tag), key), n4);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaf(a.noise_std, n4[c], o[c]);
        }
    } else if (a.kind == 1) {                                // [z, sigmoid(z.a), 0...]
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) if (d < a.dd) dot = fmaf(nrm[d], a.A[d], dot);
        const float sg = 1.f / (1.f + expf(-dot));
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int d = c0 + c; o[c] = d < a.dd ? pick(d) : (d == a.dd ? sg : 0.f); }
    } else {                                                 // g / |g|, zero padding
        float nsq = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) if (d < a.dd) nsq = fmaf(nrm[d], nrm[d], nsq);
        const float inv = 1.f / sqrtf(nsq);
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int d = c0 + c; o[c] = d < a.dd ? pick(d) * inv : 0.f; }
    }
}
// block q (4 normals) of the row's latent stream: the draw of model.py:227 in its column order, normal n of a row is element
// n & 3 of Philox block n >> 2 under tag + 2^30 (columns [0, L) are z1, [L, L + D) z2: vae.py:127-128)
__device__ __forceinline__ void latent_block(const BatchArgs& a, unsigned step, long long grow, uint2 key, int q, float (&n)[4]) {
    normals4(philox4x32_10(make_uint4((unsigned)grow, (unsigned)q, step, a.tag + 0x40000000u), key), n);
}

// Work items: [0, rows*NXB) = 4 columns of one dataset row each; then rows*NZB items = one Philox block (4
// normals) of one row's latent stream each, so that
// consecutive lanes store consecutive 16-byte pieces of z1 / z2 -- the 11.5 MB of a 65 536-row batch
// leave as coalesced stores instead of 44 scattered dwords per thread.
__device__ __forceinline__ void make_batch_items(const BatchArgs& a, unsigned step, long long item) {
    const uint2 key = make_uint2((unsigned)a.seed, (unsigned)(a.seed >> 32));
    const int nxb = (a.D + 3) / 4;
    const long long nx = a.x ? (long long)a.rows * nxb : 0;
    if (item < nx) {
        // one item = 4 consecutive columns of one dataset row (consecutive lanes store consecutive 16-byte pieces).
        // Every item of a row re-draws the row's few dataset normals -- one Philox call at dd <= 4 -- which is far
        // cheaper than one thread walking a whole row with strided dword stores (that was the long pole: 256
        // workgroups for 65 536 rows took as long as the 2 048 of the latent draw).
        const int i = (int)(item / nxb), c0 = 4 * (int)(item % nxb);
        float nrm[16], o[4];
        dataset_normals(a, step, a.row0 + i, key, nrm);
        dataset_cols4(a, step, a.row0 + i, key, nrm, c0, o);
        float* x = a.x + (long long)i * a.D + c0;
        if (a.D % 4 == 0) {
            *reinterpret_cast<float4*>(x) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c0 + c < a.D) x[c] = o[c];
        }
        return;
    }
    if (!a.z1) return;
    // latent draw of model.py:227 in its column order: z[:, :L] = z1, z[:, L:] = z2  (vae.py:127-128);
    // normal n of a row is element n & 3 of Philox block n >> 2 under tag + 2^30
    const int nzb = (a.L + a.D + 3) / 4;
    const long long zi = item - nx;
    if (zi >= (long long)a.rows * nzb) return;
    const int i = (int)(zi / nzb), q = (int)(zi % nzb);
    float n[4];
    latent_block(a, step, a.row0 + i, key, q, n);
    const int c0 = 4 * q;
    float* z1 = a.z1 + (long long)i * a.L;
    float* z2 = a.z2 + (long long)i * a.D;
    if (c0 + 3 < a.L && a.L % 4 == 0) {
        *reinterpret_cast<float4*>(z1 + c0) = make_float4(n[0], n[1], n[2], n[3]);
    } else if (c0 >= a.L && (c0 - a.L) + 3 < a.D && a.D % 4 == 0 && a.L % 4 == 0) {
        *reinterpret_cast<float4*>(z2 + (c0 - a.L)) = make_float4(n[0], n[1], n[2], n[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            if (c < a.L) z1[c] = n[k];
            else if (c < a.L + a.D) z2[c - a.L] = n[k];
        }
    }
}


__device__ __forceinline__ unsigned make_batch_step(const BatchArgs& a) {
    return a.counter ? (unsigned)a.counter[a.which] : (a.step_dev ? (unsigned)a.step_dev[0] : a.step_host);
}

// Self-advancing generator (vaek_make_batch_next / vaek_train_step_gen).  The counter is a PAIR and launches
// alternate `which`: a launch reads counter[which] and one of its threads stores counter[which ^ 1] = step + 1 for
// the next launch -- nobody in this launch reads the slot that is written, so there is nothing to order and no
// atomic.  (A single slot advanced by a last-block ticket was measured first: 640 device-scope atomics on one
// address serialise at the memory side, ~70 ns each -- 45 us on a 7 us kernel.)
__device__ __forceinline__ void make_batch_advance(const BatchArgs& a, unsigned step, bool first_thread) {
    if (a.counter && first_thread) a.counter[a.which ^ 1] = (int32_t)(step + 1);
}

__host__ __device__ inline long long make_batch_item_count(const BatchArgs& a) {
    return (a.x ? (long long)a.rows * ((a.D + 3) / 4) : 0) + (a.z1 ? (long long)a.rows * ((a.L + a.D + 3) / 4) : 0);
}

}  // namespace vaek
