// K7 of SURVEY.md 7.1: on-device generation of what feeds the hot path every iteration -- the dataset
// batch (datasets.py:75-84 sphere, :183-195 linear_gaussian, :240-249 sigmoid) and the latent draw
// z ~ N(0,1)^(B x (L+D)) (model.py:225-228, split at vae.py:127-128) -- as one Philox4x32-10 kernel.
// Counter-based: row i of step t draws block q from counter (i, q, t, tag) under key = seed, so a
// batch is reproducible, shardable (a rank generates only its rows, with GLOBAL row indices) and
// replayable from a hipGraph (t is the device-resident Adam step counter).  jax.random's threefry
// streams are not reproduced (not possible without JAX); the distributions are (tests/test_rng.py).
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
__device__ __forceinline__ void normals4(uint4 b, float (&n)[4]) {
    const float u1a = ((float)(b.x >> 8) + 0.5f) * 5.9604644775390625e-08f, u1b = ((float)(b.z >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float ra = sqrtf(-2.f * logf(u1a)), rb = sqrtf(-2.f * logf(u1b));
    float sa, ca, sb, cb;
    sincospif(2.f * ((float)b.y * 2.3283064365386963e-10f), &sa, &ca);
    sincospif(2.f * ((float)b.w * 2.3283064365386963e-10f), &sb, &cb);
    n[0] = ra * ca; n[1] = ra * sa; n[2] = rb * cb; n[3] = rb * sb;
}

struct NormalStream {           // sequential normals of one row
    uint2 key; unsigned row, step, tag, q; int have; float buf[4];
    __device__ __forceinline__ float next() {
        if (have == 0) { normals4(philox4x32_10(make_uint4(row, q++, step, tag), key), buf); have = 4; }
        return buf[4 - have--];
    }
};

struct BatchArgs {
    int kind;                   // 0 linear_gaussian, 1 sigmoid, 2 sphere
    const float* A;             // linear: [dd][did] row-major; sigmoid: [dd]; sphere: unused
    int dd, did, pad; float noise_std;
    float* x; float* z1; float* z2;
    int rows; long long row0; int D, L;
    unsigned long long seed; const int32_t* step_dev; unsigned step_host, tag;
};

// Work items: [0, rows) = one dataset row each (x: a handful of normals, row written as one contiguous
// run); [rows, rows + rows*NZB) = one Philox block (4 normals) of one row's latent stream each, so that
// consecutive lanes store consecutive 16-byte pieces of z1 / z2 -- the 11.5 MB of a 65 536-row batch
// leave as coalesced stores instead of 44 scattered dwords per thread.
__global__ __launch_bounds__(256) void make_batch_kernel(const BatchArgs a) {
    const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const uint2 key = make_uint2((unsigned)a.seed, (unsigned)(a.seed >> 32));
    const unsigned step = a.step_dev ? (unsigned)a.step_dev[0] : a.step_host;
    const long long nx = a.x ? a.rows : 0;
    if (item < nx) {
        const int i = (int)item;
        // up to 16 normals of the row's dataset stream, kept in registers: every index below is static
        // (a runtime-indexed array would live in scratch memory)
        const int nn = a.kind == 0 ? a.did : a.dd;
        float nrm[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float n4[4] = {0.f, 0.f, 0.f, 0.f};
            if (4 * q < nn) normals4(philox4x32_10(make_uint4((unsigned)(a.row0 + i), (unsigned)q, step, a.tag), key), n4);
            nrm[4 * q] = n4[0]; nrm[4 * q + 1] = n4[1]; nrm[4 * q + 2] = n4[2]; nrm[4 * q + 3] = n4[3];
        }
        float* x = a.x + (long long)i * a.D;
        if (a.kind == 0) {                                       // Y = (A X^T)^T, zero padding, optional noise
            for (int d = 0; d < a.dd; ++d) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) if (k < a.did) v = fmaf(a.A[d * a.did + k], nrm[k], v);
                x[d] = v;
            }
            for (int d = a.dd; d < a.D; ++d) x[d] = 0.f;
            if (a.noise_std > 0.f) {                             // noise normals: blocks (did+3)/4 .. of the same stream
                const int q0 = (a.did + 3) / 4;
                for (int d0 = 0; d0 < a.D; d0 += 4) {
                    float n4[4];
                    normals4(philox4x32_10(make_uint4((unsigned)(a.row0 + i), (unsigned)(q0 + d0 / 4), step, a.tag), key), n4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (d0 + k < a.D) x[d0 + k] += a.noise_std * n4[k];
                }
            }
        } else if (a.kind == 1) {                                // [z, sigmoid(z.a), 0...]
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < 16; ++d) if (d < a.dd) { x[d] = nrm[d]; dot = fmaf(nrm[d], a.A[d], dot); }
            x[a.dd] = 1.f / (1.f + expf(-dot));
            for (int d = a.dd + 1; d < a.D; ++d) x[d] = 0.f;
        } else {                                                 // g / |g|, zero padding
            float nsq = 0.f;
#pragma unroll
            for (int d = 0; d < 16; ++d) if (d < a.dd) nsq = fmaf(nrm[d], nrm[d], nsq);
            const float inv = 1.f / sqrtf(nsq);
#pragma unroll
            for (int d = 0; d < 16; ++d) if (d < a.dd) x[d] = nrm[d] * inv;
            for (int d = a.dd; d < a.D; ++d) x[d] = 0.f;
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
    normals4(philox4x32_10(make_uint4((unsigned)(a.row0 + i), (unsigned)q, step, a.tag + 0x40000000u), key), n);
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

__global__ __launch_bounds__(256) void rng_fill_kernel(float* out_n, unsigned* out_u, long long n, unsigned long long seed,
                                                      unsigned step, unsigned tag) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one Philox block = 4 outputs
    if (4 * b >= n) return;
    const uint4 r = philox4x32_10(make_uint4((unsigned)b, (unsigned)(b >> 32), step, tag), make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    float nv[4];
    normals4(r, nv);
    const unsigned uv[4] = {r.x, r.y, r.z, r.w};
    for (int k = 0; k < 4 && 4 * b + k < n; ++k) {
        if (out_n) out_n[4 * b + k] = nv[k];
        if (out_u) out_u[4 * b + k] = uv[k];
    }
}

}  // namespace vaek

using namespace vaek;

extern "C" {

int vaek_make_batch(vaek_ctx* ctx, int32_t kind, const float* A, int32_t dd, int32_t did, int32_t pad, float var_added,
                    float* x, float* z1, float* z2, int32_t rows, int64_t row0, uint64_t seed, const int32_t* step_dev,
                    uint32_t step_host, uint32_t tag, void* stream) {
    if (!ctx || (!z1) != (!z2) || (!x && !z1) || rows <= 0 || kind < 0 || kind > 2 || dd <= 0 || dd > 16 || did > 16 || pad < 0 ||
        (kind != 2 && x && !A) || tag >= 0x40000000u) {
        set_error("vaek_make_batch: invalid argument");
        return VAEK_ERR_INVALID;
    }
    const int D = dd + pad + (kind == 1 ? 1 : 0);
    if (z1 && D != ctx->D) { set_error("vaek_make_batch: dataset dimension %d != context data_dim %d", D, ctx->D); return VAEK_ERR_INVALID; }
    BatchArgs a{};
    a.kind = kind; a.A = A; a.dd = dd; a.did = did; a.pad = pad; a.noise_std = var_added > 0.f ? sqrtf(var_added) : 0.f;
    a.x = x; a.z1 = z1; a.z2 = z2; a.rows = rows; a.row0 = row0; a.D = D; a.L = ctx->L;
    a.seed = seed; a.step_dev = step_dev; a.step_host = step_host; a.tag = tag;
    g_prof = &ctx->prof;
    {
        ProfScope ps("make_batch", (hipStream_t)stream);
        const long long items = (x ? rows : 0) + (z1 ? (long long)rows * ((a.L + a.D + 3) / 4) : 0);
        launch_k(ps, make_batch_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    }
    g_prof = nullptr;
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int vaek_rng_fill(vaek_ctx* ctx, float* normals, uint32_t* bits, int64_t n, uint64_t seed, uint32_t step, uint32_t tag,
                  void* stream) {
    if (!ctx || n < 0 || (!normals && !bits)) { set_error("vaek_rng_fill: invalid argument"); return VAEK_ERR_INVALID; }
    if (n == 0) return VAEK_OK;
    const long long nb = (n + 3) / 4;
    ProfScope ps("rng_fill", (hipStream_t)stream);
    launch_k(ps, rng_fill_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, normals, bits,
             (long long)n, (unsigned long long)seed, step, tag);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int vaek_set_loss_history(vaek_ctx* ctx, float* buf, int64_t cap) {
    if (!ctx || (buf && cap <= 0)) { set_error("vaek_set_loss_history: invalid argument"); return VAEK_ERR_INVALID; }
    ctx->loss_hist = buf;
    ctx->loss_hist_cap = buf ? cap : 0;
    return VAEK_OK;
}

}  // extern "C"
