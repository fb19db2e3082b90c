// Wide LINEAR decoder (one Dense L -> D with D in the thousands: BASELINE config 4, linear-padding at ambient dimension 4096): the
// decoder's forward, the ELBO's elementwise pass and BOTH of the decoder's backward products in ONE pass over x and z2 --
//     x_hat = samples Wd + bd + sigma z2                    (networks.py:80-83)
//     r = x_hat - x,  mse / d eps terms,  dy = r e^{-eps} / B                     (networks.py:95-98 and its gradient)
//     g = dy Wd^T  (dL / d samples),   [dWd | dbd] = [samples | 1]^T dy           (value_and_grad, networks.py:99)
// The layer-by-layer path wrote dy (B x D floats: 537 MB per rank at config 4) and read it back twice, and read x for the ELBO
// pass on top: 3.6 x the algorithmic traffic of the step.  Here dy never leaves the registers.
//
// Decomposition: workgroup (rb, cb) owns a ROW block (RB rows) x a COLUMN block (256 columns), so every element of x and z2 is
// read exactly once, the 20 x 256 slice of Wd lives in registers for the workgroup's lifetime (no weight re-reads), the kernel
// gradient of the slice accumulates in registers over all RB rows (one slab per row block: B / RB = 16 slabs, not one per tile),
// and only g needs a second stage: one partial per column block ([D / 256][B][L] floats, 8 % of the step's algorithmic bytes),
// summed in a fixed order by lwd_reparam_bwd.  Wave w of the 8 owns 32 of the 256 columns; rows go by in sub-slabs of 16.
//
// Everything is in the TRANSPOSED orientation (features on MFMA rows, samples on MFMA columns: v_mfma_f32_16x16x4_f32, exact f32):
//     x_hat^T tile [16 d][16 s] = Wd^T[d, :] . samples^T        A = the weights (registers), B = the sub-slab's samples (LDS)
//     elementwise pass in the accumulator layout: lane (s, g) holds d = 4 g .. 4 g + 3 -- exactly ONE 16-byte load of x and of z2
//     g^T [32 l][16 s] += Wd[l, d] . dy^T tile                  B = the accumulator tile itself (register r = k-step r, the A
//                                                               operand loaded in the same permuted k order: no data movement)
//     [dWd | dbd] [32 l][16 d] += [samples | 1]^T . dy          contraction over the samples, which sit on the lanes: the dy tile
//                                                               goes through a wave-private 1.3 KB LDS transpose
// 42 MFMAs per wave and sub-slab against 32 KB of x and z2 per workgroup: the matrix pipe (157 TF f32) and HBM are balanced
// (~140 us of each at config 4).  Deterministic: fixed-order sums everywhere, no atomics.
#include "vaek_internal.h"

namespace vaek {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WT = 512, WW = 8;          // threads, waves per workgroup
constexpr int WDC = 256, WWC = 32;       // columns per workgroup / per wave
constexpr int WSR = 16;                  // rows per sub-slab
constexpr int WLP = 36;                  // floats per row of the staged samples [s][l] (32 + pad: the A-operand reads walk the rows)
constexpr int WGS = 20;                  // floats per row of a wave's g^T partial [l][s] (4 rows apart = 16 banks apart)
constexpr int WTS = 20;                  // floats per row of the transposed dy tile [s][d]

struct LwdArgs {
    const float* samples; const float* Wd; const float* bd; const float* x; const float* z2; const float* eps_param;
    float eps_cli, inv_bt;
    float* gpart;                        // [ncb][B][L]
    float* slab0; long long slab_stride; // the decoder layer's [kernel | bias] slabs: slab rb at slab0 + rb * slab_stride, element [l * D + d], bias row l = L
    float* part;                         // [nrb * ncb][2]: {mse, d eps} sums of the workgroup's block
    int B, D, L, RB, ncb;
};

template <int KLT>          // k-steps over the latent dimension held in registers: 5 (L <= 20) or 8 (L <= 31)
__global__ __launch_bounds__(WT, 4) void lwd_kernel(const LwdArgs a) {       // two workgroups per CU: their barriers and LDS phases interleave
    extern __shared__ __attribute__((aligned(16))) char lwd_smem[];
    float* sS = reinterpret_cast<float*>(lwd_smem);                      // [2][16][WLP] samples of the sub-slab (+ ones column at l = L)
    float* gB = sS + 2 * WSR * WLP;                                      // [2][8 waves][32][WGS] g^T partials
    float* tT = gB + 2 * WW * 32 * WGS;                                  // [8 waves][2 tiles][16][WTS] transposed dy
    float* red = tT + WW * 2 * WSR * WTS;                                // [8][2]
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, j = lane & 15, g = lane >> 4;
    const int cb = blockIdx.x % a.ncb, rb = blockIdx.x / a.ncb;
    const int D = a.D, L = a.L;
    const int col0 = cb * WDC + wave * WWC;
    const long long row_lo = (long long)rb * a.RB, row_hi = min((long long)a.B, row_lo + a.RB);
    const int nsub = (int)((row_hi - row_lo + WSR - 1) / WSR);
    const float eps = a.eps_param ? a.eps_param[0] * a.eps_cli : a.eps_cli;
    const float inv_var = expf(-eps), sigma = expf(0.5f * eps), dscale = inv_var * a.inv_bt;
    // ---- the wave's slice of the weights, as MFMA A operands --------------------------------------------------------------------
    float aY[2][KLT];    // x_hat^T: tile t2 row d = col0 + 16 t2 + j, k-step kk: l = 4 kk + g
    float aG[2][8];      // g^T: row l = 16 lt + j, k index (t2, r): d = col0 + 16 t2 + 4 g + r
    float bdv[2][4];     // bias in the accumulator layout: d = col0 + 16 t2 + 4 g + r
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
        for (int kk = 0; kk < KLT; ++kk) {
            const int l = 4 * kk + g;
            aY[t2][kk] = l < L ? a.Wd[(long long)min(l, L - 1) * D + col0 + 16 * t2 + j] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bdv[t2][r] = a.bd[col0 + 16 * t2 + 4 * g + r];
    }
#pragma unroll
    for (int lt = 0; lt < 2; ++lt)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int l = 16 * lt + j, d = col0 + 16 * (q >> 2) + 4 * g + (q & 3);
            aG[lt][q] = l < L ? a.Wd[(long long)min(l, L - 1) * D + d] : 0.f;
        }
    f32x4 accW[2][2];
#pragma unroll
    for (int lt = 0; lt < 2; ++lt)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) accW[lt][t2] = f32x4{0.f, 0.f, 0.f, 0.f};
    float e_mse = 0.f, e_deps = 0.f;
    // ---- staging of a sub-slab's samples: thread -> (s = t / 32, l = t % 32); ones at l == L (the bias row of [samples | 1]) -------
    // (two steps: the value is LOADED one sub-slab before it is written to LDS -- written right behind its load, every sub-slab
    // waited out a global-memory round trip, and drained the x / z2 prefetch with it)
    auto stage_load = [&](int i) __attribute__((always_inline)) -> float {
        const int s = t >> 5, l = t & 31;
        const long long row = row_lo + (long long)i * WSR + s;
        const bool in = i < nsub && row < row_hi;
        const float v = a.samples[min(row, row_hi - 1) * L + min(l, L - 1)];       // unconditional, clamped
        return in ? (l < L ? v : (l == L ? 1.f : 0.f)) : 0.f;
    };
    auto stage_put = [&](int i, float v) __attribute__((always_inline)) { sS[((i & 1) * WSR + (t >> 5)) * WLP + (t & 31)] = v; };
    auto load_xz = [&](int i, f32x4 (&xv)[2], f32x4 (&zv)[2]) __attribute__((always_inline)) {
        const long long row = min(row_lo + (long long)i * WSR + j, row_hi - 1);       // (clamped: masked below)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const long long o = row * D + col0 + 16 * t2 + 4 * g;
            xv[t2] = *reinterpret_cast<const f32x4*>(a.x + o);
            zv[t2] = *reinterpret_cast<const f32x4*>(a.z2 + o);
        }
    };
    // fixed-order sum of the 8 waves' g^T partials of sub-slab i -> this column block's partial of g
    auto gsum = [&](int i) __attribute__((always_inline)) {
        for (int o = t; o < WSR * L; o += WT) {
            const int s = o / L, l = o - s * L;
            const long long row = row_lo + (long long)i * WSR + s;
            if (row < row_hi) {
                const float* Gs = gB + (i & 1) * WW * 32 * WGS + l * WGS + s;
                float v[WW];
#pragma unroll
                for (int w = 0; w < WW; ++w) v[w] = Gs[w * 32 * WGS];
                float sum = v[0];
#pragma unroll
                for (int w = 1; w < WW; ++w) sum += v[w];
                a.gpart[((long long)cb * a.B + row) * L + l] = sum;
            }
        }
    };
    f32x4 xv[2], zv[2], xn[2], zn[2];
    stage_put(0, stage_load(0));
    float sv = stage_load(1);
    load_xz(0, xv, zv);
    __syncthreads();
    for (int i = 0; i < nsub; ++i) {
        stage_put(i + 1, sv);
        sv = stage_load(i + 2);
        if (i + 1 < nsub) load_xz(i + 1, xn, zn);
        const float* S = sS + (i & 1) * WSR * WLP;
        const bool valid = row_lo + (long long)i * WSR + j < row_hi;
        // ---- x_hat^T, the elementwise pass, dy^T ---------------------------------------------------------------------------
        float bS[KLT];
#pragma unroll
        for (int kk = 0; kk < KLT; ++kk) bS[kk] = S[j * WLP + 4 * kk + g];       // (columns l >= L of the staged samples are zeros or the ones column, which meets a zero weight)
        f32x4 dy[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KLT; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aY[t2][kk], bS[kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = zv[t2][r];
                const float rr = (acc[r] + bdv[t2][r]) + sigma * z - xv[t2][r];           // x_hat - x, x_hat = y + z2 e^{eps/2}
                const float q = rr * rr * inv_var;
                if (valid) { e_mse += 0.5f * q; e_deps += -0.5f * q + 0.5f * sigma * z * rr * inv_var; }
                dy[t2][r] = valid ? rr * dscale : 0.f;
            }
        }
        // ---- g^T partial of the wave's 32 columns: B operand = the dy^T tiles as they stand ------------------------------------
        f32x4 gacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int lt = 0; lt < 2; ++lt) gacc[lt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aG[lt][q], dy[q >> 2][q & 3], gacc[lt], 0, 0, 0);
        float* G = gB + ((i & 1) * WW + wave) * 32 * WGS;
#pragma unroll
        for (int lt = 0; lt < 2; ++lt)
#pragma unroll
            for (int r = 0; r < 4; ++r) G[(16 * lt + 4 * g + r) * WGS + j] = gacc[lt][r];
        // ---- [dWd | dbd] += [samples | 1]^T dy: dy through the wave's own transpose image ---------------------------------------
        float* T = tT + wave * 2 * WSR * WTS;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) *reinterpret_cast<f32x4*>(T + (t2 * WSR + j) * WTS + 4 * g) = dy[t2];
        // (wave-private: the reads below only need this wave's own writes -- program order + lgkmcnt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float aS[2], bT[2];
#pragma unroll
            for (int lt = 0; lt < 2; ++lt) aS[lt] = S[(4 * kk + g) * WLP + 16 * lt + j];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) bT[t2] = T[(t2 * WSR + 4 * kk + g) * WTS + j];
#pragma unroll
            for (int lt = 0; lt < 2; ++lt)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) accW[lt][t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(aS[lt], bT[t2], accW[lt][t2], 0, 0, 0);
        }
        __syncthreads();             // the sub-slab's g^T partials are in LDS; the next sub-slab's samples are staged
        gsum(i);
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) { xv[t2] = xn[t2]; zv[t2] = zn[t2]; }
    }
    // ---- the slice of [dWd | dbd] of this row block: accumulator (lane (d, g), register r -> row l = 16 lt + 4 g + r) ---------------
    float* slab = a.slab0 + (long long)rb * a.slab_stride;
#pragma unroll
    for (int lt = 0; lt < 2; ++lt)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int l = 16 * lt + 4 * g + r;
                if (l <= L) slab[(long long)l * D + col0 + 16 * t2 + j] = accW[lt][t2][r];
            }
    // ---- the block's two loss sums: lanes by xor-shuffle, the waves in order -------------------------------------------------
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { e_mse += __shfl_xor(e_mse, o, 64); e_deps += __shfl_xor(e_deps, o, 64); }
    if (lane == 0) { red[2 * wave] = e_mse; red[2 * wave + 1] = e_deps; }
    __syncthreads();
    if (t == 0) {
        float m = red[0], d = red[1];
#pragma unroll
        for (int w = 1; w < WW; ++w) { m += red[2 * w]; d += red[2 * w + 1]; }
        a.part[2 * blockIdx.x] = m; a.part[2 * blockIdx.x + 1] = d;
    }
}

// Sum over a 256-thread block; result valid in thread 0 (as elbo.hip's)
__device__ __forceinline__ float lwd_block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += red[i];
    return r;
}
// The second stage, one launch, split s = rows [r0, r1):
//   * dsamp = sum over the column blocks of g's partials (fixed order) + mu / B, and the partial sums of dsamp . z1 for d logvar_e:
//     reparam_bwd_kernel (elbo.hip) with its input in ncb pieces;
//   * the ELBO partials elbo_kernel would have written (epartial[s] = {mse, mu^2, d eps, 0}): every split the mu^2 of its rows,
//     split 0 on top the block sums of lwd_kernel (fixed order) -- and the Adam step counter.
__global__ __launch_bounds__(256) void lwd_second_kernel(const float* gpart, int ncb, float* dsamp, const float* mu, const float* z1,
                                                         float* partial, int rows, int L, int rows_per_split, float inv_bt,
                                                         const float* part, int nblk, float* epartial, int32_t* step_dev) {
    extern __shared__ float sh[];   // 256 + 8 floats
    float* red = sh + 256;
    const int s = blockIdx.x;
    const int r0 = s * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    const int G = 256 / L;
    const int col = threadIdx.x % L, grp = threadIdx.x / L;
    float acc = 0.f, musq = 0.f;
    if (grp < G) {
        for (int r = r0 + grp; r < r1; r += G) {
            const long long o = (long long)r * L + col, cs = (long long)rows * L;
            float d = 0.f;
            for (int c0 = 0; c0 < ncb; c0 += 16) {         // 16 loads in flight, then their sum in block order
                float v[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) v[c] = gpart[(long long)min(c0 + c, ncb - 1) * cs + o];
#pragma unroll
                for (int c = 0; c < 16; ++c) d += c0 + c < ncb ? v[c] : 0.f;
            }
            const float mv = mu[o];
            acc += d * z1[o];
            musq += mv * mv;
            dsamp[o] = d + mv * inv_bt;
        }
    }
    sh[threadIdx.x] = grp < G ? acc : 0.f;
    __syncthreads();
    if (threadIdx.x < L) {
        float t = 0.f;
        for (int g2 = 0; g2 < G; ++g2) t += sh[g2 * L + threadIdx.x];
        partial[(long long)s * L + threadIdx.x] = t;
    }
    float mse = 0.f, deps = 0.f;
    if (s == 0)
        for (int k = threadIdx.x; k < nblk; k += blockDim.x) { mse += part[2 * k]; deps += part[2 * k + 1]; }
    const float t_mse = lwd_block_sum(mse, red), t_deps = lwd_block_sum(deps, red), t_musq = lwd_block_sum(musq, red);
    if (threadIdx.x == 0) {
        float* p = epartial + (long long)s * 4;
        p[0] = t_mse; p[1] = t_musq; p[2] = t_deps; p[3] = 0.f;
        if (s == 0 && step_dev) step_dev[0] += 1;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
bool lwd_supported(int B, int D, int L) { return D >= 1024 && D % WDC == 0 && L >= 1 && L <= 31 && B >= 256; }
// rows per row block: the grid (row blocks x column blocks) should be about one workgroup per CU
int lwd_row_block(int B, int D, int n_cu) {
    const int ncb = D / WDC, nrb = std::max(1, 2 * n_cu / ncb);      // two workgroups per CU
    return std::max(WSR, (int)(((long long)B + nrb - 1) / nrb + WSR - 1) / WSR * WSR);
}
size_t lwd_gpart_bytes(int B, int D, int L) { return (size_t)(D / WDC) * B * L * sizeof(float); }

int launch_lwd(const float* samples, const float* Wd, const float* bd, const float* x, const float* z2, const float* eps_param, float eps_cli,
               float inv_bt, float* gpart, float* slab0, int64_t slab_stride, float* part, int B, int D, int L, int RB, hipStream_t st) {
    LwdArgs a{};
    a.samples = samples; a.Wd = Wd; a.bd = bd; a.x = x; a.z2 = z2; a.eps_param = eps_param; a.eps_cli = eps_cli; a.inv_bt = inv_bt;
    a.gpart = gpart; a.slab0 = slab0; a.slab_stride = slab_stride; a.part = part; a.B = B; a.D = D; a.L = L; a.RB = RB; a.ncb = D / WDC;
    const int nrb = (B + RB - 1) / RB;
    const size_t lds = sizeof(float) * (2 * WSR * WLP + 2 * WW * 32 * WGS + WW * 2 * WSR * WTS + 2 * WW) + 64;
    int dev = 0;
    VAEK_HIP_CHECK(hipGetDevice(&dev));
    static thread_local unsigned char attr_set[64] = {};
    if (!attr_set[dev & 63]) {
        VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)lwd_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)lwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set[dev & 63] = 1;
    }
    ProfScope ps("lwd_decoder_fwd_bwd", st);
    if (L <= 20) launch_k(ps, lwd_kernel<5>, dim3((unsigned)(nrb * a.ncb)), dim3(WT), lds, st, a);
    else launch_k(ps, lwd_kernel<8>, dim3((unsigned)(nrb * a.ncb)), dim3(WT), lds, st, a);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}
int launch_lwd_second(const float* gpart, int ncb, float* dsamp, const float* mu, const float* z1, float* partial, int rows, int L, int S,
                      int rows_per_split, float inv_bt, const float* part, int nblk, float* epartial, int32_t* step_dev, hipStream_t st) {
    if (L > 256) { set_error("lwd second stage: latent_dim %d > 256", L); return VAEK_ERR_INVALID; }
    ProfScope ps("lwd_reparam_bwd_elbo_reduce", st);
    launch_k(ps, lwd_second_kernel, dim3(S), dim3(256), 264 * sizeof(float), st, gpart, ncb, dsamp, mu, z1, partial, rows, L, rows_per_split, inv_bt,
             part, nblk, epartial, step_dev);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

}  // namespace vaek
