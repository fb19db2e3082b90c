// Fused train step for the reference's LINEAR VAEs (encoder = one Dense D->L, decoder = one Dense
// L->D, optionally the sigmoid dataset's second SigDecoder): every "" layer-size experiment of
// seed_linpadding_expts.sh / sigmoid_vae_padding_expts.sh, i.e. the configuration the headline
// metric is quoted on (D=12, L=20, batch 65 536).  Two launches per VAE.train_step
// (networks.py:87-101) instead of ~10:
//
//   fused_linear_kernel   x, z1, z2 are read from HBM exactly once (the algorithmic 4*(2D+L) bytes
//       per sample).  Phase 1, one thread per sample on the VALU with the 2 KB of weights broadcast
//       from LDS: mu = x We + be; samples = mu + e^{lv/2} z1; y = samples Wd + bd (+ sigmoid head);
//       r = y + e^{eps/2} z2 - x; dy = r e^{-eps}/B; g = dy Wd^T; dmu = g + mu/B  (networks.py:61-84,
//       :94-98 and their hand-derived backward).
//       Phase 2, the batch-reduction GEMMs on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact
//       fmaf chains): [samples|1]^T [dy|dys] and [x|1]^T [dmu|g*z1] give dWd, dbd, (dWs, dbs,) dWe,
//       dbe and the reparameterisation part of d epsilon_p in one pass; the operands go through a
//       feature-major LDS image T[feature][sample] (row stride = 2 mod 32 banks: the thread-per-
//       sample writes and the MFMA operand reads are both conflict-free).
//       Each workgroup emits ONE partial gradient row; no float atomics anywhere.
//   fused_finalize_kernel fixed-order sum of the partial rows, the closed-form KL / log-variance
//       terms, loss/Dkl/mse means, and Adam (flax.optim.Adam.apply_gradient, networks.py:100).
#include "comm_dev.h"
#include "rng_dev.h"
#include "vaek_internal.h"

namespace vaek {

using f32x4 = __attribute__((ext_vector_type(4))) float;

using f32x2 = __attribute__((ext_vector_type(2))) float;

// In-kernel phase stamps (cdna_hip_programming.md section 7): ONE asm statement, fenced for the scheduler.
#ifdef VAEK_STAMPS
#define VAEK_STAMP(i)                                                                        \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        unsigned long long _t;                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); \
        if (a.stamps && (threadIdx.x & 63) == 0) a.stamps[((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = _t; \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define VAEK_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int DP, int LP, bool SIG, int TILE>
struct FusedGeom {
    static constexpr int NW = TILE / 64;
    static constexpr int TS = TILE + 2;                 // = 2 (mod 32): conflict-free both ways
    static constexpr int NB1 = DP * (SIG ? 2 : 1);      // columns of B1 = [dy | dys]
    static constexpr int NB2 = 2 * LP;                  // columns of B2 = [dmu | g*z1]
    static constexpr int A1 = 0;                        // feature rows of T
    static constexpr int A2 = A1 + LP + 1;
    static constexpr int B1 = A2 + DP + 1;
    static constexpr int B2 = B1 + NB1;
    static constexpr int NF = B2 + NB2;
    static constexpr int IB1 = (LP + 1 + 15) / 16, JB1 = (NB1 + 15) / 16;
    static constexpr int IB2 = (DP + 1 + 15) / 16, JB2 = (NB2 + 15) / 16;
    static constexpr int NBLK = IB1 * JB1 + IB2 * JB2;
    // The MFMA phase reads 16-row blocks without masking: rows past a block's last feature alias the
    // NEXT block's rows (finite data) and only feed output rows/columns nobody reads; NF_PAD keeps the
    // last block's overrun inside the allocation.
    static constexpr int NF_PAD = B2 + JB2 * 16;
    static constexpr int T_FLOATS = NF_PAD * TS;
    static constexpr int R_FLOATS = NW * NBLK * 256;
    // weight image = the flat parameter layout at the PADDED sizes (identical to the real one when exact)
    static constexpr int W_WE = 0, W_BE = DP * LP, W_WD = W_BE + LP, W_BD = W_WD + LP * DP, W_WS = W_BD + DP,
                         W_BS = W_WS + (SIG ? LP * DP : 0), W_LV = W_BS + (SIG ? DP : 0), W_SD = W_LV + LP + 4,
                         W_N = (W_SD + LP + 3) / 4 * 4;
    static constexpr int LDS_FLOATS = W_N + (T_FLOATS > R_FLOATS ? T_FLOATS : R_FLOATS) + 16;
};

// Streams the rows of a [R][C] weight matrix of the LDS image through registers in blocks of RB rows,
// double buffered: the ds_read_b128s of block k+1 are all issued before the packed FMAs of block k
// run.  With ONE wave per SIMD (B = 65 536 gives each CU 256 samples = 4 waves) nothing else hides
// the ~100-cycle LDS latency; left to itself hipcc keeps only two reads in flight.
template <int R, int C, int RB, typename F>
__device__ __forceinline__ void stream_rows(const float* w, F&& f) {
    static_assert(C % 2 == 0, "even row length");
    constexpr int NBK = (R + RB - 1) / RB;
    f32x2 buf[2][RB][C / 2];
    auto load = [&](int k, int which) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (k * RB + r < R) {
#pragma unroll
                for (int j = 0; j < C / 2; ++j) buf[which][r][j] = *reinterpret_cast<const f32x2*>(w + (k * RB + r) * C + 2 * j);
            }
    };
    load(0, 0);
#pragma unroll
    for (int k = 0; k < NBK; ++k) {
        if (k + 1 < NBK) load(k + 1, (k + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (k * RB + r < R) f(k * RB + r, buf[k & 1][r]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
constexpr int rows_per_block(int c) { return c >= 80 ? 1 : 80 / c; }

template <int DP, int LP, bool SIG, int TILE, bool EXACT>
__global__ __launch_bounds__(TILE) void fused_linear_kernel(const float* __restrict__ params, const FusedArgs a) {
    using G = FusedGeom<DP, LP, SIG, TILE>;
    static_assert(DP % 2 == 0 && LP % 2 == 0, "packed f32 math wants even padded dims");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sW0 = lds;
    float* T = lds + G::W_N;
    float* red = T + (G::T_FLOATS > G::R_FLOATS ? G::T_FLOATS : G::R_FLOATS);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int D = EXACT ? DP : a.D, L = EXACT ? LP : a.L;
    const int off_be = D * L, off_wd = off_be + L, off_bd = off_wd + L * D, off_ws = off_bd + D, off_bs = off_ws + L * D;
    const int off_epsp = SIG ? off_bs + D : off_ws;
    const bool vecD = (D % 4 == 0) && (DP % 4 == 0), vecL = (L % 4 == 0) && (LP % 4 == 0);

    float xr[DP], z2r[DP], z1r[LP];
    auto load_inputs = [&](int tile) -> bool {         // one thread = one sample; issued before anything waits
        const long long b = (long long)tile * TILE + t;
        const bool valid = b < a.B;
#pragma unroll
        for (int d = 0; d < DP; ++d) { xr[d] = 0.f; z2r[d] = 0.f; }
#pragma unroll
        for (int l = 0; l < LP; ++l) z1r[l] = 0.f;
        if (valid) {
            const float* px = a.x + b * D; const float* pz2 = a.z2 + b * D; const float* pz1 = a.z1 + b * L;
            if (vecD) {
#pragma unroll
                for (int d = 0; d < DP; d += 4) if (d < D) {
                    const float4 u = *reinterpret_cast<const float4*>(px + d);
                    const float4 w = *reinterpret_cast<const float4*>(pz2 + d);
                    xr[d] = u.x; xr[d + 1] = u.y; xr[d + 2] = u.z; xr[d + 3] = u.w;
                    z2r[d] = w.x; z2r[d + 1] = w.y; z2r[d + 2] = w.z; z2r[d + 3] = w.w;
                }
            } else {
#pragma unroll
                for (int d = 0; d < DP; ++d) if (d < D) { xr[d] = px[d]; z2r[d] = pz2[d]; }
            }
            if (vecL) {
#pragma unroll
                for (int l = 0; l < LP; l += 4) if (l < L) {
                    const float4 u = *reinterpret_cast<const float4*>(pz1 + l);
                    z1r[l] = u.x; z1r[l + 1] = u.y; z1r[l + 2] = u.z; z1r[l + 3] = u.w;
                }
            } else {
#pragma unroll
                for (int l = 0; l < LP; ++l) if (l < L) z1r[l] = pz1[l];
            }
        }
        return valid;
    };
    bool valid = load_inputs(blockIdx.x);              // in flight under the weight-image build below

    // ---- weight image -> LDS: a straight copy of the flat parameters when exact -----------------
    if (EXACT) {
        for (int i = t; i < G::W_LV + LP; i += TILE) sW0[i] = params[i];
    } else {
        for (int i = t; i < G::W_N; i += TILE) sW0[i] = 0.f;
        __syncthreads();
        for (int i = t; i < D * L; i += TILE) {
            sW0[G::W_WE + (i / L) * LP + (i % L)] = params[i];
            sW0[G::W_WD + (i / D) * DP + (i % D)] = params[off_wd + i];
            if (SIG) sW0[G::W_WS + (i / D) * DP + (i % D)] = params[off_ws + i];
        }
        if (t < L) sW0[G::W_BE + t] = params[off_be + t];
        if (t < D) { sW0[G::W_BD + t] = params[off_bd + t]; if (SIG) sW0[G::W_BS + t] = params[off_bs + t]; }
    }
    if (t < LP) sW0[G::W_SD + t] = t < L ? expf(0.5f * params[off_epsp + t]) : 0.f;   // e^{lv/2}, networks.py:73
    const float eps = a.off_eps >= 0 ? params[a.off_eps] * a.eps_cli : a.eps_cli;
    const float inv_var = expf(-eps), sigma = expf(0.5f * eps);
    const float dscale = inv_var * a.inv_bt;

    f32x4 acc1[G::IB1][G::JB1], acc2[G::IB2][G::JB2];
#pragma unroll
    for (int i = 0; i < G::IB1; ++i)
#pragma unroll
        for (int j = 0; j < G::JB1; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < G::IB2; ++i)
#pragma unroll
        for (int j = 0; j < G::JB2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s_mse = 0.f, s_deps = 0.f, s_musq = 0.f;
    __syncthreads();           // weight image complete
    VAEK_STAMP(0);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        // ---- phase 1: one thread = one sample, packed f32 FMAs (v_pk_fma_f32), weights broadcast
        //      from the LDS image (ds_read_b128, every lane the same address) ---------------------
        if (tile != (int)blockIdx.x) valid = load_inputs(tile);
        VAEK_STAMP(1);         // inputs landed
        float* Tt = T + t;     // this sample's column of the feature-major image (wave-private columns)
        // opaque (scalar) offset, x4 so the 16-byte alignment stays visible: without it LICM hoists the
        // ~500 loop-invariant weight reads out of the tile loop and spills them
        int wo = 0;
        asm volatile("" : "+s"(wo));
        const float* sW = sW0 + 4 * wo;
        // mu = x We + be, two latents per packed FMA
        // (e^{lv/2} is read here, with the bias, ahead of every write to the sample image: the opaque
        // offset makes the two LDS regions look aliased, so a later read would be fenced per element)
        f32x2 mu2[LP / 2], sd2[LP / 2];
#pragma unroll
        for (int j = 0; j < LP / 2; ++j) {
            mu2[j] = *reinterpret_cast<const f32x2*>(sW + G::W_BE + 2 * j);
            sd2[j] = *reinterpret_cast<const f32x2*>(sW + G::W_SD + 2 * j);
        }
        stream_rows<DP, LP, rows_per_block(LP)>(sW + G::W_WE, [&](int d, const f32x2(&w)[LP / 2]) {
#pragma unroll
            for (int j = 0; j < LP / 2; ++j) mu2[j] = __builtin_elementwise_fma(f32x2{xr[d], xr[d]}, w[j], mu2[j]);
        });
        float mu[LP], smp[LP];
#pragma unroll
        for (int l = 0; l < LP; ++l) {
            mu[l] = mu2[l / 2][l & 1];
            smp[l] = fmaf(sd2[l / 2][l & 1], z1r[l], mu[l]);               // networks.py:73-74
            Tt[(G::A1 + l) * G::TS] = smp[l];
            s_musq = valid ? fmaf(mu[l], mu[l], s_musq) : s_musq;
        }
        Tt[(G::A1 + LP) * G::TS] = 1.f;
        VAEK_STAMP(2);         // mu, samples done
        // y = samples Wd + bd (and the sigmoid head), two data dims per packed FMA
        f32x2 y2[DP / 2], ys2[SIG ? DP / 2 : 1];
#pragma unroll
        for (int i = 0; i < DP / 2; ++i) {
            y2[i] = *reinterpret_cast<const f32x2*>(sW + G::W_BD + 2 * i);
            if (SIG) ys2[i] = *reinterpret_cast<const f32x2*>(sW + G::W_BS + 2 * i);
        }
        stream_rows<LP, DP, rows_per_block(DP)>(sW + G::W_WD, [&](int l, const f32x2(&w)[DP / 2]) {
#pragma unroll
            for (int i = 0; i < DP / 2; ++i) y2[i] = __builtin_elementwise_fma(f32x2{smp[l], smp[l]}, w[i], y2[i]);
        });
        if (SIG)
            stream_rows<LP, DP, rows_per_block(DP)>(sW + G::W_WS, [&](int l, const f32x2(&w)[DP / 2]) {
#pragma unroll
                for (int i = 0; i < DP / 2; ++i) ys2[i] = __builtin_elementwise_fma(f32x2{smp[l], smp[l]}, w[i], ys2[i]);
            });
        f32x2 dy2[DP / 2], dys2[SIG ? DP / 2 : 1];
#pragma unroll
        for (int d = 0; d < DP; ++d) {
            float xh = fmaf(sigma, z2r[d], y2[d / 2][d & 1]);
            float sg = 0.f;
            if (SIG) { sg = 1.f / (1.f + expf(-ys2[d / 2][d & 1])); xh += sg; }
            const float r = (valid && d < D) ? xh - xr[d] : 0.f;
            const float q = r * r * inv_var;
            s_mse = fmaf(0.5f, q, s_mse);
            s_deps += -0.5f * q + 0.5f * sigma * z2r[d] * r * inv_var;
            const float dyd = r * dscale;
            dy2[d / 2][d & 1] = dyd;
            Tt[(G::B1 + d) * G::TS] = dyd;
            if (SIG) { const float ds = dyd * sg * (1.f - sg); dys2[d / 2][d & 1] = ds; Tt[(G::B1 + DP + d) * G::TS] = ds; }
            Tt[(G::A2 + d) * G::TS] = xr[d];
        }
        Tt[(G::A2 + DP) * G::TS] = 1.f;
        VAEK_STAMP(3);         // y, dy done
        // g = dy Wd^T (+ dys Ws^T): packed over the data dim, the two halves added at the end.  The
        // decoder weights are read again through a second opaque offset so they are not kept live
        // (and spilled) from their first use.
        int wo2 = 0;
        asm volatile("" : "+s"(wo2));
        const float* sW2 = sW0 + 4 * wo2;
        f32x2 g2[LP];
#pragma unroll
        for (int l = 0; l < LP; ++l) g2[l] = f32x2{0.f, 0.f};
        stream_rows<LP, DP, rows_per_block(DP)>(sW2 + G::W_WD, [&](int l, const f32x2(&w)[DP / 2]) {
#pragma unroll
            for (int i = 0; i < DP / 2; ++i) g2[l] = __builtin_elementwise_fma(dy2[i], w[i], g2[l]);
        });
        if (SIG)
            stream_rows<LP, DP, rows_per_block(DP)>(sW2 + G::W_WS, [&](int l, const f32x2(&w)[DP / 2]) {
#pragma unroll
                for (int i = 0; i < DP / 2; ++i) g2[l] = __builtin_elementwise_fma(dys2[i], w[i], g2[l]);
            });
#pragma unroll
        for (int l = 0; l < LP; ++l) {
            const float gl = g2[l][0] + g2[l][1];
            Tt[(G::B2 + l) * G::TS] = valid ? fmaf(mu[l], a.inv_bt, gl) : 0.f;       // dmu = g + mu/B
            Tt[(G::B2 + LP + l) * G::TS] = gl * z1r[l];                               // reparam part of d lv
        }
        // Each wave reads back only the 64 columns it wrote itself: LDS operations of one wave execute
        // in order, so a wave-level fence (no s_barrier) is all the hand-off needs.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        VAEK_STAMP(4);         // g, dmu done; image written

        // ---- phase 2: batch-reduction GEMMs on the f32 matrix cores ------------------------------
        // A[i = feature][k = sample], B[k = sample][j = feature]; lane l supplies (i|j = l&15, k = l>>4).
        // Fully unrolled and unmasked (see NF_PAD): all operand reads can be in flight ahead of the MFMAs.
        {
            const float* Tk = T + wave * 64 + (lane >> 4) + (lane & 15) * G::TS;
            constexpr int NOP = G::IB1 + G::JB1 + G::IB2 + G::JB2;   // operand reads per k-step
            constexpr int GS = 4, NG = 16 / GS;                      // k-steps per prefetch group
            float op[3][GS][NOP];
            auto load_group = [&](int g, int which) {
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    const int s4 = 4 * (g * GS + u);
                    int n = 0;
#pragma unroll
                    for (int i = 0; i < G::IB1; ++i) op[which][u][n++] = Tk[(G::A1 + 16 * i) * G::TS + s4];
#pragma unroll
                    for (int j = 0; j < G::JB1; ++j) op[which][u][n++] = Tk[(G::B1 + 16 * j) * G::TS + s4];
#pragma unroll
                    for (int i = 0; i < G::IB2; ++i) op[which][u][n++] = Tk[(G::A2 + 16 * i) * G::TS + s4];
#pragma unroll
                    for (int j = 0; j < G::JB2; ++j) op[which][u][n++] = Tk[(G::B2 + 16 * j) * G::TS + s4];
                }
            };
            load_group(0, 0);
            load_group(1, 1);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 2 < NG) load_group(g + 2, (g + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < GS; ++u) {
                    const float* o = op[g % 3][u];
#pragma unroll
                    for (int i = 0; i < G::IB1; ++i)
#pragma unroll
                        for (int j = 0; j < G::JB1; ++j)
                            acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[i], o[G::IB1 + j], acc1[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < G::IB2; ++i)
#pragma unroll
                        for (int j = 0; j < G::JB2; ++j)
                            acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[G::IB1 + G::JB1 + i], o[G::IB1 + G::JB1 + G::IB2 + j],
                                                                              acc2[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }

    VAEK_STAMP(5);             // MFMA loop done
    // ---- cross-wave sum through LDS, then one partial row per workgroup -----------------------
    // R[wave][blk][row = 4*(lane>>4) + r][col = lane&15]
    __syncthreads();           // every wave is done with its T columns before T is reused as R
    float* R = T;
    {
        const int col = lane & 15, row0 = 4 * (lane >> 4);
        int blk = 0;
#pragma unroll
        for (int i = 0; i < G::IB1; ++i)
#pragma unroll
            for (int j = 0; j < G::JB1; ++j, ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) R[((wave * G::NBLK + blk) * 16 + row0 + r) * 16 + col] = acc1[i][j][r];
#pragma unroll
        for (int i = 0; i < G::IB2; ++i)
#pragma unroll
            for (int j = 0; j < G::JB2; ++j, ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) R[((wave * G::NBLK + blk) * 16 + row0 + r) * 16 + col] = acc2[i][j][r];
    }
    s_mse = wsum(s_mse); s_deps = wsum(s_deps); s_musq = wsum(s_musq);
    if (lane == 0) { red[wave * 3 + 0] = s_mse; red[wave * 3 + 1] = s_musq; red[wave * 3 + 2] = s_deps; }
    __syncthreads();
    auto fetch = [&](int gemm, int i, int j) -> float {   // sum over waves of output (i, j) of GEMM 1 / 2
        const int blk = gemm == 1 ? (i >> 4) * G::JB1 + (j >> 4) : G::IB1 * G::JB1 + (i >> 4) * G::JB2 + (j >> 4);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < G::NW; ++w) v += R[((w * G::NBLK + blk) * 16 + (i & 15)) * 16 + (j & 15)];
        return v;
    };
    float* out = a.partials + (long long)blockIdx.x * a.pstride;
    for (int idx = t; idx < a.P + kExtra; idx += TILE) {
        float v = 0.f;
        if (idx < off_be) v = fetch(2, idx / L, idx % L);                             // dWe = x^T dmu
        else if (idx < off_wd) v = fetch(2, DP, idx - off_be);                        // dbe = 1^T dmu
        else if (idx < off_bd) { const int k = idx - off_wd; v = fetch(1, k / D, k % D); }   // dWd = samples^T dy
        else if (idx < off_bd + D) v = fetch(1, LP, idx - off_bd);                    // dbd = 1^T dy
        else if (SIG && idx < off_bs) { const int k = idx - off_ws; v = fetch(1, k / D, DP + k % D); }
        else if (SIG && idx < off_bs + D) v = fetch(1, LP, DP + idx - off_bs);
        else if (idx >= off_epsp && idx < off_epsp + L) v = fetch(2, DP, LP + idx - off_epsp);   // sum g*z1
        else if (idx >= a.P && idx < a.P + 3) {
#pragma unroll
            for (int w = 0; w < G::NW; ++w) v += red[w * 3 + (idx - a.P)];
        }
        out[idx] = v;
    }
    VAEK_STAMP(6);             // partial row written
    if (blockIdx.x == 0 && t == 0 && a.step_dev) a.step_dev[0] += 1;
}

// ---- finalize: sum of the partial rows (fixed order), closed-form terms, Adam -------------------
struct FusedFinArgs {
    const float* partials; int pstride; int G;
    int P, off_epsp, off_eps, L, D;
    const float* params; float eps_cli, rows_over_bt, inv_bt, rows;
    float* grads;
    float* params_rw; float* m; float* v; const int32_t* step_dev; float lr;
    CommDev comm;              // comm.world > 1: sum over ranks inside this kernel (epoch = Adam step)
    float* loss_hist; long long loss_hist_cap;    // optional: loss of Adam step t -> loss_hist[(t-1) % cap]
};


// block b covers outputs [n - 64(b+1), n - 64b): the LAST 64 (epsilon_p, epsilon and the three
// scalar sums, which need each other) always sit together in block 0.  1024 threads = 64 outputs x
// 16 row groups; each thread issues its (up to 16 x FIN_UNROLL) independent loads back to back.
constexpr int FIN_Q = 16;
__device__ __forceinline__ void fused_finalize_block(const FusedFinArgs& a) {
    __shared__ float part[FIN_Q][64];
    // every kernel argument the block uses, read once, up front (hipcc otherwise re-reads them from the kernarg segment
    // inside each branch, one scalar round trip at a time)
    const int P = a.P, G = a.G, L = a.L, D = a.D, pstride = a.pstride, off_epsp = a.off_epsp, off_eps = a.off_eps;
    const float eps_cli = a.eps_cli, rows_over_bt = a.rows_over_bt, inv_bt = a.inv_bt, rows = a.rows, lr = a.lr;
    const float* const partials = a.partials; const float* const params = a.params;
    float* const grads = a.grads; float* const params_rw = a.params_rw; float* const mp = a.m; float* const vp = a.v;
    const int32_t* const step_dev = a.step_dev;
    const int n = P + kExtra;
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int idx = n - 64 * ((int)blockIdx.x + 1) + o;
    // ---- ALL global loads first, UNCONDITIONAL (clamped indices, dummy sources for absent arrays), selects afterwards.
    // With `cond ? ptr[i] : 0` hipcc loads inside a branch and merges the value there -- an s_waitcnt per load site: this
    // kernel was four dependent memory round trips (Adam state, logvar_e, epsilon, then the partial rows) where one does.
    const int idc = idx > 0 ? idx : 0;                       // outputs below 0 do not exist (first block's low lanes)
    const int ipc = idc < P ? idc : P - 1;                    // parameter index for lanes that own a parameter
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int gq = q + u * FIN_Q;
        v[u] = partials[(long long)(gq < G ? gq : G - 1) * pstride + idc];
    }
    const float* const ps = params_rw ? params_rw : params;          // same memory; m / v absent in grads-only mode
    const float p_ld = ps[ipc], m_ld = (mp ? mp : params)[ipc], v_ld = (vp ? vp : params)[ipc];
    const int tstep = step_dev ? step_dev[0] : 0;             // uniform: a scalar load
    const float eps_ld = params[off_eps > 0 ? off_eps : 0];
    // ---- from here on: arithmetic on registers
    const bool adam = params_rw != nullptr && q == 0 && idx >= 0 && idx < P;
    float p_old = p_ld, m_old = m_ld, v_old = v_ld;
    const bool is_lv = q == 0 && idx >= off_epsp && idx < off_epsp + L;
    const float lv_own = is_lv ? p_ld : 0.f;
    const float eps_par = off_eps >= 0 ? eps_ld : 0.f;
    // closed-form KL constant sum_l (1 + lv - e^lv): every epsilon_p lane sits in wave 0 of block 0 (tail-aligned blocks), so
    // one wave reduction gives it to the loss lanes -- under the first barrier, instead of a chain of L dependent LDS reads
    // everything that depends only on the loaded state is worked out HERE, under the partial rows' latency, not on the tail
    // behind the reduction: e^lv, e^{lv/2}, the KL constant, Adam's bias corrections
    const float e_lv = expf(lv_own), e_hlv = expf(0.5f * lv_own);
    const float klc_w = q == 0 ? wsum(is_lv ? 1.f + lv_own - e_lv : 0.f) : 0.f;
    const float bc1 = -expm1f((float)tstep * -0.10536051565782628f);
    const float bc2 = -expm1f((float)tstep * -0.0010005003335835335f);
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (idx >= 0 && q + u * FIN_Q < G) ? v[u] : 0.f;
    for (int g0 = q + FIN_Q * 16; g0 < G; g0 += FIN_Q * 16) {            // more than 256 partial rows: further batches
        float w[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int gq = g0 + u * FIN_Q;
            w[u] = partials[(long long)(gq < G ? gq : G - 1) * pstride + idc];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (idx >= 0 && g0 + u * FIN_Q < G) ? w[u] : 0.f;
    }
    part[q][o] = acc;
    __syncthreads();
    if (q != 0) return;                // wave 0 finishes alone: nothing below needs another workgroup barrier
    float sacc = 0.f;
#pragma unroll
    for (int u = 0; u < FIN_Q; ++u) sacc += part[u][o];
    const bool live = idx >= 0;
    float g = live ? sacc : 0.f;
    // block 0 covers outputs [n - 64, n): the three scalar sums sit in lanes 60, 61, 62 of this wave -- v_readlane, not LDS
    const float s_mse = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sacc), 60));
    const float s_musq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sacc), 61));
    const float s_deps = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sacc), 62));
    if (!live) {
    } else if (idx >= off_epsp && idx < off_epsp + L) {
        g = 0.5f * e_hlv * g - 0.5f * (1.f - e_lv) * rows_over_bt;
    } else if (idx == off_eps) {
        g = eps_cli * (s_deps + 0.5f * rows * (float)D) * inv_bt;
    } else if (idx >= P) {
        if (idx < P + 3) {
            const float klc = klc_w;
            const float eps = off_eps >= 0 ? eps_par * eps_cli : eps_cli;
            const float dkl = (0.5f * s_musq - 0.5f * rows * klc) * inv_bt;
            const float mse = (s_mse + 0.5f * rows * (float)D * (kLog2Pi + eps)) * inv_bt;
            g = idx == P ? dkl + mse : (idx == P + 1 ? dkl : mse);
        } else {
            g = 0.f;
        }
    }
    // (no barrier needed here for "params read before written": every global load of this block was issued at the top and
    // has landed -- __syncthreads drains vmcnt -- before the first barrier; nothing below reads global memory)
    if (!live) return;
    if (a.comm.world > 1) g = comm_exchange_sum(a.comm, (unsigned)tstep, idx, g);   // xGMI, all ranks
    grads[idx] = g;
    if (idx == P && a.loss_hist) a.loss_hist[(long long)(tstep - 1) % a.loss_hist_cap] = g;
    if (adam) {
        adam_apply_f(p_old, g, m_old, v_old, lr, bc1, bc2);
        params_rw[idx] = p_old; mp[idx] = m_old; vp[idx] = v_old;
    }
}

__global__ __launch_bounds__(1024) void fused_finalize_kernel(const FusedFinArgs a) { fused_finalize_block(a); }

// vaek_train_step_gen: the finalize needs ceil((P+4)/64) blocks -- 9 of 256 CUs at the metric's size -- and is a
// chain of two dependent memory round trips.  The draw of the NEXT step's batch does not depend on the weights, so
// it rides in the same launch: blocks [0, nfin) finalize (dispatched first: they are the critical path), the rest
// are K7's work items, 1024 per block.  One launch, one stream: no cross-queue dependency anywhere.
// Measured (tools/time_step_gen.py): at B = 100 the step loses a whole launch (15.7 -> 12.7 us in the graph loop);
// at B = 65 536 the draw itself is ALU-bound (Philox + Box-Muller, ~3 us of issue over the whole chip) and the
// combined launch takes 10.9 us against 4.5 + 7.6 apart -- a wash there.  Tried and slower: a resident-sized
// generator grid walking contiguous chunks (13.4-14.4 us).
__global__ __launch_bounds__(1024) void fused_finalize_gen_kernel(const FusedFinArgs a, const BatchArgs b, const int nfin) {
    if ((int)blockIdx.x < nfin) { fused_finalize_block(a); return; }
    const unsigned step = make_batch_step(b);
    make_batch_items(b, step, (long long)(blockIdx.x - nfin) * 1024 + threadIdx.x);
    make_batch_advance(b, step, (int)blockIdx.x == nfin && threadIdx.x == 0);
}

// ---- host side ------------------------------------------------------------------------------------
typedef void (*FusedKernel)(const float*, const FusedArgs);
struct FusedVariant { int dp, lp, sig, tile, exact; FusedKernel fn; size_t lds_bytes; };

#define VAEK_FUSED(DP, LP, SIG, TILE, EXACT) \
    {DP, LP, SIG, TILE, EXACT, fused_linear_kernel<DP, LP, (SIG) != 0, TILE, (EXACT) != 0>, \
     sizeof(float) * FusedGeom<DP, LP, (SIG) != 0, TILE>::LDS_FLOATS}

static const FusedVariant kVariants[] = {
    // exact shapes of the linear-padding scripts (seed_linpadding_expts.sh): the metric's config first
    VAEK_FUSED(12, 20, 0, 256, 1),
#ifndef VAEK_FUSED_ONLY_M
    VAEK_FUSED(20, 20, 0, 256, 1), VAEK_FUSED(20, 10, 0, 256, 1),
    // zero-padded coverage for every other D, L <= 32
    VAEK_FUSED(8, 8, 0, 256, 0), VAEK_FUSED(12, 4, 0, 256, 0), VAEK_FUSED(16, 16, 0, 256, 0), VAEK_FUSED(24, 24, 0, 256, 0),
    VAEK_FUSED(32, 32, 0, 128, 0),
    // sigmoid dataset (two decoders), sigmoid_vae_padding_expts.sh shapes padded to even sizes
    VAEK_FUSED(8, 6, 1, 256, 0), VAEK_FUSED(12, 10, 1, 256, 0), VAEK_FUSED(16, 14, 1, 256, 0), VAEK_FUSED(18, 8, 1, 256, 0),
    VAEK_FUSED(22, 16, 1, 128, 1), VAEK_FUSED(28, 24, 1, 128, 1), VAEK_FUSED(32, 32, 1, 128, 0),
#endif
};

static const FusedVariant* pick_variant(const vaek_ctx* c) {
    if (c->cfg.n_enc_hidden != 0 || c->cfg.n_dec_hidden != 0 || c->cfg.dtype != VAEK_F32) return nullptr;
    const FusedVariant* best = nullptr;
    for (const auto& v : kVariants) {
        if (v.sig != (c->cfg.sigmoid_decoder ? 1 : 0) || v.dp < c->D || v.lp < c->L) continue;
        if (v.exact && (v.dp != c->D || v.lp != c->L)) continue;
        if (v.lds_bytes > 160 * 1024) continue;
        if (!best || v.dp * v.lp < best->dp * best->lp) best = &v;
    }
    return best;
}

// cfg.reserved[0]: 0 = matrix-core chain (fused_mfma.hip) where available, 1 = VALU chain (this file)
static bool use_mfma(const vaek_ctx* c) { return c->cfg.reserved[0] != 1 && fused_mfma_supported(c); }
static int fused_tile(const vaek_ctx* c, const FusedVariant* v) { return use_mfma(c) ? 256 : v->tile; }
static int fused_grid(const vaek_ctx* c, const FusedVariant* v) {
    const int tile = fused_tile(c, v);
    const int ntiles = (c->B + tile - 1) / tile;
    return std::max(1, std::min(ntiles, 2 * c->n_cu));
}
static int fused_pstride(const vaek_ctx* c) { return (int)((c->P + kExtra + 63) / 64 * 64); }

bool fused_supported(const vaek_ctx* c) { return pick_variant(c) != nullptr || mlp1_supported(c); }

size_t fused_workspace_bytes(const vaek_ctx* c) {
    const FusedVariant* v = pick_variant(c);
    if (!v) return mlp1_supported(c) ? (size_t)mlp1_grid(c) * fused_pstride(c) * sizeof(float) : 0;
    return (size_t)fused_grid(c, v) * fused_pstride(c) * sizeof(float);
}

int fused_train_step(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev,
                     const float* x, const float* z1, const float* z2, float lr, bool apply_adam, bool exchange,
                     void* ws, hipStream_t st, const BatchArgs* gen) {
    const FusedVariant* var = pick_variant(c);
    const bool mlp1 = !var && mlp1_supported(c);        // one-hidden-layer MLPs: fused_mlp1.hip writes the partial rows
    if (!var && !mlp1) { set_error("fused path not available for this configuration"); return VAEK_ERR_INVALID; }
    static thread_local const void* lds_set[sizeof(kVariants) / sizeof(kVariants[0])] = {};
    if (var) {
        const size_t vi = var - kVariants;
        if (var->lds_bytes > 64 * 1024 && lds_set[vi] == nullptr) {
            VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)var->fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)var->lds_bytes));
            lds_set[vi] = (const void*)var->fn;
        }
    }
    float* partials = reinterpret_cast<float*>(static_cast<char*>(ws) + c->ws_fused);
    const int grid = mlp1 ? mlp1_grid(c) : fused_grid(c, var), pstride = fused_pstride(c);
    FusedArgs a{};
    a.x = x; a.z1 = z1; a.z2 = z2; a.partials = partials; a.pstride = pstride;
    a.B = c->B; a.D = c->D; a.L = c->L; a.ntiles = mlp1 ? 0 : (c->B + fused_tile(c, var) - 1) / fused_tile(c, var);
    a.inv_bt = (float)(1.0 / (double)c->Bt); a.eps_cli = c->cfg.eps_cli;
    const int D = c->D, L = c->L;
    a.off_be = D * L; a.off_wd = a.off_be + L; a.off_bd = a.off_wd + L * D;
    a.off_ws = a.off_bd + D; a.off_bs = a.off_ws + L * D;
    a.off_epsp = (int)c->off_epsp; a.off_eps = (int)c->off_eps; a.P = (int)c->P;
    a.step_dev = step_dev;
    a.stamps = c->dbg_stamps;
    if (mlp1) {
        const int rc = mlp1_launch(c, params, x, z1, z2, partials, pstride, step_dev, st);
        if (rc) return rc;
    } else if (use_mfma(c) && grid == 1 && apply_adam && !exchange && c->cfg.world == 1) {
        // the whole batch is one workgroup's tile: ONE launch (fused_mfma.hip, single-launch step); with a batch to draw,
        // the generator's work items are workgroups 1.. of the same launch
        a.single = 1;
        a.grads = grads; a.params_rw = params; a.m = m; a.v = v; a.lr = lr;
        a.rows_over_bt = (float)((double)c->B / (double)c->Bt); a.rows = (float)c->B;
        a.loss_hist = c->loss_hist; a.loss_hist_cap = c->loss_hist_cap;
        int launch_grid = 1;
        if (gen) { a.has_gen = 1; a.gen = *gen; launch_grid += (int)((make_batch_item_count(*gen) + 255) / 256); }
        return fused_mfma_launch(c, params, &a, launch_grid, st);
    }
    if (mlp1) {
    } else if (use_mfma(c)) {
        int rc = fused_mfma_launch(c, params, &a, grid, st);
        if (rc) return rc;
    } else {
        ProfScope ps("fused_linear_fwd_bwd", st);
        launch_k(ps, var->fn, dim3(grid), dim3(var->tile), var->lds_bytes, st, (const float*)params, a);
        VAEK_HIP_CHECK(hipGetLastError());
    }
    FusedFinArgs f{};
    f.partials = partials; f.pstride = pstride; f.G = grid;
    f.P = (int)c->P; f.off_epsp = (int)c->off_epsp; f.off_eps = (int)c->off_eps; f.L = L; f.D = D;
    f.params = params; f.eps_cli = c->cfg.eps_cli;
    f.rows_over_bt = (float)((double)c->B / (double)c->Bt); f.inv_bt = a.inv_bt; f.rows = (float)c->B;
    f.grads = grads;
    f.params_rw = apply_adam ? params : nullptr; f.m = m; f.v = v; f.step_dev = step_dev; f.lr = lr;
    f.comm = CommDev{};
    if (exchange) f.comm = comm_dev(c, 0);
    f.loss_hist = (apply_adam && (exchange || c->cfg.world == 1)) ? c->loss_hist : nullptr; f.loss_hist_cap = c->loss_hist_cap;
    const int nfin = (int)((c->P + kExtra + 63) / 64);
    if (gen) {
        ProfScope ps("fused_finalize_adam_gen", st);
        const long long ngen = (make_batch_item_count(*gen) + 1023) / 1024;
        launch_k(ps, fused_finalize_gen_kernel, dim3((unsigned)(nfin + ngen)), dim3(1024), 0, st, f, *gen, nfin);
    } else {
        ProfScope ps(apply_adam ? "fused_finalize_adam" : "fused_finalize", st);
        launch_k(ps, fused_finalize_kernel, dim3((unsigned)nfin), dim3(1024), 0, st, f);
    }
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

}  // namespace vaek
