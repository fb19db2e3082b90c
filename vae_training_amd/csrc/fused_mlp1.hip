// Whole-network train-step kernel for VAEs whose encoder and decoder(s) have ONE hidden layer of at most 256 units
// (BASELINE config 2: sigmoid dataset, 7 -> 256 -> 6, two decoders 6 -> 256 -> 7): VAE.apply forward (networks.py:61-84 --
// encoder Dense/relu/Dense networks.py:26-44, reparameterisation :73-74, decoder(s) :75-80, decoder noise :81-83), the ELBO
// (:94-98) and the whole backward pass (what jax.value_and_grad returns at :99) in ONE launch; x, z1, z2 are read once,
// no activation ever reaches HBM, and each workgroup leaves one partial row of the flat gradient (+ the three scalar
// sums) for fused_finalize_kernel (fused_small.hip: fixed-order sum, closed-form KL terms, loss, optional P2P exchange,
// Adam) -- the same two-launch step as the linear models' fused path.  The layer-by-layer path took ~20 launches for
// this model (172 us per step at B = 8 192, 2 % of the f32 matrix peak).
//
// One workgroup = 256 threads = 4 waves, a tile of 32 samples at a time (tiles tile, tile + grid, ...; the gradient
// accumulates in registers across them).  Thread j IS hidden unit j of all three networks:
//   * its columns / rows of the six weight matrices live in its registers (loaded once, coalesced);
//   * the layers with a small inner dimension (x -> hidden, samples -> hidden, and every backward product through them)
//     are per-thread loops over the 32 samples with the small operand broadcast from LDS;
//   * the three hidden activations live in LDS as [unit][sample] (row stride 36 floats), later overwritten in place by
//     their gradients;
//   * the layers that REDUCE over the hidden units (hidden -> mu, hidden -> x_hat, d hidden -> d samples) run on
//     v_mfma_f32_16x16x4_f32 (exact f32) with features on the rows and samples on the columns: A = the thread-held
//     weights staged as [unit][feature] in a small LDS scratch, B = the activation image; wave w takes sample block
//     w & 1 and half (w >> 1) of the units, the two halves meet through LDS.
// Every sum over samples or units has a fixed order: bitwise run-to-run repeatable, like the rest of the library.
#include "vaek_internal.h"

namespace vaek {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int M1_TS = 32;            // samples per tile
constexpr int M1_NT = 256;           // threads = hidden units covered
constexpr int M1_HS = 36;            // row stride (floats) of the [unit][sample] activation images
constexpr int M1_FS = 16;            // row stride of the [sample][feature] images (features padded to 16)

struct Mlp1Args {
    const float* x; const float* z1; const float* z2; const float* params; float* partials;
    int pstride, B, D, L, He, Hd, ntiles;
    float inv_bt, eps_cli;
    int off_e1w, off_e1b, off_e2w, off_e2b, off_d1w, off_d1b, off_d2w, off_d2b, off_s1w, off_s1b, off_s2w, off_s2b;
    int off_epsp, off_eps, P;
    int32_t* step_dev;                       // the Adam step counter: advanced here, read by the finalize launch behind
    unsigned long long* stamps;              // -DVAEK_M1_STAMPS builds (tools/m1_stamps.sh): s_memrealtime at the phase boundaries
};

#ifdef VAEK_M1_STAMPS
#define M1_STAMP(i)                                                                                          \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        unsigned long long _t;                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); \
        if (a.stamps && blockIdx.x == 7 && threadIdx.x == 0) a.stamps[i] = _t;                               \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#else
#define M1_STAMP(i) do {} while (0)
#endif

struct Mlp1Lds {
    float X[M1_TS][M1_FS], Z1[M1_TS][M1_FS], Z2[M1_TS][M1_FS];
    float MU[M1_TS][M1_FS], SMP[M1_TS][M1_FS], XL[M1_TS][M1_FS], XS[M1_TS][M1_FS];
    float DY[M1_TS][M1_FS], DYS[M1_TS][M1_FS], DS[M1_TS][M1_FS], DMU[M1_TS][M1_FS];
    float WS[M1_NT][M1_FS];                  // MFMA A operand of the current pass: [unit][feature]
    float TMP[2][64][4];                     // the upper unit-half's accumulators on their way to the lower half's waves
    float RED[4][4];                         // per-wave scalar partials
    float HE[M1_NT][M1_HS], HD[M1_NT][M1_HS], HSg[M1_NT][M1_HS];
};

// out^T[feature][sample] (+)= sum_units W[unit][feature] * Himg[unit][sample] for this wave's sample block and unit half;
// the lower half's waves return the full sum for (sample = 16 b + lane % 16, features 4 g .. 4 g + 3), the others zeros.
// Callers put a barrier between staging WS / finishing Himg and this, and before re-staging WS.
__device__ __forceinline__ f32x4 m1_reduce_units(Mlp1Lds& s, const float (*Himg)[M1_HS], f32x4 acc, int wave, int lane) {
    const int b = wave & 1, kh = wave >> 1, n = lane & 15, kq = lane >> 4;
    // 32 k-steps of 4 units in chunks of 8; the operands of chunk c + 1 are read while chunk c is multiplied (hipcc left
    // alone issues ds_read, s_waitcnt lgkmcnt(0), v_mfma per step: ~2 us per pass instead of ~0.6); two accumulator chains
    float av[2][8], bv[2][8];
    auto fetch = [&](int set, int q0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int u = 4 * (q0 + i) + kq; av[set][i] = s.WS[u][n]; bv[set][i] = Himg[u][16 * b + n]; }
    };
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    fetch(0, 32 * kh);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c + 1 < 4) fetch((c + 1) & 1, 32 * kh + 8 * (c + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][i], bv[c & 1][i], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][i + 1], bv[c & 1][i + 1], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc + acc1;
}
__device__ __forceinline__ f32x4 m1_meet_halves(Mlp1Lds& s, f32x4 acc, int wave, int lane) {
    const int b = wave & 1, kh = wave >> 1;
    if (kh == 1) *reinterpret_cast<f32x4*>(s.TMP[b][lane]) = acc;
    __syncthreads();
    if (kh == 0) acc += *reinterpret_cast<const f32x4*>(s.TMP[b][lane]);
    return acc;
}

// rows s4 .. s4 + 3 of a [sample][16] image, N (8 or 16) features each, into registers: all reads issued back to back
template <int N>
__device__ __forceinline__ void m1_rows4(const float (*img)[M1_FS], int s4, float (&r)[4][N]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < N; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&img[s4 + u][c]);
            r[u][c] = v[0]; r[u][c + 1] = v[1]; r[u][c + 2] = v[2]; r[u][c + 3] = v[3];
        }
}

template <int DP, int LP, bool SIG>
__global__ __launch_bounds__(M1_NT) void mlp1_fused_kernel(const Mlp1Args a) {
    extern __shared__ __attribute__((aligned(16))) char m1_smem[];
    Mlp1Lds& s = *reinterpret_cast<Mlp1Lds*>(m1_smem);
    const int j = threadIdx.x, lane = j & 63, wave = __builtin_amdgcn_readfirstlane(j >> 6);
    const int D = a.D, L = a.L;
    const float* const P = a.params;
    const bool je = j < a.He, jd = j < a.Hd;
    if (blockIdx.x == 0 && j == 0 && a.step_dev) a.step_dev[0] += 1;
    M1_STAMP(0);
    // the inputs of a tile travel one tile ahead in registers (two elements of each [32][16] image per thread): the first tile's
    // loads share the weights' memory round trip, later ones land under the previous tile's arithmetic
    float xv[2], z1v[2], z2v[2];
    auto fetch_inputs = [&](int tile) {
        const int row0 = tile * M1_TS, valid = min(M1_TS, a.B - row0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {                      // unconditional at clamped indices, selects when they are written to LDS
            const int e = j + i * M1_NT, r = e / M1_FS, c = e % M1_FS;
            const long long row = row0 + min(r, valid - 1);
            xv[i] = a.x[row * a.D + min(c, a.D - 1)]; z1v[i] = a.z1[row * a.L + min(c, a.L - 1)]; z2v[i] = a.z2[row * a.D + min(c, a.D - 1)];
        }
    };
    fetch_inputs(min((int)blockIdx.x, a.ntiles - 1));

    // ---- this unit's weights (zero for units / features that do not exist: they then contribute nothing anywhere)
    // (every load unconditional at a clamped index, the selects afterwards: hipcc waits for a conditional load where it is issued,
    // which made this block a chain of ~45 memory round trips)
    float we1[DP], we2[LP], wd1[LP], wd2[DP], ws1[LP], ws2[DP];
    const int jce = min(j, a.He - 1), jcd = min(j, a.Hd - 1);
    const int s1w = SIG ? a.off_s1w : a.off_d1w, s2w = SIG ? a.off_s2w : a.off_d2w, s1b = SIG ? a.off_s1b : a.off_d1b,
              s2b = SIG ? a.off_s2b : a.off_d2b;
    float be1 = P[a.off_e1b + jce], bd1 = P[a.off_d1b + jcd], bs1 = P[s1b + jcd];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
        const int kc = min(k, D - 1);
        we1[k] = P[a.off_e1w + kc * a.He + jce]; wd2[k] = P[a.off_d2w + jcd * D + kc]; ws2[k] = P[s2w + jcd * D + kc];
    }
#pragma unroll
    for (int l = 0; l < LP; ++l) {
        const int lc = min(l, L - 1);
        we2[l] = P[a.off_e2w + jce * L + lc]; wd1[l] = P[a.off_d1w + lc * a.Hd + jcd]; ws1[l] = P[s1w + lc * a.Hd + jcd];
    }
    // small vectors every epilogue lane needs: biases of the reducing layers, e^{lv/2}, for features 4 g .. 4 g + 3
    const int f0 = 4 * (lane >> 4);
    float be2[4], bd2[4], bs2[4], shl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + r, fl = min(f, L - 1), fd = min(f, D - 1);
        be2[r] = P[a.off_e2b + fl]; bd2[r] = P[a.off_d2b + fd]; bs2[r] = P[s2b + fd]; shl[r] = P[a.off_epsp + fl];
    }
    const float eps_ld = P[a.off_eps >= 0 ? a.off_eps : 0];
    be1 = je ? be1 : 0.f; bd1 = jd ? bd1 : 0.f; bs1 = (SIG && jd) ? bs1 : 0.f;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
        we1[k] = (je && k < D) ? we1[k] : 0.f; wd2[k] = (jd && k < D) ? wd2[k] : 0.f; ws2[k] = (SIG && jd && k < D) ? ws2[k] : 0.f;
    }
#pragma unroll
    for (int l = 0; l < LP; ++l) {
        we2[l] = (je && l < L) ? we2[l] : 0.f; wd1[l] = (jd && l < L) ? wd1[l] : 0.f; ws1[l] = (SIG && jd && l < L) ? ws1[l] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = f0 + r;
        be2[r] = f < L ? be2[r] : 0.f; bd2[r] = f < D ? bd2[r] : 0.f; bs2[r] = (SIG && f < D) ? bs2[r] : 0.f;
        shl[r] = f < L ? expf(0.5f * shl[r]) : 0.f;
    }
    const float eps = a.off_eps >= 0 ? eps_ld * a.eps_cli : a.eps_cli;
    const float sigma = expf(0.5f * eps), inv_var = expf(-eps);

    M1_STAMP(1);
    // ---- gradient accumulators of this unit (registers, across all tiles of this workgroup)
    float g_we1[DP], g_we2[LP], g_wd1[LP], g_wd2[DP], g_ws1[LP], g_ws2[DP];
    float g_be1 = 0.f, g_bd1 = 0.f, g_bs1 = 0.f;
#pragma unroll
    for (int k = 0; k < DP; ++k) { g_we1[k] = 0.f; g_wd2[k] = 0.f; g_ws2[k] = 0.f; }
#pragma unroll
    for (int l = 0; l < LP; ++l) { g_we2[l] = 0.f; g_wd1[l] = 0.f; g_ws1[l] = 0.f; }
    // small outputs: thread t < 16 owns feature t of d be2 / d bd2 / d bs2 / the epsilon_p partial; scalar sums per thread
    float g_be2 = 0.f, g_bd2 = 0.f, g_bs2 = 0.f, g_epsp = 0.f;
    float p_mse = 0.f, p_musq = 0.f, p_deps = 0.f;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int row0 = tile * M1_TS, valid = min(M1_TS, a.B - row0);
        __syncthreads();                                   // the previous tile's images are no longer read
        // ---- inputs, zero-padded to [32][16]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = j + i * M1_NT, r = e / M1_FS, c = e % M1_FS;
            const bool in = r < valid;
            (&s.X[0][0])[e] = (in && c < D) ? xv[i] : 0.f;
            (&s.Z1[0][0])[e] = (in && c < L) ? z1v[i] : 0.f;
            (&s.Z2[0][0])[e] = (in && c < D) ? z2v[i] : 0.f;
        }
        if (tile + (int)gridDim.x < a.ntiles) fetch_inputs(tile + gridDim.x);
        __syncthreads();
        M1_STAMP(2);
        // ---- encoder layer 1: HE[j][s] = relu(x[s] . We1[:, j] + be1[j])
        // (per group of 4 samples the broadcast operand rows are read into registers first, then used: hipcc otherwise waits out
        // the LDS latency per product.  Tried and backed out: reading the NEXT group's rows under the current group's arithmetic
        // in all four per-unit loops -- the second register set spills, 3 to 11 KB of scratch, 240 us per step; and two threads
        // per unit (512 threads, two waves per SIMD, each half of the samples) -- 256 registers per thread no longer hold the
        // weights, the accumulators and the staged rows: 47 us per step against 33.)
#pragma unroll 2
        for (int s4 = 0; s4 < M1_TS; s4 += 4) {
            float xr[4][DP];
            m1_rows4<DP>(s.X, s4, xr);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 h;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float acc = be1;
#pragma unroll
                for (int k = 0; k < DP; ++k) acc = fmaf(xr[u][k], we1[k], acc);
                h[u] = fmaxf(acc, 0.f);
            }
            *reinterpret_cast<f32x4*>(&s.HE[j][s4]) = h;
        }
#pragma unroll
        for (int l = 0; l < M1_FS; ++l) s.WS[j][l] = l < LP ? we2[l < LP ? l : 0] : 0.f;
        __syncthreads();
        M1_STAMP(3);
        // ---- mu = HE^T We2 + be2; samples = mu + e^{lv/2} z1
        {
            f32x4 acc = m1_reduce_units(s, s.HE, f32x4{0.f, 0.f, 0.f, 0.f}, wave, lane);
            acc = m1_meet_halves(s, acc, wave, lane);
            if ((wave >> 1) == 0) {
                const int smp = 16 * (wave & 1) + (lane & 15);
                const f32x4 z = *reinterpret_cast<const f32x4*>(&s.Z1[smp][f0]);
                f32x4 mu, sm;
#pragma unroll
                for (int r = 0; r < 4; ++r) { mu[r] = acc[r] + be2[r]; sm[r] = fmaf(shl[r], z[r], mu[r]); }
                *reinterpret_cast<f32x4*>(&s.MU[smp][f0]) = mu;
                *reinterpret_cast<f32x4*>(&s.SMP[smp][f0]) = sm;
                if (smp < valid) p_musq += (mu[0] * mu[0] + mu[1] * mu[1]) + (mu[2] * mu[2] + mu[3] * mu[3]);
            }
        }
        __syncthreads();
        M1_STAMP(4);
        // ---- decoder(s) layer 1
#pragma unroll 2
        for (int s4 = 0; s4 < M1_TS; s4 += 4) {
            float sr[4][LP];
            m1_rows4<LP>(s.SMP, s4, sr);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 h, hs;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float acc = bd1, accs = bs1;
#pragma unroll
                for (int l = 0; l < LP; ++l) {
                    const float sv = sr[u][l];
                    acc = fmaf(sv, wd1[l], acc);
                    if (SIG) accs = fmaf(sv, ws1[l], accs);
                }
                h[u] = fmaxf(acc, 0.f); hs[u] = fmaxf(accs, 0.f);
            }
            *reinterpret_cast<f32x4*>(&s.HD[j][s4]) = h;
            if (SIG) *reinterpret_cast<f32x4*>(&s.HSg[j][s4]) = hs;
        }
#pragma unroll
        for (int k = 0; k < M1_FS; ++k) s.WS[j][k] = k < DP ? wd2[k < DP ? k : 0] : 0.f;
        __syncthreads();
        M1_STAMP(5);
        // ---- x_hat (linear decoder) = HD^T Wd2 + bd2
        {
            f32x4 acc = m1_reduce_units(s, s.HD, f32x4{0.f, 0.f, 0.f, 0.f}, wave, lane);
            acc = m1_meet_halves(s, acc, wave, lane);
            if ((wave >> 1) == 0) {
                const int smp = 16 * (wave & 1) + (lane & 15);
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[r] + bd2[r];
                *reinterpret_cast<f32x4*>(&s.XL[smp][f0]) = o;
            }
        }
        if (SIG) {
            __syncthreads();                               // every wave is done with WS
#pragma unroll
            for (int k = 0; k < M1_FS; ++k) s.WS[j][k] = k < DP ? ws2[k < DP ? k : 0] : 0.f;
            __syncthreads();
            f32x4 acc = m1_reduce_units(s, s.HSg, f32x4{0.f, 0.f, 0.f, 0.f}, wave, lane);
            acc = m1_meet_halves(s, acc, wave, lane);
            if ((wave >> 1) == 0) {
                const int smp = 16 * (wave & 1) + (lane & 15);
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = 1.f / (1.f + expf(-(acc[r] + bs2[r])));
                *reinterpret_cast<f32x4*>(&s.XS[smp][f0]) = o;
            }
        }
        __syncthreads();
        M1_STAMP(6);
        // ---- ELBO, elementwise over [32][16]: decoder noise, residual, dL/dx_hat, the sigmoid head's gradient, scalar sums
        for (int e = j; e < M1_TS * M1_FS; e += M1_NT) {
            const int r = e / M1_FS, c = e % M1_FS;
            const bool in = r < valid && c < D;
            const float sg = SIG ? (&s.XS[0][0])[e] : 0.f;
            const float z2v = (&s.Z2[0][0])[e];
            const float res = (&s.XL[0][0])[e] + sg + z2v * sigma - (&s.X[0][0])[e];
            const float dy = in ? res * inv_var * a.inv_bt : 0.f;
            (&s.DY[0][0])[e] = dy;
            if (SIG) (&s.DYS[0][0])[e] = dy * sg * (1.f - sg);
            if (in) {
                const float q = 0.5f * res * res * inv_var;
                p_mse += q;
                p_deps += -q + 0.5f * sigma * z2v * res * inv_var;
            }
        }
        __syncthreads();
        M1_STAMP(7);
        // ---- decoder(s) backward through layer 2 and the relu; the activation images become gradient images
#pragma unroll 2
        for (int s4 = 0; s4 < M1_TS; s4 += 4) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(&s.HD[j][s4]);
            f32x4 hs = {0.f, 0.f, 0.f, 0.f}, dh, dhs;
            if (SIG) hs = *reinterpret_cast<const f32x4*>(&s.HSg[j][s4]);
            float dyr[4][DP], dsr[4][SIG ? DP : 4], sr[4][LP];
            m1_rows4<DP>(s.DY, s4, dyr);
            if (SIG) m1_rows4<SIG ? DP : 4>(s.DYS, s4, dsr);
            m1_rows4<LP>(s.SMP, s4, sr);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float d1 = 0.f, d2 = 0.f;
#pragma unroll
                for (int k = 0; k < DP; ++k) {
                    const float dyv = dyr[u][k];
                    g_wd2[k] = fmaf(h[u], dyv, g_wd2[k]);
                    d1 = fmaf(dyv, wd2[k], d1);
                    if (SIG) {
                        const float dsv = dsr[u][SIG ? k : 0];
                        g_ws2[k] = fmaf(hs[u], dsv, g_ws2[k]);
                        d2 = fmaf(dsv, ws2[k], d2);
                    }
                }
                d1 = h[u] > 0.f ? d1 : 0.f;
                d2 = hs[u] > 0.f ? d2 : 0.f;
                g_bd1 += d1; g_bs1 += d2;
#pragma unroll
                for (int l = 0; l < LP; ++l) {
                    const float sv = sr[u][l];
                    g_wd1[l] = fmaf(sv, d1, g_wd1[l]);
                    if (SIG) g_ws1[l] = fmaf(sv, d2, g_ws1[l]);
                }
                dh[u] = d1; dhs[u] = d2;
            }
            *reinterpret_cast<f32x4*>(&s.HD[j][s4]) = dh;
            if (SIG) *reinterpret_cast<f32x4*>(&s.HSg[j][s4]) = dhs;
        }
#pragma unroll
        for (int l = 0; l < M1_FS; ++l) s.WS[j][l] = l < LP ? wd1[l < LP ? l : 0] : 0.f;
        __syncthreads();
        M1_STAMP(8);
        // ---- d samples = dHD^T Wd1^T (+ dHS^T Ws1^T); d mu = d samples + mu / Bt
        {
            f32x4 acc = m1_reduce_units(s, s.HD, f32x4{0.f, 0.f, 0.f, 0.f}, wave, lane);
            if (SIG) {
                __syncthreads();
#pragma unroll
                for (int l = 0; l < M1_FS; ++l) s.WS[j][l] = l < LP ? ws1[l < LP ? l : 0] : 0.f;
                __syncthreads();
                acc = m1_reduce_units(s, s.HSg, acc, wave, lane);
            }
            acc = m1_meet_halves(s, acc, wave, lane);
            if ((wave >> 1) == 0) {
                const int smp = 16 * (wave & 1) + (lane & 15);
                const f32x4 mu = *reinterpret_cast<const f32x4*>(&s.MU[smp][f0]);
                f32x4 dmu;
#pragma unroll
                for (int r = 0; r < 4; ++r) dmu[r] = smp < valid ? fmaf(mu[r], a.inv_bt, acc[r]) : 0.f;
                *reinterpret_cast<f32x4*>(&s.DS[smp][f0]) = acc;
                *reinterpret_cast<f32x4*>(&s.DMU[smp][f0]) = dmu;
            }
        }
        __syncthreads();
        M1_STAMP(9);
        // ---- encoder backward
#pragma unroll 2
        for (int s4 = 0; s4 < M1_TS; s4 += 4) {
            const f32x4 h = *reinterpret_cast<const f32x4*>(&s.HE[j][s4]);
            float dmr[4][LP], xr[4][DP];
            m1_rows4<LP>(s.DMU, s4, dmr);
            m1_rows4<DP>(s.X, s4, xr);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float d1 = 0.f;
#pragma unroll
                for (int l = 0; l < LP; ++l) {
                    const float dm = dmr[u][l];
                    g_we2[l] = fmaf(h[u], dm, g_we2[l]);
                    d1 = fmaf(dm, we2[l], d1);
                }
                d1 = h[u] > 0.f ? d1 : 0.f;
                g_be1 += d1;
#pragma unroll
                for (int k = 0; k < DP; ++k) g_we1[k] = fmaf(xr[u][k], d1, g_we1[k]);
            }
        }
        M1_STAMP(10);
        // ---- the small vectors: thread t < 16 sums feature t over the samples
        if (j < M1_FS) {
            for (int r = 0; r < M1_TS; ++r) {
                g_be2 += s.DMU[r][j]; g_bd2 += s.DY[r][j];
                if (SIG) g_bs2 += s.DYS[r][j];
                g_epsp = fmaf(s.DS[r][j], s.Z1[r][j], g_epsp);
            }
        }
    }

    M1_STAMP(11);
    // ---- this workgroup's partial row
    float* const row = a.partials + (long long)blockIdx.x * a.pstride;
    if (je) {
        row[a.off_e1b + j] = g_be1;
#pragma unroll
        for (int k = 0; k < DP; ++k) if (k < D) row[a.off_e1w + k * a.He + j] = g_we1[k];
#pragma unroll
        for (int l = 0; l < LP; ++l) if (l < L) row[a.off_e2w + j * L + l] = g_we2[l];
    }
    if (jd) {
        row[a.off_d1b + j] = g_bd1;
#pragma unroll
        for (int l = 0; l < LP; ++l) if (l < L) row[a.off_d1w + l * a.Hd + j] = g_wd1[l];
#pragma unroll
        for (int k = 0; k < DP; ++k) if (k < D) row[a.off_d2w + j * D + k] = g_wd2[k];
        if (SIG) {
            row[a.off_s1b + j] = g_bs1;
#pragma unroll
            for (int l = 0; l < LP; ++l) if (l < L) row[a.off_s1w + l * a.Hd + j] = g_ws1[l];
#pragma unroll
            for (int k = 0; k < DP; ++k) if (k < D) row[a.off_s2w + j * D + k] = g_ws2[k];
        }
    }
    if (j < L) { row[a.off_e2b + j] = g_be2; row[a.off_epsp + j] = g_epsp; }
    if (j < D) { row[a.off_d2b + j] = g_bd2; if (SIG) row[a.off_s2b + j] = g_bs2; }
    // the three scalar sums: lanes by xor-shuffle, the four waves in order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        p_mse += __shfl_xor(p_mse, o, 64); p_musq += __shfl_xor(p_musq, o, 64); p_deps += __shfl_xor(p_deps, o, 64);
    }
    __syncthreads();
    if (lane == 0) { s.RED[wave][0] = p_mse; s.RED[wave][1] = p_musq; s.RED[wave][2] = p_deps; }
    __syncthreads();
    if (j < 3) row[a.P + j] = (s.RED[0][j] + s.RED[1][j]) + (s.RED[2][j] + s.RED[3][j]);
    if (j == 3) { row[a.P + 3] = 0.f; if (a.off_eps >= 0) row[a.off_eps] = 0.f; }
    M1_STAMP(12);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
bool mlp1_supported(const vaek_ctx* c) {
    return c->cfg.n_enc_hidden == 1 && c->cfg.n_dec_hidden == 1 && c->cfg.dtype == VAEK_F32 && c->D <= 16 && c->L <= 16 &&
           c->cfg.enc_hidden[0] <= M1_NT && c->cfg.dec_hidden[0] <= M1_NT && c->cfg.enc_hidden[0] >= 1 && c->cfg.dec_hidden[0] >= 1;
}
int mlp1_grid(const vaek_ctx* c) {
    const int ntiles = (c->B + M1_TS - 1) / M1_TS;
    return std::max(1, std::min(ntiles, c->n_cu));          // one workgroup per CU (its LDS image is ~135 KB)
}

typedef void (*Mlp1Kernel)(const Mlp1Args);
int mlp1_launch(vaek_ctx* c, const float* params, const float* x, const float* z1, const float* z2, float* partials, int pstride,
                int32_t* step_dev, hipStream_t st) {
    Mlp1Args a{};
    a.step_dev = step_dev; a.stamps = c->dbg_stamps;
    a.x = x; a.z1 = z1; a.z2 = z2; a.params = params; a.partials = partials; a.pstride = pstride;
    a.B = c->B; a.D = c->D; a.L = c->L; a.He = c->cfg.enc_hidden[0]; a.Hd = c->cfg.dec_hidden[0];
    a.ntiles = (c->B + M1_TS - 1) / M1_TS;
    a.inv_bt = (float)(1.0 / (double)c->Bt); a.eps_cli = c->cfg.eps_cli;
    const int D = c->D, L = c->L, He = a.He, Hd = a.Hd;
    int off = 0;
    a.off_e1w = off; off += D * He; a.off_e1b = off; off += He; a.off_e2w = off; off += He * L; a.off_e2b = off; off += L;
    a.off_d1w = off; off += L * Hd; a.off_d1b = off; off += Hd; a.off_d2w = off; off += Hd * D; a.off_d2b = off; off += D;
    if (c->cfg.sigmoid_decoder) {
        a.off_s1w = off; off += L * Hd; a.off_s1b = off; off += Hd; a.off_s2w = off; off += Hd * D; a.off_s2b = off; off += D;
    }
    if (off != (int)c->off_epsp) { set_error("mlp1: parameter layout mismatch"); return VAEK_ERR_INVALID; }
    a.off_epsp = (int)c->off_epsp; a.off_eps = (int)c->off_eps; a.P = (int)c->P;
    const bool big = D > 8 || L > 8, sig = c->cfg.sigmoid_decoder != 0;
    const Mlp1Kernel fn = big ? (sig ? mlp1_fused_kernel<16, 16, true> : mlp1_fused_kernel<16, 16, false>)
                              : (sig ? mlp1_fused_kernel<8, 8, true> : mlp1_fused_kernel<8, 8, false>);
    static thread_local PerDeviceOnce attr_set[4];
    const int vi = (big ? 2 : 0) + (sig ? 1 : 0);
    if (attr_set[vi].need()) {
        VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Mlp1Lds)));
        attr_set[vi].mark();
    }
    ProfScope ps("fused_mlp1_fwd_bwd", st);
    launch_k(ps, fn, dim3((unsigned)mlp1_grid(c)), dim3(M1_NT), sizeof(Mlp1Lds), st, a);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

}  // namespace vaek
