// N consecutive train steps of a LINEAR VAE (encoder = one Dense D -> L, decoder = one Dense L -> D: every "" layer-size
// experiment of seed_linpadding_expts.sh, the configuration the headline metric is quoted on), software-pipelined over
// launches: vaek_train_steps (include/vaek.h).  Same function as N calls of VAE.train_step (networks.py:87-101: forward,
// ELBO, value_and_grad, Adam), evaluated through its sufficient statistic.
//
// Why this is exact.  With no non-linearity between input and loss, every per-sample quantity is LINEAR in
//     u_b = [z1_b (L) | x_b (D) | z2_b (D) | 1]                                   (NF = L + 2 D + 1 features):
//     mu_b = E u_b,        E = [0 | We^T | 0 | be]                                  (networks.py:67-68)
//     samples_b = S u_b,   S = E + [diag(e^{lv/2}) | 0 | 0 | 0]                     (networks.py:73-74)
//     r_b = x_hat_b - x_b = R u_b,   R = [Wd^T diag(e^{lv/2}) | Wd^T We^T - I | e^{eps/2} I | Wd^T be + bd]   (:80-83)
// and the loss and every gradient of SURVEY.md 8(a) row a5 are quadratic in u_b summed over the batch, i.e. functions of
//     M = sum_b u_b u_b^T                                                           (NF x NF, parameter-INDEPENDENT)
// and of the parameters alone:  sum_b |r_b|^2 = tr(R M R^T),  sum_b |mu_b|^2 = tr(E M E^T),  dWd = c0 S M R^T,
// [dWe | dbe] = columns of c0 Wd R M + E M / B,  d lv_l = 1/2 e^{lv_l/2} c0 (Wd R M)[l, z1_l] - 1/2 (1 - e^{lv_l}),
// d eps = (-1/2 tr(RMR^T) e^{-eps} + 1/2 B D + 1/2 e^{-eps/2} sum_d (R M)[d, z2_d]) / B,  c0 = e^{-eps} / B.
//
// What it buys.  The streaming pass over the batch (11.5 MB at the metric's size, the whole algorithmic traffic) no longer
// depends on the parameters, so nothing couples the 256 workgroups of a step to the previous step's all-reduce -> Adam ->
// broadcast: the two-kernel step (fused_mfma.hip + fused_finalize) spends most of its 13 us waiting on exactly that chain.
// Here launch n carries three roles at once:
//     streamers   (one per 256-sample tile)  batch n:   x, z1, z2 tiles land in LDS by global_load_lds, then
//                 M_tile = U^T U on v_mfma_f32_16x16x4_f32 (exact f32 fmaf chains; upper block triangle only) -> one
//                 partial image per workgroup.  No weights, no elementwise pass, no transposition (the k axis of the MFMA
//                 is the sample axis and any sample order will do: operands are read straight from the row-major images).
//     reducers    batch n-1: fixed-order float64 sum of the partial images -> M.
//     updater     (one workgroup) batch n-2: the small dense algebra above in float64, the closed-form KL terms, the three
//                 loss means, Adam (flax.optim.Adam.apply_gradient, networks.py:100), the step counter, the loss ring.
// Stream order is the only synchronisation (no flags, no spins, nothing to dead-lock); the updater chain is the only thing
// that is sequential in the parameters, and it is one workgroup's few microseconds.
//
// Numerics: M accumulates exact-f32 products in chains of 64 samples (one wave's 16 MFMA k-steps), summed further in
// float64; everything downstream is float64 until the final rounding of each gradient to float32.  Against the float64
// oracle the loss sits at ~1e-7 relative (tests/test_gpu_steps.py), like the sample-by-sample kernels -- but it is a
// different summation order, so this path is NOT bitwise comparable with vaek_train_step.
#include <stdlib.h>

#include "vaek_internal.h"

namespace vaek {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

struct LinArgs {
    // roles by blockIdx.x: [0, has_update) the updater, then n_reduce reducers, then n_stream streamers
    int has_update, n_reduce, n_stream;
    int B, D, L, ntiles;
    // streamers: the batch of this launch
    const float* x; const float* z1; const float* z2; float* partial_out;        // [ntiles][NBLK * 256]
    // reducers: the batch of the previous launch
    const float* partial_in; double* M_out;                                       // [NBLK * 256]
    // updater: the batch before that
    const double* M_in;
    float* params; float* grads; float* m; float* v; int32_t* step_dev; float lr;
    float inv_bt, eps_cli, rows, rows_over_bt; int off_eps, P;
    float* loss_hist; long long loss_hist_cap;
};

#ifdef VAEK_LIN_STAMPS      // diagnostic build (tools/lin_stamps.sh): s_memtime at the updater's phase boundaries, into a buffer nothing reads
__device__ unsigned long long* g_lin_stamp_buf = nullptr;
#define LIN_STAMP(i)                                                                                         \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        unsigned long long _t;                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); \
        if (g_lin_stamp_buf && threadIdx.x == 0) g_lin_stamp_buf[i] = _t;                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#else
#define LIN_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ void lin_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

// block (b1, b2), b1 <= b2, of the upper block triangle -> its index in the packed image
__device__ __host__ constexpr int lin_blk(int NB, int b1, int b2) { return b1 * NB - b1 * (b1 - 1) / 2 + (b2 - b1); }

constexpr int LNT = 1024, LNW = LNT / 64;          // threads / waves per workgroup, every role: the updater wants the latency hiding

// ---- streamer: M_tile = U^T U over this workgroup's 256 samples (16 per wave) ---------------------------------------------
template <int NB>
__device__ __forceinline__ void lin_stream(const LinArgs& a, char* smem, int tile) {
    constexpr int NBLK = NB * (NB + 1) / 2;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int D = a.D, L = a.L;
    // LDS: the three tiles exactly as they lie in HBM (row-major [256][L] / [256][D]), each padded to whole 4 KB, then the
    // validity column V[256] (the "1" feature; 0 for rows past the batch end) and a zero word
    const int z1_bytes = (L * 1024 + 4095) / 4096 * 4096, x_bytes = (D * 1024 + 4095) / 4096 * 4096;
    const int oX = z1_bytes, oZ2 = oX + x_bytes, oV = oZ2 + x_bytes, oC = oV + 1024;
    const long long row0 = (long long)tile * 256;
    auto copy = [&](const float* src, int cols, int bytes, int lds_off) {
        const long long tot = (long long)a.B * cols * 4, base = row0 * cols * 4;
        for (int i = 0; i * (LNT * 16) < bytes; ++i) {
            const int local = i * (LNT * 16) + t * 16;           // a wave's 1 KB lies inside or outside the 4 KB-padded image as a whole
            if (local >= bytes) continue;
            long long off = base + local;
            // a 16-byte piece that would run past the tensor's last whole 16 bytes: a valid, aligned address instead; what it
            // brings lands in rows that are zeroed below -- or in the tensor's last <= 3 floats, which patch_tail() rewrites
            off = off + 16 <= (tot & ~15ll) ? off : ((tot & ~15ll) - 16);
            lin_glds16(reinterpret_cast<const char*>(src) + off, smem + lds_off + i * (LNT * 16) + wave * 1024);
        }
    };
    // the floats behind the tensor's last whole 16 bytes (B * cols * 4 not a multiple of 16: e.g. L = 2 with an odd batch)
    auto patch_tail = [&](const float* src, int cols, int lds_off) {
        const long long tot = (long long)a.B * cols * 4, full = tot & ~15ll, base = row0 * cols * 4;
        if (t < (int)((tot - full) / 4) && full >= base) reinterpret_cast<float*>(smem + lds_off)[(full - base) / 4 + t] = src[full / 4 + t];
    };
    copy(a.z1, L, z1_bytes, 0);
    copy(a.x, D, x_bytes, oX);
    copy(a.z2, D, x_bytes, oZ2);
    const int valid = (int)min(256ll, (long long)a.B - row0);
    if (t < 256) reinterpret_cast<float*>(smem + oV)[t] = t < valid ? 1.f : 0.f;
    if (t == 0) *reinterpret_cast<float*>(smem + oC) = 0.f;
    // per block: where this lane's feature 16 b + (lane & 15) lives (byte address of sample 0, byte stride per sample)
    int fbase[NB], fstride[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int f = 16 * b + (lane & 15);
        if (f < L) { fbase[b] = f * 4; fstride[b] = L * 4; }
        else if (f < L + D) { fbase[b] = oX + (f - L) * 4; fstride[b] = D * 4; }
        else if (f < L + 2 * D) { fbase[b] = oZ2 + (f - L - D) * 4; fstride[b] = D * 4; }
        else if (f == L + 2 * D) { fbase[b] = oV; fstride[b] = 4; }
        else { fbase[b] = oC; fstride[b] = 0; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (valid < 256) {                                   // last tile of a ragged batch: rows past the end contribute nothing
        patch_tail(a.z1, L, 0); patch_tail(a.x, D, oX); patch_tail(a.z2, D, oZ2);
        for (int e = valid * L + t; e < 256 * L; e += LNT) reinterpret_cast<float*>(smem)[e] = 0.f;
        for (int e = valid * D + t; e < 256 * D; e += LNT) { reinterpret_cast<float*>(smem + oX)[e] = 0.f; reinterpret_cast<float*>(smem + oZ2)[e] = 0.f; }
        __syncthreads();
    }
    f32x4 acc[NBLK];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // k-step s: lane group g = lane >> 4 takes sample 16 wave + s + 4 g.  The four samples of a step lie 4 rows apart: with 80- and
    // 48-byte rows that is 16 banks, so the two lane groups ds_read_b32 services together never collide.
    const int g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int sample = wave * 16 + s + 4 * g;
        float op[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) op[b] = *reinterpret_cast<const float*>(smem + fbase[b] + sample * fstride[b]);
#pragma unroll
        for (int b1 = 0; b1 < NB; ++b1)
#pragma unroll
            for (int b2 = b1; b2 < NB; ++b2)
                acc[lin_blk(NB, b1, b2)] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[b1], op[b2], acc[lin_blk(NB, b1, b2)], 0, 0, 0);
    }
    __syncthreads();                                     // the input images are dead: reuse them for the waves' block images
    float* R = reinterpret_cast<float*>(smem);
    constexpr int NWR = NBLK <= 6 ? LNW : LNW / 2;         // images that fit LDS at once (16 x 6 KB = 96 KB; NB = 4 folds once first)
    if (NWR < LNW) {
        if (wave >= NWR) {
#pragma unroll
            for (int k = 0; k < NBLK; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) R[((wave - NWR) * NBLK + k) * 256 + (4 * g + r) * 16 + (lane & 15)] = acc[k][r];
        }
        __syncthreads();
        if (wave < NWR) {
#pragma unroll
            for (int k = 0; k < NBLK; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[k][r] += R[(wave * NBLK + k) * 256 + (4 * g + r) * 16 + (lane & 15)];
        }
        __syncthreads();
    }
    if (wave < NWR) {
#pragma unroll
        for (int k = 0; k < NBLK; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) R[(wave * NBLK + k) * 256 + (4 * g + r) * 16 + (lane & 15)] = acc[k][r];   // row 4 g + r, column lane & 15
    }
    __syncthreads();
    float* out = a.partial_out + (long long)tile * (NBLK * 256);
    for (int o = t; o < NBLK * 256; o += LNT) {          // the waves' images in wave order
        float sum = R[o];
#pragma unroll
        for (int w = 1; w < NWR; ++w) sum += R[w * NBLK * 256 + o];
        out[o] = sum;
    }
}

// ---- reducer: 32 outputs x 32 row groups per workgroup, float64, fixed order ------------------------------------------------
__device__ __forceinline__ void lin_reduce(const LinArgs& a, char* smem, int rb, int no) {
    double* sums = reinterpret_cast<double*>(smem);       // [32][32]
    const int t = threadIdx.x, o = rb * 32 + (t & 31), rg = t >> 5;
    const int rpg = (a.ntiles + 31) / 32, r_lo = rg * rpg, r_hi = min(a.ntiles, r_lo + rpg);
    double s = 0.0;
    if (o < no)
        for (int r0 = r_lo; r0 < r_hi; r0 += 8) {
            float tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tv[u] = a.partial_in[(long long)min(r0 + u, r_hi - 1) * no + o];     // unconditional, clamped
#pragma unroll
            for (int u = 0; u < 8; ++u) s += r0 + u < r_hi ? (double)tv[u] : 0.0;
        }
    sums[rg * 32 + (t & 31)] = s;
    __syncthreads();
    if (t < 32 && o < no) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) tot += sums[k * 32 + t];
        a.M_out[o] = tot;
    }
}

// ---- updater: gradients and loss from (M, parameters), Adam -----------------------------------------------------------------
// R and E are never formed.  With S = E + [diag(s) | 0 | 0 | 0] (samples = S u) and r = Wd^T samples + bd - x + sigma z2:
//     SM = S M                      (L x NF; S has D + 2 non-zeros per row)          Q = E M = SM - diag(s) M[z1 rows]
//     P1 = R M = Wd^T SM - M[x rows] + sigma M[z2 rows] + bd M[one row]              (D x NF)
//     G  = Wd P1                    (L x NF)          dwd[l][d] = S[l,:] . P1[d,:]   (= sum_b samples_bl r_bd)
//     sum_b |r_b|^2 = tr(R M R^T) = sum_{l,d} Wd[l][d] dwd[l][d] + sum_d (-P1[d][x_d] + sigma P1[d][z2_d] + bd_d P1[d][one])
//     sum_b |mu_b|^2 = tr(E M E^T) = sum_l (sum_dd We[dd][l] Q[l][x_dd] + be_l Q[l][one])
// DT, LT > 0: the dimensions at compile time (the metric's 12 / 20: fully unrolled inner products); 0: run-time.
template <int NB, int DT, int LT>
__device__ __forceinline__ void lin_update(const LinArgs& a, char* smem) {
    constexpr int NFP = 16 * NB, NBLK = NB * (NB + 1) / 2;
    const int t = threadIdx.x, D = DT ? DT : a.D, L = LT ? LT : a.L, P = a.P;
    const int fone = L + 2 * D;
    const int off_be = D * L, off_wd = off_be + L, off_bd = off_wd + L * D, off_epsp = off_bd + D;
    // LDS carve, all float64
    double* Mf = reinterpret_cast<double*>(smem);         // [NFP][NFP] symmetric
    double* SM = Mf + NFP * NFP;                           // [L][NFP]
    double* P1 = SM + L * NFP;                             // [D][NFP]
    double* G = P1 + D * NFP;                              // [L][NFP]
    double* Wed = G + L * NFP;                             // [D][L]   encoder kernel (flax [in, out])
    double* Wdd = Wed + D * L;                             // [L][D]   decoder kernel
    double* bed = Wdd + L * D;                             // [L]
    double* bdd = bed + L;                                 // [D]
    double* sd = bdd + D;                                  // [L] e^{lv/2}
    double* elv = sd + L;                                  // [L] e^{lv}
    double* dwd = elv + L;                                 // [L][D]
    double* red = dwd + L * D;                             // [4 sums][16 waves]
    LIN_STAMP(0);
    const int tstep = a.step_dev[0] + 1;
    // this thread's outputs idx = t + 256 k: parameter and Adam state now, used at the very end (their latency is free here)
    constexpr int KOUT = 2;                               // P + 4 <= 2 048 at L + 2 D + 1 <= 64
    float p_old[KOUT], m_old[KOUT], v_old[KOUT];
#pragma unroll
    for (int k = 0; k < KOUT; ++k) {
        const int idx = min(t + LNT * k, P - 1);
        p_old[k] = a.params[idx]; m_old[k] = a.m[idx]; v_old[k] = a.v[idx];
    }
    for (int i = t; i < off_epsp + L; i += LNT) {           // parameters as float64, each in the array the products read
        const double pv = (double)a.params[i];
        if (i < off_be) Wed[i] = pv;
        else if (i < off_wd) bed[i - off_be] = pv;
        else if (i < off_bd) Wdd[i - off_wd] = pv;
        else if (i < off_epsp) bdd[i - off_bd] = pv;
        else { sd[i - off_epsp] = exp(0.5 * pv); elv[i - off_epsp] = exp(pv); }
    }
    const double eps = a.off_eps >= 0 ? (double)a.params[a.off_eps] * (double)a.eps_cli : (double)a.eps_cli;
    for (int e = t; e < NBLK * 256; e += LNT) {
        const int k = e >> 8, i = (e >> 4) & 15, j = e & 15;
        int b1 = 0, rem = k;
        while (rem >= NB - b1) { rem -= NB - b1; ++b1; }
        const int b2 = b1 + rem;
        const double v = a.M_in[e];
        Mf[(16 * b1 + i) * NFP + 16 * b2 + j] = v;
        Mf[(16 * b2 + j) * NFP + 16 * b1 + i] = v;         // (diagonal blocks are bitwise symmetric: same products, same order)
    }
    const double sigma = exp(0.5 * eps), inv_var = exp(-eps);
    __syncthreads();
    LIN_STAMP(1);
    for (int e = t; e < L * NFP; e += LNT) {              // SM = S M
        const int l = e / NFP, f = e % NFP;
        double s = sd[l] * Mf[l * NFP + f] + bed[l] * Mf[fone * NFP + f];
#pragma unroll
        for (int dd = 0; dd < (DT ? DT : 32); ++dd) if (DT || dd < D) s += Wed[dd * L + l] * Mf[(L + dd) * NFP + f];
        SM[e] = s;
    }
    __syncthreads();
    LIN_STAMP(2);
    for (int e = t; e < D * NFP; e += LNT) {              // P1 = R M
        const int d = e / NFP, f = e % NFP;
        double s = sigma * Mf[(L + D + d) * NFP + f] - Mf[(L + d) * NFP + f] + bdd[d] * Mf[fone * NFP + f];
#pragma unroll
        for (int l = 0; l < (LT ? LT : 32); ++l) if (LT || l < L) s += Wdd[l * D + d] * SM[l * NFP + f];
        P1[e] = s;
    }
    __syncthreads();
    LIN_STAMP(3);
    for (int e = t; e < L * NFP; e += LNT) {              // G = Wd P1
        const int l = e / NFP, f = e % NFP;
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < (DT ? DT : 32); ++d) if (DT || d < D) s += Wdd[l * D + d] * P1[d * NFP + f];
        G[e] = s;
    }
    for (int e = t; e < L * D; e += LNT) {                // dwd[l][d] = S[l,:] . P1[d,:]
        const int l = e / D, d = e % D;
        double s = sd[l] * P1[d * NFP + l] + bed[l] * P1[d * NFP + fone];
#pragma unroll
        for (int dd = 0; dd < (DT ? DT : 32); ++dd) if (DT || dd < D) s += Wed[dd * L + l] * P1[d * NFP + L + dd];
        dwd[e] = s;
    }
    __syncthreads();
    LIN_STAMP(4);
    // the four scalar sums: each thread a strided share, lanes by xor-shuffle, the four waves in order (all fixed order)
    double p_ssq = 0.0, p_musq = 0.0, p_z2r = 0.0, p_klc = 0.0;
    for (int e = t; e < L * D; e += LNT) p_ssq += Wdd[e] * dwd[e];
    if (t < D) {
        p_ssq += -P1[t * NFP + L + t] + sigma * P1[t * NFP + L + D + t] + bdd[t] * P1[t * NFP + fone];
        p_z2r = P1[t * NFP + L + D + t];
    }
    for (int e = t; e < L * (D + 1); e += LNT) {
        const int l = e / (D + 1), dd = e % (D + 1);
        const int f = dd < D ? L + dd : fone;
        const double q = SM[l * NFP + f] - sd[l] * Mf[l * NFP + f];            // Q = E M
        p_musq += (dd < D ? Wed[dd * L + l] : bed[l]) * q;
    }
    if (t < L) p_klc = 1.0 + 2.0 * log(sd[t]) - elv[t];                        // 1 + lv - e^{lv}
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        p_ssq += __shfl_xor(p_ssq, o, 64); p_musq += __shfl_xor(p_musq, o, 64);
        p_z2r += __shfl_xor(p_z2r, o, 64); p_klc += __shfl_xor(p_klc, o, 64);
    }
    if ((t & 63) == 0) { red[t >> 6] = p_ssq; red[LNW + (t >> 6)] = p_musq; red[2 * LNW + (t >> 6)] = p_z2r; red[3 * LNW + (t >> 6)] = p_klc; }
    __syncthreads();
    LIN_STAMP(5);
    double ssq = 0.0, musq = 0.0, z2r = 0.0, klc = 0.0;
#pragma unroll
    for (int w = 0; w < LNW; ++w) { ssq += red[w]; musq += red[LNW + w]; z2r += red[2 * LNW + w]; klc += red[3 * LNW + w]; }
    LIN_STAMP(6);
    const double inv_bt = (double)a.inv_bt, rows = (double)a.rows, c0 = inv_var * inv_bt;
    float bc1, bc2;
    bc1 = -expm1f((float)tstep * -0.10536051565782628f);
    bc2 = -expm1f((float)tstep * -0.0010005003335835335f);
#pragma unroll
    for (int k = 0; k < KOUT; ++k) {
        const int idx = t + LNT * k;
        if (idx >= P + kExtra) continue;
        double gd = 0.0;
        if (idx < off_be) { const int d = idx / L, l = idx % L; gd = c0 * G[l * NFP + L + d] + (SM[l * NFP + L + d] - sd[l] * Mf[l * NFP + L + d]) * inv_bt; }
        else if (idx < off_wd) { const int l = idx - off_be; gd = c0 * G[l * NFP + fone] + (SM[l * NFP + fone] - sd[l] * Mf[l * NFP + fone]) * inv_bt; }
        else if (idx < off_bd) gd = c0 * dwd[idx - off_wd];
        else if (idx < off_epsp) gd = c0 * P1[(idx - off_bd) * NFP + fone];
        else if (idx < off_epsp + L) {
            const int l = idx - off_epsp;
            gd = 0.5 * sd[l] * c0 * G[l * NFP + l] - 0.5 * (1.0 - elv[l]) * (double)a.rows_over_bt;
        } else if (idx == a.off_eps) {
            gd = (double)a.eps_cli * (-0.5 * ssq * inv_var + 0.5 * rows * D + 0.5 * sigma * z2r * inv_var) * inv_bt;
        } else if (idx >= P && idx < P + 3) {
            const double dkl = (0.5 * musq - 0.5 * rows * klc) * inv_bt;
            const double mse = (0.5 * ssq * inv_var + 0.5 * rows * D * ((double)kLog2Pi + eps)) * inv_bt;
            gd = idx == P ? dkl + mse : (idx == P + 1 ? dkl : mse);
        }
        const float gf = (float)gd;
        a.grads[idx] = gf;
        if (idx == P && a.loss_hist) a.loss_hist[(long long)(tstep - 1) % a.loss_hist_cap] = gf;
        if (idx < P) {
            float p = p_old[k], mm = m_old[k], vv = v_old[k];
            adam_apply_f(p, gf, mm, vv, a.lr, bc1, bc2);
            a.params[idx] = p; a.m[idx] = mm; a.v[idx] = vv;
        }
    }
    if (t == 0) a.step_dev[0] = tstep;
    LIN_STAMP(7);
}

template <int NB, int DT, int LT>
__global__ __launch_bounds__(LNT) void lin_step_kernel(const LinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lin_smem[];
    const int b = blockIdx.x;
    if (b < a.has_update) lin_update<NB, DT, LT>(a, lin_smem);
    else if (b < a.has_update + a.n_reduce) lin_reduce(a, lin_smem, b - a.has_update, NB * (NB + 1) / 2 * 256);
    else lin_stream<NB>(a, lin_smem, b - a.has_update - a.n_reduce);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
// 16-feature blocks of the kernel instantiation that serves this model: 3 (up to 48 features: the metric's 45) or 4
static int lin_nb(const vaek_ctx* c) { return (c->L + 2 * c->D + 1 + 15) / 16 <= 3 ? 3 : 4; }

bool lin_steps_supported(const vaek_ctx* c) {
    return c->cfg.n_enc_hidden == 0 && c->cfg.n_dec_hidden == 0 && !c->cfg.sigmoid_decoder && c->cfg.dtype == VAEK_F32 &&
           c->cfg.world == 1 && c->L + 2 * c->D + 1 <= 64 && (long long)c->B * std::min(c->D, c->L) >= 8;
}

static size_t lin_lds_bytes(const vaek_ctx* c) {
    const int NB = lin_nb(c), NFP = 16 * NB, NBLK = NB * (NB + 1) / 2, D = c->D, L = c->L;
    const size_t stream_in = (size_t)(L * 1024 + 4095) / 4096 * 4096 + 2 * ((size_t)(D * 1024 + 4095) / 4096 * 4096) + 1024 + 16;
    const size_t stream_red = (size_t)(NBLK <= 6 ? LNW : LNW / 2) * NBLK * 256 * sizeof(float);
    const size_t upd = sizeof(double) * ((size_t)NFP * NFP + (size_t)(D + 2 * L) * NFP + 3 * (size_t)D * L + 4 * L + D + 4 * LNW);
    return std::max(std::max(stream_in, stream_red), std::max(upd, (size_t)32 * 32 * sizeof(double)));
}

size_t lin_steps_workspace_bytes(const vaek_ctx* c) {
    if (!lin_steps_supported(c)) return 0;
    const int NB = lin_nb(c), no = NB * (NB + 1) / 2 * 256, ntiles = (c->B + 255) / 256;
    return 2 * ((size_t)ntiles * no * sizeof(float) + 256) + 2 * ((size_t)no * sizeof(double) + 256);
}

int lin_train_steps(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev, const float* const* xs,
                    const float* const* z1s, const float* const* z2s, int n_steps, float lr, void* ws, hipStream_t st) {
    const int NB = lin_nb(c), no = NB * (NB + 1) / 2 * 256, ntiles = (c->B + 255) / 256;
    char* base = static_cast<char*>(ws) + c->ws_lin;
    const size_t pbytes = ((size_t)ntiles * no * sizeof(float) + 255) / 256 * 256, mbytes = ((size_t)no * sizeof(double) + 255) / 256 * 256;
    float* partial[2] = {reinterpret_cast<float*>(base), reinterpret_cast<float*>(base + pbytes)};
    double* Mbuf[2] = {reinterpret_cast<double*>(base + 2 * pbytes), reinterpret_cast<double*>(base + 2 * pbytes + mbytes)};
    const size_t lds = lin_lds_bytes(c);
    // the metric's shape with its dimensions at compile time; every other linear model on the run-time instantiations
    const int which = (c->D == 12 && c->L == 20) ? 0 : (NB <= 3 ? 1 : 2);
    void (*fn)(const LinArgs) = which == 0 ? lin_step_kernel<3, 12, 20> : which == 1 ? lin_step_kernel<3, 0, 0> : lin_step_kernel<4, 0, 0>;
    static thread_local bool attr_set[3] = {false, false, false};
    if (!attr_set[which]) {
        VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[which] = true;
    }
    static const int roles = getenv("VAEK_LIN_ROLES") ? atoi(getenv("VAEK_LIN_ROLES")) : 7;   // diagnostic: 1 stream, 2 reduce, 4 update
    for (int n = 0; n < n_steps + 2; ++n) {       // launch n: stream batch n, reduce batch n - 1, update batch n - 2
        LinArgs a{};
        a.has_update = (n >= 2 && (roles & 4)) ? 1 : 0;
        a.n_reduce = (n >= 1 && n <= n_steps && (roles & 2)) ? (no + 31) / 32 : 0;
        a.n_stream = (n < n_steps && (roles & 1)) ? ntiles : 0;
        if (a.has_update + a.n_reduce + a.n_stream == 0) continue;
        a.B = c->B; a.D = c->D; a.L = c->L; a.ntiles = ntiles;
        if (a.n_stream) { a.x = xs[n]; a.z1 = z1s[n]; a.z2 = z2s[n]; a.partial_out = partial[n & 1]; }
        if (a.n_reduce) { a.partial_in = partial[(n - 1) & 1]; a.M_out = Mbuf[(n - 1) & 1]; }
        a.M_in = Mbuf[n & 1];                      // (n - 2) & 1
        a.params = params; a.grads = grads; a.m = m; a.v = v; a.step_dev = step_dev; a.lr = lr;
        a.inv_bt = (float)(1.0 / (double)c->Bt); a.eps_cli = c->cfg.eps_cli; a.rows = (float)c->B;
        a.rows_over_bt = (float)((double)c->B / (double)c->Bt); a.off_eps = (int)c->off_eps; a.P = (int)c->P;
        a.loss_hist = c->loss_hist; a.loss_hist_cap = c->loss_hist_cap;
        const unsigned grid = (unsigned)(a.has_update + a.n_reduce + a.n_stream);
        ProfScope ps(a.n_stream ? "lin_moments_step" : "lin_moments_drain", st);
        launch_k(ps, fn, dim3(grid), dim3(LNT), lds, st, a);
    }
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

}  // namespace vaek

#ifdef VAEK_LIN_STAMPS
extern "C" int vaek_debug_lin_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(vaek::g_lin_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#endif
