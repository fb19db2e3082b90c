// Layer-by-layer Dense kernels (K1-K3 of SURVEY.md 7.1), float32 in / float32 accumulate on the
// gfx950 f32 matrix cores: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain, so this
// path carries the 1e-5-relative-ELBO parity contract for every layer shape.
//
// One kernel template, three uses (flax.nn.Dense call sites networks.py:32-34 and their
// value_and_grad transposes, networks.py:99):
//   forward   Y[B,N]   = act(X[B,K] W[K,N] + b)            A: k-contiguous, B: n-contiguous
//   dX        dX[B,K]  = (dY[B,N] W^T) * relu'(X)          A: k-contiguous, B: k-contiguous
//   dW | db   G[K+1,N] = [X | 1]^T dY   split over batch   A: m-contiguous (+ones row), B: n-contiguous
// Block = 256 threads = 4 waves, each wave TM x TN 32x32 MFMA tiles; output tile 64x64, 128x128, 128x32 or 32x128.
// LDS tiles are k-major ([k][m] / [k][n]) so every MFMA operand read is 32 consecutive floats
// (conflict-free ds_read_b32); the next K-tile is fetched into registers while the MFMAs of the
// current one run (issue-early / write-late staging).
#include "vaek_internal.h"

namespace vaek {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int NT = 256;      // BK (k-tile depth) is a template parameter: 16, or 32 for the streaming skinny shapes
// Tile shapes: WM x WN waves of 32x32 each (WM * WN = 4): 2x2 = 64x64 for square-ish layers, 4x1 = 128x32
// for skinny outputs (N <= 32: the latent / data dims of every VAE here), 1x4 = 32x128 for skinny M (the
// dW|db of an input layer with a handful of features).  With a 64-wide tile a 20-column output keeps half
// of each block's MFMAs busy on columns nobody stores.

enum { EPI_FWD = 0, EPI_REPARAM = 1, EPI_DX = 2, EPI_DW = 3, EPI_ELBO = 4 };

// Operand storage types: float everywhere on the f32 path; in the bf16-storage mode (gemm_bf16s.hip, api.hip) the
// skinny first / last layer of a stack runs on THIS exact-f32 kernel with the hidden-side operand stored as bf16
// (TA / TB: operands, TC: output, TX: the relu-mask source of the dX epilogue), converted while it is staged.
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<__bf16>(const __bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <typename T> __device__ __forceinline__ bool vec4_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(T) - 1)) == 0; }

struct GemmArgs {
    const void* A; const void* B; void* C;      // element types are the kernel's TA / TB / TC
    int M, N, K;              // logical dims, K = reduction
    int lda, ldb, ldc;
    int a_mem;                // A, m-contiguous mode: number of real m columns in memory (ones row at index a_mem)
    const float* bias; int relu;
    const void* aux;          // DX: x_post [M, ldc] (type TX); REPARAM: z1 [M, ldc]; ELBO: x [M, ldc] (float)
    float* C2;                // REPARAM: samples
    const float* lv;          // REPARAM: logvar_e [N]
    int accumulate;
    int k_per_split; long long slab_stride;   // DW
    // ELBO (decoder's last layer, networks.py:80-83 + 94-98 fused into its epilogue): aux = data batch x [M, ldc],
    // aux2 = z2 [M, ldc]; C receives dL/dx_hat instead of the layer output; part[(by * nbx + bx) * 2] = this tile's
    // {sum of mse terms, sum of d eps terms}
    const float* aux2; float* part; const float* eps_param; float eps_cli, inv_bt;
};

// Operand tile = T (mn) x BK (k) floats, staged k-major into LDS rows of T + 4 floats.
// k-contiguous source p[mn*ld + k]: float4 unit u -> row u / (BK/4), k = 4 (u % (BK/4)); T*BK/4 units over 256 threads.
template <int T, int BK, typename E>
__device__ __forceinline__ void fetch_kcont(const E* __restrict__ p, int ld, int mn0, int MN, int k0, int kend,
                                            bool vec_ok, float (&v)[(T * BK / 4 + NT - 1) / NT][4]) {
    constexpr int NU = (T * BK / 4 + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = threadIdx.x + i * NT;
        v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        if (u >= T * BK / 4) continue;
        const int mn = mn0 + u / (BK / 4), k = k0 + (u % (BK / 4)) * 4;
        if (mn < MN) {
            const E* q = p + (long long)mn * ld + k;
            if (vec_ok && k + 3 < kend) {
                const float4 f = ld4<E>(q);
                v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (k + c < kend) v[i][c] = (float)q[c];
            }
        }
    }
}
template <int T, int BK>
__device__ __forceinline__ void store_kcont(float* s, const float (&v)[(T * BK / 4 + NT - 1) / NT][4]) {
    constexpr int NU = (T * BK / 4 + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = threadIdx.x + i * NT;
        if (u >= T * BK / 4) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) s[((u % (BK / 4)) * 4 + c) * (T + 4) + u / (BK / 4)] = v[i][c];
    }
}
// mn-contiguous source p[k*ld + mn]: float4 unit u -> k = u / (T/4), mn = 4 (u % (T/4)); mn == mem (only for the
// augmented operand) reads as 1, mn > mem as 0.
template <int T, int BK, typename E>
__device__ __forceinline__ void fetch_mncont(const E* __restrict__ p, int ld, int mn0, int mem, bool aug, int k0,
                                             int kend, bool vec_ok, float (&v)[(T * BK / 4 + NT - 1) / NT][4]) {
    constexpr int NU = (T * BK / 4 + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = threadIdx.x + i * NT;
        v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        if (u >= T * BK / 4) continue;
        const int k = k0 + u / (T / 4), mn = mn0 + (u % (T / 4)) * 4;
        if (k < kend) {
            const E* q = p + (long long)k * ld + mn;
            if (vec_ok && mn + 3 < mem) {
                const float4 f = ld4<E>(q);
                v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (mn + c < mem) v[i][c] = (float)q[c];
                    else if (aug && mn + c == mem) v[i][c] = 1.f;
                }
            }
        }
    }
}
template <int T, int BK>
__device__ __forceinline__ void store_mncont(float* s, const float (&v)[(T * BK / 4 + NT - 1) / NT][4]) {
    constexpr int NU = (T * BK / 4 + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = threadIdx.x + i * NT;
        if (u >= T * BK / 4) continue;
        *reinterpret_cast<float4*>(&s[(u / (T / 4)) * (T + 4) + (u % (T / 4)) * 4]) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
    }
}

// Full-tile forms: every load unconditional.  The general forms above load under `if (mn < MN)` / `if (k + 3 < kend)`;
// hipcc merges a conditionally loaded value into its zero-initialised registers INSIDE the branch, i.e. it waits for each
// load where it is issued -- the "prefetch under the MFMAs" was a chain of exposed latencies.  A tile that lies fully
// inside the operand (almost all of them) takes these.
template <int T, int BK, typename E>
__device__ __forceinline__ void fetch_kcont_full(const E* __restrict__ p, int ld, int mn0, int k0,
                                                 float (&v)[(T * BK / 4 + NT - 1) / NT][4]) {
    constexpr int NU = (T * BK / 4 + NT - 1) / NT;
    static_assert(T * BK / 4 % NT == 0, "whole passes");
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = threadIdx.x + i * NT;
        const float4 f = ld4<E>(p + (long long)(mn0 + u / (BK / 4)) * ld + k0 + (u % (BK / 4)) * 4);
        v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;
    }
}
template <int T, int BK, typename E>
__device__ __forceinline__ void fetch_mncont_full(const E* __restrict__ p, int ld, int mn0, int k0,
                                                  float (&v)[(T * BK / 4 + NT - 1) / NT][4]) {
    constexpr int NU = (T * BK / 4 + NT - 1) / NT;
    static_assert(T * BK / 4 % NT == 0, "whole passes");
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = threadIdx.x + i * NT;
        const float4 f = ld4<E>(p + (long long)(k0 + u / (T / 4)) * ld + mn0 + (u % (T / 4)) * 4);
        v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;
    }
}

template <typename TA, typename TB, typename TC, typename TX, bool A_KCONT, bool B_KCONT, int EPI, int WM, int WN, int BK, int TM = 1, int TN = 1>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(const GemmArgs g) {
    const TA* const gA = static_cast<const TA*>(g.A);
    const TB* const gB = static_cast<const TB*>(g.B);
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, SA = BM + 4, SB = BN + 4;
    __shared__ __attribute__((aligned(16))) float As[BK * SA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * SB];
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so give each XCD
    // a CONTIGUOUS run of the linearised (x fastest) tile space: the tiles that share an A row panel / B
    // column panel then hit the same L2 instead of each fetching the panel again (speed only, never
    // correctness; needs the tile count divisible by 8, else the natural order is kept).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, nb = gx * gy * gridDim.z;
        if (nb % 8 == 0) {
            const unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
            const unsigned sw = (id % 8) * (nb / 8) + id / 8;
            bx = sw % gx; by = (sw / gx) % gy; bz = sw / (gx * gy);
        }
    }
    const int m0 = by * BM, n0 = bx * BN;
    int kbeg = 0, kend = g.K;
    if (EPI == EPI_DW) {
        kbeg = bz * g.k_per_split;
        kend = min(g.K, kbeg + g.k_per_split);
    }
    const bool a_vec = (g.lda % 4 == 0) && vec4_aligned<TA>(g.A) && (A_KCONT ? (kbeg % 4 == 0) : true);
    const bool b_vec = (g.ldb % 4 == 0) && vec4_aligned<TB>(g.B) && (B_KCONT ? (kbeg % 4 == 0) : true);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;

    // Register staging: one k-tile ahead, or TWO for the tall-skinny streaming shape (128 x 32 tile, a few hundred
    // workgroups = one per CU: a single 16 KB tile in flight per CU cannot cover HBM latency).
    constexpr bool DEEP = (WM == 4 && EPI != EPI_DW);
    constexpr int NSET = DEEP ? 2 : 1;      // (four k-tiles in flight measured slower than two: C4's encoder forward 208 -> 233 us, dX 160 -> 171)
    float ra[NSET][(BM * BK / 4 + NT - 1) / NT][4], rb[NSET][(BN * BK / 4 + NT - 1) / NT][4];
    // does this workgroup's tile lie fully inside each operand (rows / columns; the k range is checked per k-tile)?
    const bool a_in = a_vec && (A_KCONT ? m0 + BM <= g.M : m0 + BM <= g.a_mem);
    const bool b_in = b_vec && n0 + BN <= g.N;
    auto fetch = [&](int k0, int set) {
        const bool k_in = k0 + BK <= kend;                 // uniform: a scalar branch
        if (a_in && k_in) { if (A_KCONT) fetch_kcont_full<BM, BK>(gA, g.lda, m0, k0, ra[set]); else fetch_mncont_full<BM, BK>(gA, g.lda, m0, k0, ra[set]); }
        else if (A_KCONT) fetch_kcont<BM, BK>(gA, g.lda, m0, g.M, k0, kend, a_vec, ra[set]);
        else fetch_mncont<BM, BK>(gA, g.lda, m0, g.a_mem, EPI == EPI_DW, k0, kend, a_vec, ra[set]);
        if (b_in && k_in) { if (B_KCONT) fetch_kcont_full<BN, BK>(gB, g.ldb, n0, k0, rb[set]); else fetch_mncont_full<BN, BK>(gB, g.ldb, n0, k0, rb[set]); }
        else if (B_KCONT) fetch_kcont<BN, BK>(gB, g.ldb, n0, g.N, k0, kend, b_vec, rb[set]);
        else fetch_mncont<BN, BK>(gB, g.ldb, n0, g.N, false, k0, kend, b_vec, rb[set]);
    };
    auto stage_and_multiply = [&](int k0, int set) {
        __syncthreads();                       // previous tile fully consumed
        if (A_KCONT) store_kcont<BM, BK>(As, ra[set]); else store_mncont<BM, BK>(As, ra[set]);
        if (B_KCONT) store_kcont<BN, BK>(Bs, rb[set]); else store_mncont<BN, BK>(Bs, rb[set]);
        __syncthreads();
        if (k0 + NSET * BK < kend) fetch(k0 + NSET * BK, set);    // in flight under the MFMAs below (and the next tile's)
        const float* pa = As + (lane >> 5) * SA + wm * 32 * TM + (lane & 31);
        const float* pb = Bs + (lane >> 5) * SB + wn * 32 * TN + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = pa[kk * SA + 32 * i];
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) bv[jn] = pb[kk * SB + 32 * jn];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[jn], acc[i][jn], 0, 0, 0);
        }
    };
#pragma unroll
    for (int s2 = 0; s2 < NSET; ++s2)
        if (kbeg + s2 * BK < kend) fetch(kbeg + s2 * BK, s2);
    for (int k0 = kbeg; k0 < kend; k0 += NSET * BK) {
#pragma unroll
        for (int s2 = 0; s2 < NSET; ++s2)
            if (k0 + s2 * BK < kend) stage_and_multiply(k0 + s2 * BK, s2);
    }
    // C/D map of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float e_mse = 0.f, e_deps = 0.f, e_inv_var = 0.f, e_sigma = 0.f, e_dscale = 0.f;
    if (EPI == EPI_ELBO) {
        const float eps = g.eps_param ? g.eps_param[0] * g.eps_cli : g.eps_cli;
        e_inv_var = expf(-eps); e_sigma = expf(0.5f * eps); e_dscale = e_inv_var * g.inv_bt;
    }
    // The epilogue's own inputs (relu mask source / z1 / x and z2) are gathered into registers BEFORE the first store:
    // C and aux are plain pointers, so every load after a store to C has to wait behind it (they may alias as far as the
    // compiler knows) and the epilogue became a chain of exposed load latencies -- dX ran 25 % slower than the forward
    // GEMM of the same shape.
    constexpr bool kAux = EPI == EPI_REPARAM || EPI == EPI_DX || EPI == EPI_ELBO;
    float auxv[kAux ? TM : 1][kAux ? TN : 1][16], aux2v[EPI == EPI_ELBO ? TM : 1][EPI == EPI_ELBO ? TN : 1][16];
    if (kAux && (EPI != EPI_DX || g.relu)) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = n0 + (wn * TN + jn) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const bool in = col < g.N && row < g.M;
                    const long long o = in ? (long long)row * g.ldc + col : 0;
                    auxv[kAux ? i : 0][kAux ? jn : 0][r] = EPI == EPI_DX ? (float)static_cast<const TX*>(g.aux)[o] : static_cast<const float*>(g.aux)[o];
                    if (EPI == EPI_ELBO) aux2v[EPI == EPI_ELBO ? i : 0][EPI == EPI_ELBO ? jn : 0][r] = g.aux2[o];
                }
        }
    }
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int col = n0 + (wn * TN + jn) * 32 + (lane & 31);
        if (col >= g.N) continue;
        float bias = 0.f, sdev = 0.f;
        if (EPI == EPI_FWD || EPI == EPI_REPARAM || EPI == EPI_ELBO) bias = g.bias ? g.bias[col] : 0.f;
        if (EPI == EPI_REPARAM) sdev = expf(0.5f * g.lv[col]);
        TC* C = static_cast<TC*>(g.C);
        if (EPI == EPI_DW) C += (long long)bz * g.slab_stride;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                const long long o = (long long)row * g.ldc + col;
                float v = acc[i][jn][r];
                const float ax = auxv[kAux ? i : 0][kAux ? jn : 0][r];
                if (EPI == EPI_FWD) {
                    v += bias;
                    if (g.relu) v = fmaxf(v, 0.f);
                    C[o] = (TC)v;
                } else if (EPI == EPI_REPARAM) {
                    v += bias;
                    C[o] = (TC)v;
                    g.C2[o] = v + sdev * ax;
                } else if (EPI == EPI_DX) {
                    if (g.relu) v = ax > 0.f ? v : 0.f;
                    if (g.accumulate) v += (float)C[o];
                    C[o] = (TC)v;
                } else if (EPI == EPI_ELBO) {
                    const float z = aux2v[EPI == EPI_ELBO ? i : 0][EPI == EPI_ELBO ? jn : 0][r];
                    const float rr = (v + bias) + e_sigma * z - ax;              // x_hat - x, x_hat = y + z2 e^{eps/2}
                    const float q = rr * rr * e_inv_var;
                    e_mse += 0.5f * q;
                    e_deps += -0.5f * q + 0.5f * e_sigma * z * rr * e_inv_var;
                    C[o] = (TC)(rr * e_dscale);
                } else {
                    C[o] = (TC)v;
                }
            }
        }
    }
    if (EPI == EPI_ELBO) {      // this tile's two sums, fixed order: lanes by xor-shuffle, then the four waves in order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { e_mse += __shfl_xor(e_mse, o, 64); e_deps += __shfl_xor(e_deps, o, 64); }
        __syncthreads();                                   // the operand tiles in As are dead
        if (lane == 0) { As[2 * wave] = e_mse; As[2 * wave + 1] = e_deps; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float* p = g.part + ((long long)by * gridDim.x + bx) * 2;
            p[0] = ((As[0] + As[2]) + As[4]) + As[6];
            p[1] = ((As[1] + As[3]) + As[5]) + As[7];
        }
    }
}

// ---- tall-skinny streaming form: C[M, N <= 32] = A[M, K] . B[K, N] with a long K (C4's 4096 -> 20 encoder forward and the decoder's
// input gradient; the conv VAE's 4096 -> 32 bottleneck).  The tiled kernel above gives such a layer ONE 128 x 32 tile = four waves
// per CU, far too few bytes in flight for an HBM stream (2.6 TB/s).  Here a workgroup owns 32 rows and its four waves SPLIT K: no
// LDS staging, no barrier in the loop -- lane (row, h) streams its own row with 16-byte loads (k = k0 + 8 u + 4 h .. + 3 pairs with
// the other half-wave's k + 4: any pairing works as long as A and B agree), 32 rows per workgroup = 4x the workgroups, and the four
// partial tiles meet through LDS in a fixed order before the epilogue.
struct TsArgs {
    const float* A; const float* B; float* C; float* C2; const float* bias; const float* aux; const float* lv;
    int M, N, K, lda, ldb, ldc, accumulate;
};
template <int EPI, bool B_KCONT>
__global__ __launch_bounds__(256) void ts_gemm_kernel(const TsArgs g) {
    __shared__ float red[4][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, row = min(m0 + i, g.M - 1), col = min(i, g.N - 1);
    const int kq = g.K / 4, kbeg = wave * kq;                 // K % 32 == 0: every wave's range is whole 8-deep steps
    const float* pa = g.A + (long long)row * g.lda + kbeg + 4 * h;
    const float* pb = B_KCONT ? g.B + (long long)col * g.ldb + kbeg + 4 * h : g.B + (long long)(kbeg + 4 * h) * g.ldb + col;
    f32x16 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    constexpr int U = 4;                                      // 16-byte loads of A per lane and trip: 32 k per trip
    float4 a[2][U], b[2][U];
    auto fetch = [&](int k, int set) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[set][u] = *reinterpret_cast<const float4*>(pa + k + 8 * u);
            if (B_KCONT) b[set][u] = *reinterpret_cast<const float4*>(pb + k + 8 * u);
            else {
                const float* q = pb + (long long)(k + 8 * u) * g.ldb;
                b[set][u] = make_float4(q[0], q[g.ldb], q[2 * g.ldb], q[3 * g.ldb]);
            }
        }
    };
    auto multiply = [&](int set) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][u].x, b[set][u].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][u].y, b[set][u].y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][u].z, b[set][u].z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][u].w, b[set][u].w, acc[1], 0, 0, 0);
        }
    };
    fetch(0, 0);
    for (int k = 0; k < kq; k += 2 * 8 * U) {                 // two trips per iteration: the other set loads under this one's MFMAs
        if (k + 8 * U < kq) fetch(k + 8 * U, 1);
        multiply(0);
        if (k + 8 * U < kq) {
            if (k + 2 * 8 * U < kq) fetch(k + 2 * 8 * U, 0);
            multiply(1);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][i] = acc[0][r] + acc[1][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * g.N; e += 256) {
        const int rr = e / g.N, c = e % g.N, m = m0 + rr;
        if (m >= g.M) continue;
        float v = (red[0][rr][c] + red[1][rr][c]) + (red[2][rr][c] + red[3][rr][c]);
        const long long o = (long long)m * g.ldc + c;
        if (EPI == EPI_REPARAM) {
            v += g.bias ? g.bias[c] : 0.f;
            g.C[o] = v;
            g.C2[o] = v + expf(0.5f * g.lv[c]) * g.aux[o];
        } else {                                              // EPI_DX without a relu mask
            if (g.accumulate) v += g.C[o];
            g.C[o] = v;
        }
    }
}
// shapes the streaming form takes: a skinny output, a long reduction in whole 32-deep trips per wave, enough rows to fill the chip
static bool ts_shape_ok(int M, int N, int K, int lda, int ldb, const void* A, const void* B, bool b_kcont) {
    return N <= 32 && K >= 1024 && K % 128 == 0 && M >= 2048 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
           (!b_kcont || (ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0));
}

static thread_local int g_last_bm = 0, g_last_nbx = 0;      // tile rows / tile columns of the last launch (ELBO partials)

template <typename TA, typename TB, typename TC, typename TX, bool A_KCONT, bool B_KCONT, int EPI, int WM, int WN, int BK, int TM = 1, int TN = 1>
static int launch_shape(const GemmArgs& g, int splits, hipStream_t st) {
    // the 128 x 128 register-blocked shape carries its own label, so that tests can assert it ran (tests/test_gpu_wide.py)
    constexpr bool BIG = TM == 2 && TN == 2;
    ProfScope ps(EPI == EPI_FWD ? (BIG ? "gemm_f32_fwd_128x128" : "gemm_f32_fwd") : EPI == EPI_REPARAM ? "gemm_f32_fwd_reparam"
                 : EPI == EPI_DX ? (BIG ? "gemm_f32_dx_128x128" : "gemm_f32_dx") : EPI == EPI_ELBO ? "gemm_f32_fwd_elbo"
                 : (BIG ? "gemm_f32_dw_128x128" : "gemm_f32_dw"), st);
    dim3 grid((g.N + 32 * WN * TN - 1) / (32 * WN * TN), (g.M + 32 * WM * TM - 1) / (32 * WM * TM), splits);
    if (grid.y > 65535u || grid.z > 65535u) {
        set_error("gemm grid too large (M=%d N=%d splits=%d)", g.M, g.N, splits);
        return VAEK_ERR_INVALID;
    }
    g_last_bm = 32 * WM * TM; g_last_nbx = (int)grid.x;
    launch_k(ps, (gemm_f32_kernel<TA, TB, TC, TX, A_KCONT, B_KCONT, EPI, WM, WN, BK, TM, TN>), grid, dim3(NT), 0, st, g);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

template <bool A_KCONT, bool B_KCONT, int EPI, typename TA = float, typename TB = float, typename TC = float, typename TX = float>
static int launch(const GemmArgs& g, int splits, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return VAEK_OK;
    constexpr bool kAllF32 = sizeof(TA) == 4 && sizeof(TB) == 4 && sizeof(TC) == 4 && sizeof(TX) == 4;
    // skinny shapes stream a long K past a small output: a 32-deep k-tile halves their barriers per byte
    // (64-deep k-tiles for the long-row skinny forward / dX, 256-byte row pieces, measured slower again in round 2: C4's encoder
    // forward 208 -> 250 us, dX 160 -> 217)
    if (g.N <= 32) return launch_shape<TA, TB, TC, TX, A_KCONT, B_KCONT, EPI, 4, 1, 32>(g, splits, st);     // skinny output: 128 x 32
    if (g.M <= 32) return launch_shape<TA, TB, TC, TX, A_KCONT, B_KCONT, EPI, 1, 4, 32>(g, splits, st);     // skinny M:      32 x 128
    // wide layers: 128 x 128 (each wave 2 x 2 MFMA tiles) halves the operand bytes pulled through L2 per flop -- at
    // 64 x 64 the 512-wide layers of C3 need ~7 TB/s of L2 -> LDS traffic to keep the f32 matrix pipe busy (C3 forward
    // 1.75 -> 1.53 ms, dX 2.22 -> 1.93 ms, dW|db 2.23 -> 2.02 ms with the split count raised to match, api.hip).  Only
    // where the big tiles still give every CU two workgroups (C2, 8 192 rows: 128 tiles, is faster at 64 x 64) and
    // the reduction is long enough to matter (C4's 20 -> 4096 decoder is all epilogue: 148 us at 64 x 64, 186 at 128).
    // BK = 32 for this shape since the full-tile fetch made the prefetch asynchronous (C3 5.00 -> 4.91 ms; it was 5-8 %
    // slower while the conditional loads serialised it; 64 x 64 stays at 16: C2 0.156 vs 0.163 ms).
    // Measured and dropped here: a split count that makes the
    // workgroups a whole number per CU (2 per CU: 15 % slower than 2.4), the ones row of [X | 1]^T as a streaming
    // column sum instead of a fifth MFMA tile row (no change: the kernel is latency-, not MFMA-bound), dX on a
    // pre-transposed copy of W so that its B operand stages with 16-byte LDS stores like the forward's (no change),
    // (two register-staged k-tiles in flight lost 7 % while the fetch was serialised; with the full-tile fetch they win for
    // the tall-skinny forward / dX shape -- C4 0.966 -> 0.933 ms -- and still lose slightly for the dW shapes).
    // (the bf16-storage variants exist for the skinny first / last layers only: they skip the 128 x 128 instantiation)
    if constexpr (kAllF32)
        if (g.M >= 128 && g.N >= 128 && g.K >= 128 && (long long)((g.M + 127) / 128) * ((g.N + 127) / 128) * splits >= 512)
            return launch_shape<TA, TB, TC, TX, A_KCONT, B_KCONT, EPI, 2, 2, 32, 2, 2>(g, splits, st);
    return launch_shape<TA, TB, TC, TX, A_KCONT, B_KCONT, EPI, 2, 2, 16>(g, splits, st);                    // 64 x 64
}

int launch_dense_fwd(const float* x, const float* w, const float* b, float* y, int rows, int n_in,
                     int n_out, bool relu, hipStream_t st) {
    GemmArgs g{};
    g.A = x; g.B = w; g.C = y; g.M = rows; g.N = n_out; g.K = n_in;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.a_mem = 0; g.bias = b; g.relu = relu;
    return launch<true, false, EPI_FWD>(g, 1, st);
}

// The decoder's last Dense with the ELBO's elementwise pass in its epilogue: d_out receives dL/dx_hat (what the backward
// pass wants in that buffer anyway) and `part` one {mse, d eps} pair per output tile; *bm / *nbx tell the reducer
// (launch_elbo_reduce) how the tiles map to rows.  Saves writing and re-reading the B x D layer output.
int launch_dense_fwd_elbo(const float* h, const float* w, const float* b, float* d_out, const float* x, const float* z2,
                          const float* eps_param, float eps_cli, float inv_bt, float* part, int rows, int n_in, int n_out,
                          int* bm, int* nbx, hipStream_t st) {
    GemmArgs g{};
    g.A = h; g.B = w; g.C = d_out; g.M = rows; g.N = n_out; g.K = n_in;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.bias = b;
    g.aux = x; g.aux2 = z2; g.part = part; g.eps_param = eps_param; g.eps_cli = eps_cli; g.inv_bt = inv_bt;
    const int rc = launch<true, false, EPI_ELBO>(g, 1, st);
    *bm = g_last_bm; *nbx = g_last_nbx;
    return rc;
}

int launch_dense_fwd_reparam(const float* x, const float* w, const float* b, float* mu, float* samples,
                             const float* z1, const float* lv, int rows, int n_in, int n_out, hipStream_t st) {
    if (ts_shape_ok(rows, n_out, n_in, n_in, n_out, x, w, false)) {
        TsArgs t{};
        t.A = x; t.B = w; t.C = mu; t.C2 = samples; t.bias = b; t.aux = z1; t.lv = lv;
        t.M = rows; t.N = n_out; t.K = n_in; t.lda = n_in; t.ldb = n_out; t.ldc = n_out;
        ProfScope ps("gemm_f32_fwd_reparam_ts", st);
        launch_k(ps, (ts_gemm_kernel<EPI_REPARAM, false>), dim3((rows + 31) / 32), dim3(256), 0, st, t);
        VAEK_HIP_CHECK(hipGetLastError());
        return VAEK_OK;
    }
    GemmArgs g{};
    g.A = x; g.B = w; g.C = mu; g.C2 = samples; g.aux = z1; g.lv = lv;
    g.M = rows; g.N = n_out; g.K = n_in; g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.bias = b;
    return launch<true, false, EPI_REPARAM>(g, 1, st);
}

int launch_dense_bwd_dx(const float* dy, const float* w, const float* x_post, float* dx, int rows,
                        int n_in, int n_out, bool relu, bool accumulate, hipStream_t st) {
    // dX[rows, n_in] = dY[rows, n_out] . W^T ; B(k = out index, j = in index) = W[j*n_out + k]
    if (!(relu && x_post) && ts_shape_ok(rows, n_in, n_out, n_out, n_out, dy, w, true)) {
        TsArgs t{};
        t.A = dy; t.B = w; t.C = dx; t.M = rows; t.N = n_in; t.K = n_out; t.lda = n_out; t.ldb = n_out; t.ldc = n_in; t.accumulate = accumulate;
        ProfScope ps("gemm_f32_dx_ts", st);
        launch_k(ps, (ts_gemm_kernel<EPI_DX, true>), dim3((rows + 31) / 32), dim3(256), 0, st, t);
        VAEK_HIP_CHECK(hipGetLastError());
        return VAEK_OK;
    }
    GemmArgs g{};
    g.A = dy; g.B = w; g.C = dx; g.M = rows; g.N = n_in; g.K = n_out;
    g.lda = n_out; g.ldb = n_out; g.ldc = n_in; g.aux = x_post; g.relu = relu && x_post != nullptr;
    g.accumulate = accumulate;
    return launch<true, true, EPI_DX>(g, 1, st);
}

int launch_dense_bwd_dw(const float* x, const float* dy, float* slab0, int64_t slab_stride, int S,
                        int rows_per_split, int rows, int n_in, int n_out, hipStream_t st) {
    // G[(n_in+1), n_out] = [X | 1]^T . dY, reduction over the batch rows of this split
    GemmArgs g{};
    g.A = x; g.B = dy; g.C = slab0; g.M = n_in + 1; g.N = n_out; g.K = rows;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.a_mem = n_in;
    g.k_per_split = rows_per_split; g.slab_stride = slab_stride;
    return launch<false, false, EPI_DW>(g, S, st);
}

// ---- bf16-storage mode (api.hip, gemm_bf16s.hip): the first / last layer of a stack on the exact f32 kernel with its
// hidden-side operand stored as bf16 ------------------------------------------------------------------------------------
// first layer forward: x f32 [rows, n_in] -> relu(.) as bf16 [rows, n_out]
int launch_dense_fwd_out16(const float* x, const float* w, const float* b, __bf16* y, int rows, int n_in, int n_out, bool relu,
                           hipStream_t st) {
    GemmArgs g{};
    g.A = x; g.B = w; g.C = y; g.M = rows; g.N = n_out; g.K = n_in;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.bias = b; g.relu = relu;
    return launch<true, false, EPI_FWD, float, float, __bf16>(g, 1, st);
}
// last layer forward from a bf16 hidden activation: plain / reparameterisation / ELBO epilogues (f32 out)
int launch_dense_fwd_in16(const __bf16* x, const float* w, const float* b, float* y, int rows, int n_in, int n_out, hipStream_t st) {
    GemmArgs g{};
    g.A = x; g.B = w; g.C = y; g.M = rows; g.N = n_out; g.K = n_in;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.bias = b; g.relu = 0;
    return launch<true, false, EPI_FWD, __bf16>(g, 1, st);
}
int launch_dense_fwd_reparam_in16(const __bf16* x, const float* w, const float* b, float* mu, float* samples, const float* z1,
                                  const float* lv, int rows, int n_in, int n_out, hipStream_t st) {
    GemmArgs g{};
    g.A = x; g.B = w; g.C = mu; g.C2 = samples; g.aux = z1; g.lv = lv;
    g.M = rows; g.N = n_out; g.K = n_in; g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.bias = b;
    return launch<true, false, EPI_REPARAM, __bf16>(g, 1, st);
}
int launch_dense_fwd_elbo_in16(const __bf16* h, const float* w, const float* b, float* d_out, const float* x, const float* z2,
                               const float* eps_param, float eps_cli, float inv_bt, float* part, int rows, int n_in, int n_out,
                               int* bm, int* nbx, hipStream_t st) {
    GemmArgs g{};
    g.A = h; g.B = w; g.C = d_out; g.M = rows; g.N = n_out; g.K = n_in;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.bias = b;
    g.aux = x; g.aux2 = z2; g.part = part; g.eps_param = eps_param; g.eps_cli = eps_cli; g.inv_bt = inv_bt;
    const int rc = launch<true, false, EPI_ELBO, __bf16>(g, 1, st);
    *bm = g_last_bm; *nbx = g_last_nbx;
    return rc;
}
// last layer backward: dX as bf16 [rows, n_in] with the relu mask taken from the bf16 activation x_post
int launch_dense_bwd_dx_out16(const float* dy, const float* w, const __bf16* x_post, __bf16* dx, int rows, int n_in, int n_out,
                              bool accumulate, hipStream_t st) {
    GemmArgs g{};
    g.A = dy; g.B = w; g.C = dx; g.M = rows; g.N = n_in; g.K = n_out;
    g.lda = n_out; g.ldb = n_out; g.ldc = n_in; g.aux = x_post; g.relu = 1; g.accumulate = accumulate;
    return launch<true, true, EPI_DX, float, float, __bf16, __bf16>(g, 1, st);
}
// last layer dW|db: [X | 1]^T dY with X = bf16 hidden activation
int launch_dense_bwd_dw_x16(const __bf16* x, const float* dy, float* slab0, int64_t slab_stride, int S, int rows_per_split,
                            int rows, int n_in, int n_out, hipStream_t st) {
    GemmArgs g{};
    g.A = x; g.B = dy; g.C = slab0; g.M = n_in + 1; g.N = n_out; g.K = rows;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.a_mem = n_in; g.k_per_split = rows_per_split; g.slab_stride = slab_stride;
    return launch<false, false, EPI_DW, __bf16>(g, S, st);
}
// first layer dW|db: [X | 1]^T dY with dY = bf16 hidden gradient
int launch_dense_bwd_dw_dy16(const float* x, const __bf16* dy, float* slab0, int64_t slab_stride, int S, int rows_per_split,
                             int rows, int n_in, int n_out, hipStream_t st) {
    GemmArgs g{};
    g.A = x; g.B = dy; g.C = slab0; g.M = n_in + 1; g.N = n_out; g.K = rows;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.a_mem = n_in; g.k_per_split = rows_per_split; g.slab_stride = slab_stride;
    return launch<false, false, EPI_DW, float, __bf16>(g, S, st);
}
// first layer dX (the decoder's first layer: d samples, f32) from a bf16 hidden gradient
int launch_dense_bwd_dx_in16(const __bf16* dy, const float* w, float* dx, int rows, int n_in, int n_out, bool accumulate,
                             hipStream_t st) {
    GemmArgs g{};
    g.A = dy; g.B = w; g.C = dx; g.M = rows; g.N = n_in; g.K = n_out;
    g.lda = n_out; g.ldb = n_out; g.ldc = n_in; g.aux = nullptr; g.relu = 0; g.accumulate = accumulate;
    return launch<true, true, EPI_DX, __bf16>(g, 1, st);
}

}  // namespace vaek
