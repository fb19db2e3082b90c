// Dense kernels of the bf16-STORAGE mode (BASELINE configs 2-3: "MLP ... bf16 ... MFMA Dense path"): hidden activations and
// hidden gradients live in HBM as bf16, the products run on v_mfma_f32_32x32x16_bf16 with f32 accumulation
// (flax.nn.Dense call sites networks.py:32-39 and their value_and_grad transposes, networks.py:99).
//
// Why storage: at width 512 a Dense layer is 256 flop per byte of f32 activation traffic -- below the bf16 ridge
// (2.5 PF / 8 TB/s = 310 flop/B), i.e. with f32 activations (gemm_bf16.hip, round 1) the "matrix-core path" is an HBM
// stream that converts every element on every use.  Here every hidden tensor is written once as bf16 by the epilogue
// that produces it and consumed as bf16 -- half the bytes, no conversion in the operand path -- which makes direct
// global -> LDS loads (global_load_lds_dwordx4: no VGPR round trip, no ds_write) possible for every operand.
//
//   hs_nt_kernel   C[M,N] = A[M,K] . Bt[N,K]^T, both operands k-contiguous.  Forward: A = activations, Bt = W^T (a bf16
//                  transposed copy made once per step), epilogue + bias, relu.  dX: A = dY, Bt = W as stored ([in, out] =
//                  [n', k']), epilogue relu mask from the layer's bf16 input.  Operands swapped in the MFMA (D = Bt A^T),
//                  so a lane owns ONE output row and 4-column runs of it: with one v_permlane32_swap per dword the
//                  epilogue stores 16 bytes per lane, 128 contiguous bytes per row and wave (no LDS transpose, no 2-byte
//                  stores).
//   hs_tn_kernel   dW|db slab[M,N] = sum over this split's rows of X[k,m] dY[k,n]: both operands are row-major [rows,
//                  features], i.e. k-STRIDED -- the tiles are staged as they lie in memory and transposed on the way to
//                  the registers by ds_read_b64_tr_b16.  db = 1^T dY comes from two extra MFMAs with a constant-ones A
//                  operand in the workgroups of the first row of tiles.
//   cvt_w_kernel   bf16 copies W and W^T of the wide layers' kernels (weights change every step; 1 MB at C3).
//
// Main loop (both kernels): 64-deep k-tiles, two LDS buffers, ONE barrier per k-tile -- wait for tile t (vmcnt(0)),
// barrier, issue the loads of tile t+1 into the other buffer, multiply tile t.  LDS images are lane-linear (what
// global_load_lds writes) with the bank swizzle applied to the per-lane SOURCE address and the same XOR on the read.
#include "vaek_internal.h"

namespace vaek {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// 16 bytes per lane global -> LDS; the LDS side is wave-uniform base + 16 * lane.  In inline assembly since round 3: through
// __builtin_amdgcn_global_load_lds hipcc knows that LDS is being written and puts s_waitcnt vmcnt(0) in front of the next LDS
// read it cannot tell apart -- in every kernel of this file that was the operand reads of the k-tile being multiplied, RIGHT BEHIND
// the issue of the next k-tile's loads: the ring never overlapped a load with a product (ISA: "8 x global_load_lds, s_waitcnt
// vmcnt(0), ds_read_b128 ..."; round 2's ablation found loads and MFMAs additive and every ring depth equally fast).  Every wait
// on these loads is the kernels' own counted one (wait_tiles) in front of a barrier.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_void_t*)lds_wave_base);
    _Pragma("clang diagnostic push") _Pragma("clang diagnostic ignored \"-Winline-asm\"")      // (M0 on the clobber list: it is what the LDS-DMA takes its LDS address from)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "memory", "m0");
    _Pragma("clang diagnostic pop")
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

enum { HS_FWD = 0, HS_DX = 1 };

// Diagnostic build only (-DVAEK_HS_STAMPS, tools/hs_stamps.sh): per-wave s_memtime sums of the main loop's phases, written
// to a buffer nothing else reads.  The shipped build contains no stamp.
#ifdef VAEK_HS_STAMPS
__device__ unsigned long long* g_hs_stamp_buf = nullptr;
__device__ __forceinline__ unsigned long long hs_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define HS_STAMP(var) const unsigned long long var = hs_now()
#define HS_ADD(acc, a, b) acc += (b) - (a)
#else
#define HS_STAMP(var) do {} while (0)
#define HS_ADD(acc, a, b) do {} while (0)
#endif

struct HsArgs {
    const __bf16* A; const __bf16* Bt; __bf16* C;   // A [M, K] (lda), Bt [N, K] (ldb), C [M, N] (ldc); K, N multiples of 64
    int M, N, K, lda, ldb, ldc;
    const float* bias; int relu;                    // FWD
    const __bf16* aux;                              // DX: relu-mask source [M, ldc] (the layer's bf16 input), or nullptr
};

// Counted waits: s_waitcnt vmcnt(N) returns once at most N of this wave's vector-memory operations are still outstanding
// (they retire in issue order), i.e. "everything up to tile kt has landed, the `newer` younger tiles may stay in flight".
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int P, int MAXT> __device__ __forceinline__ void wait_tiles(int newer) {       // P loads per tile, newer in [0, MAXT]
    static_assert(P * MAXT <= 63, "vmcnt is a 6-bit counter");
    if constexpr (MAXT == 0) wait_vmcnt<0>();
    else if (newer >= MAXT) wait_vmcnt<P * MAXT>();
    else wait_tiles<P, MAXT - 1>(newer);
}

// LDS image of an R-row x BK-k operand tile (BK = 64 / 32): 128- / 64-byte rows, 16-byte chunk c of row r at chunk position
// c ^ f(r), f = (r >> 1) & 7 / (r >> 2) & 3: the 16 lanes ds_read_b128 services together (rows {0-3,12-15,20-27} /
// {4-11,16-19,28-31} of a fragment, one chunk column) then cover all 16 chunk slots of the 256-byte bank row.
//
// Ring of NS stages: a k-tile's MFMAs take ~0.25 us per wave, an HBM / L2 round trip under load 1-3 us, so ONE tile of
// prefetch (round 2's first version: 2.8 us per k-tile) is a latency chain; NS - 1 tiles are in flight while one is multiplied.
template <int EPI, int BM, int BN, int WM, int WN, int BK, int NS>
__global__ __launch_bounds__(64 * WM * WN) void hs_nt_kernel(const HsArgs g) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RB = 2 * BK, CPR = BK / 8;                 // row bytes, 16-byte chunks per row
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, BUF = A_BYTES + B_BYTES;
    constexpr int A_PASSES = BM * CPR / NTH, B_PASSES = BN * CPR / NTH, P = A_PASSES + B_PASSES;
    static_assert(BM * CPR % NTH == 0 && BN * CPR % NTH == 0, "whole staging passes");
    static_assert(BK == 64 || BK == 32, "k-tile depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    HS_STAMP(k0);
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of the
    // n-fastest tile space so that the tiles sharing an A row panel hit one L2 (speed only)
    const unsigned tiles_n = (g.N + BN - 1) / BN, nb = gridDim.x;
    unsigned id = blockIdx.x;
    if (nb % 8 == 0) id = (id % 8) * (nb / 8) + id / 8;
    const int m0 = (int)(id / tiles_n) * BM, n0 = (int)(id % tiles_n) * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wv / WN, wn = wv % WN;
    auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };

    // per-lane source of each staging pass (the swizzle lives here)
    const __bf16* a_src[A_PASSES]; const __bf16* b_src[B_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int p = i * NTH + t, row = p / CPR, c = (p % CPR) ^ swz(row);
        a_src[i] = g.A + (long long)min(m0 + row, g.M - 1) * g.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
        const int p = i * NTH + t, row = p / CPR, c = (p % CPR) ^ swz(row);
        b_src[i] = g.Bt + (long long)min(n0 + row, g.N - 1) * g.ldb + c * 8;
    }
    auto stage = [&](int kt, int slot) {
        char* base = smem + slot * BUF;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) glds16(a_src[i] + kt * BK, base + (i * NTH + wv * 64) * 16);
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) glds16(b_src[i] + kt * BK, base + A_BYTES + (i * NTH + wv * 64) * 16);
    };
    // fragment addresses: lane (r = lane & 31, h = lane >> 5) reads chunk 2 s + h of its row for k-step s
    const int r = lane & 31, h = lane >> 5;
    const int arow = wm * (BM / WM) + r, brow = wn * (BN / WN) + r;
    const int ax = swz(arow), bx = swz(brow);
    const int a_lo = arow * RB + ((h ^ ax) & 1) * 16, a_x6 = (ax & 6) * 16;
    const int b_lo = brow * RB + ((h ^ bx) & 1) * 16 + A_BYTES, b_x6 = (bx & 6) * 16;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int nt = g.K / BK;
    // dX with a ring of two: the relu-mask source (this wave's output tile of the layer's bf16 input, 16 x 16 bytes per lane) is
    // fetched UNDER the k-loop -- issued behind the second stage, complete at the next tile's wait, which is vmcnt(0) in a ring of
    // two anyway -- instead of in front of the epilogue, where its 64 MB per launch were 16 of the kernel's 59 us
    constexpr bool PRE_MASK = EPI == HS_DX && NS == 2;
    [[maybe_unused]] uint4 pmask[PRE_MASK ? TM : 1][PRE_MASK ? TN : 1][2];
#ifdef VAEK_HS_STAMPS
    unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_comp = 0;
#endif
    HS_STAMP(p0);
#pragma unroll
    for (int q = 0; q < NS - 1; ++q) if (q < nt) stage(q, q);
    HS_STAMP(p1);
    int slot = 0, fill = NS - 1;                          // ring positions of the tile being multiplied / being loaded
    for (int kt = 0; kt < nt; ++kt) {
        HS_STAMP(s0);
        wait_tiles<P, NS - 2>(min(NS - 2, nt - 1 - kt));   // this wave's share of tile kt has landed ...
        HS_STAMP(s1);
        __builtin_amdgcn_s_barrier();                      // ... and everyone's; everyone is done reading tile kt - 1
        HS_STAMP(s2);
        if (kt + NS - 1 < nt) stage(kt + NS - 1, fill);    // into the slot tile kt - 1 was read from
        if constexpr (PRE_MASK) {
            // (fragment (i, j) behind the stage of k-tile i TN + j: spread over the loop, each wait absorbs an eighth of the bytes)
            const int nb_ = n0 + wn * (BN / WN);
            if (g.aux && nb_ < g.N) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (kt == min(i * TN + j, nt - 1)) {
                            const int m = m0 + wm * (BM / WM) + i * 32 + (lane & 31);
                            const long long ro = (long long)(m < g.M ? m : g.M - 1) * g.ldc;
#pragma unroll
                            for (int u = 0; u < 2; ++u)
                                pmask[i][j][u] = *reinterpret_cast<const uint4*>(g.aux + ro + nb_ + j * 32 + 16 * u + 8 * (lane >> 5));
                        }
                    }
            }
        }
        HS_STAMP(s3);
        HS_ADD(st_wait, s0, s1); HS_ADD(st_bar, s1, s2); HS_ADD(st_issue, s2, s3);
        const char* base = smem + slot * BUF;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(base + a_lo + i * 32 * RB + ((32 * s) ^ a_x6));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(base + b_lo + j * 32 * RB + ((32 * s) ^ b_x6));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)      // swapped: D[n][m] -- lane & 31 = output row m, registers = columns n
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
        HS_STAMP(s4);
        HS_ADD(st_comp, s3, s4);
    }
    HS_STAMP(e0);
#ifdef VAEK_HS_STAMPS
    if (g_hs_stamp_buf && lane == 0) {
        unsigned long long* o = g_hs_stamp_buf + ((long long)blockIdx.x * (NTH / 64) + wv) * 8;
        o[0] = p1 - p0; o[1] = st_wait; o[2] = st_bar; o[3] = st_issue; o[4] = st_comp; o[5] = e0 - p0; o[6] = k0; o[7] = e0;
    }
#endif

    // ---- epilogue: register q of tile (i, j) = C[m0 + wm.. + 32 i + r][nb + 32 j + (q & 3) + 8 (q >> 2) + 4 h]
    const int nbase = n0 + wn * (BN / WN);
    if (nbase >= g.N) return;                      // wave-uniform: N is a multiple of 64 = the wave's column span
    const float floor_v = g.relu ? 0.f : -__builtin_huge_valf();      // relu as one v_max with a uniform floor: no branch per value
    float4 bias4[TN][4];
    if (EPI == HS_FWD) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) bias4[j][q] = *reinterpret_cast<const float4*>(g.bias + nbase + j * 32 + 8 * q + 4 * h);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 32 + r;
        const bool row_ok = m < g.M;
        const long long rowoff = (long long)(row_ok ? m : g.M - 1) * g.ldc;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nj = nbase + j * 32;
            // after the swaps lanes 0-31 own columns [8 qq, 8 qq + 8), lanes 32-63 [8 qq + 8, 8 qq + 16), qq = 0, 2
            uint4 mask[2];
            if (EPI == HS_DX && g.aux) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if constexpr (PRE_MASK) mask[u] = pmask[i][j][u];
                    else mask[u] = *reinterpret_cast<const uint4*>(g.aux + rowoff + nj + 16 * u + 8 * h);
                }
            }
            unsigned d[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) v[p] = acc[i][j][4 * q + p];
                if (EPI == HS_FWD) {
                    const float4 b = bias4[j][q];
#pragma unroll
                    for (int p = 0; p < 4; ++p) v[p] = fmaxf(v[p] + (p == 0 ? b.x : p == 1 ? b.y : p == 2 ? b.z : b.w), floor_v);
                }
                d[q][0] = pack_bf16(v[0], v[1]); d[q][1] = pack_bf16(v[2], v[3]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qq = 2 * u;
                const auto sx = __builtin_amdgcn_permlane32_swap(d[qq][0], d[qq + 1][0], false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(d[qq][1], d[qq + 1][1], false, false);
                uint4 o = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                if (EPI == HS_DX && g.aux) {
                    auto keep = [](unsigned a) {      // per bf16 half: 0xffff where the activation is > 0 (relu'), else 0
                        const unsigned lo = (int)(short)(a & 0xffffu) > 0 ? 0xffffu : 0u;
                        const unsigned hi = (int)a > 0xffff ? 0xffff0000u : 0u;
                        return lo | hi;
                    };
                    o.x &= keep(mask[u].x); o.y &= keep(mask[u].y); o.z &= keep(mask[u].z); o.w &= keep(mask[u].w);
                }
                if (row_ok) *reinterpret_cast<uint4*>(g.C + rowoff + nj + 16 * u + 8 * h) = o;
            }
        }
    }
#ifdef VAEK_HS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HS_STAMP(x1);
    if (g_hs_stamp_buf && lane == 0) g_hs_stamp_buf[((long long)gridDim.x * (NTH / 64) + (long long)blockIdx.x * (NTH / 64) + wv) * 8] = x1;
#endif
}

// Persistent, staggered form of hs_nt_kernel for the big layers: one workgroup of 8 waves per CU walks its output tiles.
//  * 256 x 256 tiles: a global_load_lds costs the issuing wave ~100-120 cycles per KB (tools/hs_stamps.sh), so a 128 x 128
//    tile (0.5 KB per 32-cycle MFMA) is issue-bound; 256 x 256 loads 0.25 KB per MFMA.
//  * stagger: every wave's k-tile is an issue phase (its share of the next tile's loads, ~500 cycles at BK = 32) and a
//    multiply phase (16 MFMAs, 512 cycles); the two waves of a SIMD in lockstep leave the matrix pipe idle during the former
//    and fight over it during the latter.  Waves 4-7 run multiply-then-issue, waves 0-3 issue-then-multiply.
//  * the ring keeps filling across tile seams: the first stages of the NEXT tile are issued before the epilogue of this one,
//    so the epilogue's stores run beside loads already in flight and no tile starts on an exposed HBM round trip.
template <int EPI, int BM, int BN, int WM, int WN, int BK, int NS>
__global__ __launch_bounds__(64 * WM * WN) void hs_nt_p_kernel(const HsArgs g) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RB = 2 * BK, CPR = BK / 8;
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, BUF = A_BYTES + B_BYTES;
    constexpr int A_PASSES = BM * CPR / NTH, B_PASSES = BN * CPR / NTH, P = A_PASSES + B_PASSES;
    static_assert(BM * CPR % NTH == 0 && BN * CPR % NTH == 0, "whole staging passes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wv / WN, wn = wv % WN;
    const bool late = wv >= WM * WN / 2;                 // wave-uniform: multiply first, issue afterwards
    auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };
    // this workgroup's j-th tile; each XCD (workgroups b with equal b % 8) walks a contiguous run of the n-fastest tile space
    const int G = gridDim.x, b = blockIdx.x;
    const int my_tiles = b < ntiles ? (ntiles - b + G - 1) / G : 0;
    auto tile_of = [&](int j, int& m0, int& n0) {
        int L = b + j * G;
        if (ntiles % 8 == 0 && G % 8 == 0) L = (b % 8) * (ntiles / 8) + (b / 8) + j * (G / 8);
        m0 = (L / tiles_n) * BM; n0 = (L % tiles_n) * BN;
    };
    // staging positions of this thread (row, logical chunk) per pass: the swizzle lives in the source address
    int a_row[A_PASSES], a_c[A_PASSES], b_row[B_PASSES], b_c[B_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) { const int p = i * NTH + t; a_row[i] = p / CPR; a_c[i] = ((p % CPR) ^ swz(p / CPR)) * 8; }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) { const int p = i * NTH + t; b_row[i] = p / CPR; b_c[i] = ((p % CPR) ^ swz(p / CPR)) * 8; }
    const __bf16* a_src[A_PASSES]; const __bf16* b_src[B_PASSES];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) a_src[i] = g.A + (long long)min(m0 + a_row[i], g.M - 1) * g.lda + a_c[i];
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) b_src[i] = g.Bt + (long long)min(n0 + b_row[i], g.N - 1) * g.ldb + b_c[i];
    };
    auto stage = [&](int kt, int slot) {
        char* base = smem + slot * BUF;
#if defined(VAEK_HS_ABLATE) && (VAEK_HS_ABLATE & 4)      // diagnostic: no operand loads at all
        (void)base; (void)kt;
#else
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) glds16(a_src[i] + kt * BK, base + (i * NTH + wv * 64) * 16);
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) glds16(b_src[i] + kt * BK, base + A_BYTES + (i * NTH + wv * 64) * 16);
#endif
    };
    const int r = lane & 31, h = lane >> 5;
    const int arow = wm * (BM / WM) + r, brow = wn * (BN / WN) + r;
    const int ax = swz(arow), bx = swz(brow);
    const int a_lo = arow * RB + ((h ^ ax) & 1) * 16, a_x6 = (ax & 6) * 16;
    const int b_lo = brow * RB + ((h ^ bx) & 1) * 16 + A_BYTES, b_x6 = (bx & 6) * 16;
    const float floor_v = g.relu ? 0.f : -__builtin_huge_valf();

    // the bias of the tile being multiplied, loaded when the tile starts: a load issued in the epilogue would be YOUNGER than the
    // next tile's stages already in flight, and vmcnt retires in order -- using it would drain the ring first
    float4 bias4[TN][4];
    auto load_bias = [&](int n0) {
        if (EPI != HS_FWD) return;
        const int nbase = min(n0 + wn * (BN / WN), g.N - BN / WN);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) bias4[j][q] = *reinterpret_cast<const float4*>(g.bias + nbase + j * 32 + 8 * q + 4 * h);
    };
    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    };
    auto epilogue = [&](int m0, int n0) {
        const int nbase = n0 + wn * (BN / WN);
        if (nbase >= g.N) return;                      // wave-uniform: N is a multiple of 64 = the wave's column span
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (BM / WM) + i * 32 + r;
            const bool row_ok = m < g.M;
            const long long rowoff = (long long)(row_ok ? m : g.M - 1) * g.ldc;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nj = nbase + j * 32;
                uint4 mask[2];
                if (EPI == HS_DX && g.aux) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) mask[u] = *reinterpret_cast<const uint4*>(g.aux + rowoff + nj + 16 * u + 8 * h);
                }
                unsigned d[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) v[p] = acc[i][j][4 * q + p];
                    if (EPI == HS_FWD) {
                        const float4 bq = bias4[j][q];
#pragma unroll
                        for (int p = 0; p < 4; ++p) v[p] = fmaxf(v[p] + (p == 0 ? bq.x : p == 1 ? bq.y : p == 2 ? bq.z : bq.w), floor_v);
                    }
                    d[q][0] = pack_bf16(v[0], v[1]); d[q][1] = pack_bf16(v[2], v[3]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int qq = 2 * u;
                    const auto sx = __builtin_amdgcn_permlane32_swap(d[qq][0], d[qq + 1][0], false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(d[qq][1], d[qq + 1][1], false, false);
                    uint4 o = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                    if (EPI == HS_DX && g.aux) {
                        auto keep = [](unsigned a) {
                            const unsigned lo = (int)(short)(a & 0xffffu) > 0 ? 0xffffu : 0u;
                            const unsigned hi = (int)a > 0xffff ? 0xffff0000u : 0u;
                            return lo | hi;
                        };
                        o.x &= keep(mask[u].x); o.y &= keep(mask[u].y); o.z &= keep(mask[u].z); o.w &= keep(mask[u].w);
                    }
#if defined(VAEK_HS_ABLATE) && (VAEK_HS_ABLATE & 2)      // diagnostic: no output stores (values kept alive)
                    asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
#else
                    if (row_ok) *reinterpret_cast<uint4*>(g.C + rowoff + nj + 16 * u + 8 * h) = o;
#endif
                }
            }
        }
    };

    // ---- one flat stream of k-tiles over all of this workgroup's output tiles: q = tile j * nt + kt --------------------
    const int nt = g.K / BK, Q = my_tiles * nt;
    if (Q == 0) return;
    int lj = 0, lkt = 0;                                 // load cursor: tile index / k-tile of the NEXT stage to issue
    int lm0, ln0; tile_of(0, lm0, ln0); set_sources(lm0, ln0);
    int cm0 = lm0, cn0 = ln0, ckt = 0;                   // compute cursor
    auto issue_next = [&](int slot) {                    // issue stage number `issued`, advance the load cursor
        stage(lkt, slot);
        if (++lkt == nt) { lkt = 0; ++lj; if (lj < my_tiles) { tile_of(lj, lm0, ln0); set_sources(lm0, ln0); } }
    };
    int issued = 0;
#pragma unroll
    for (int qq = 0; qq < NS - 1; ++qq) if (issued < Q) { issue_next(qq); ++issued; }
    int slot = 0, fill = NS - 1, cj = 0;
    zero_acc();
    load_bias(cn0);
    for (int q = 0; q < Q; ++q) {
        // loads of the stores of an earlier epilogue count in vmcnt too, but they are OLDER than every tile in flight, so
        // "at most P * newer outstanding" still implies tile q has landed
        wait_tiles<P, NS - 2>(min(NS - 2, Q - 1 - q));
        __builtin_amdgcn_s_barrier();
        const bool more = issued < Q;
        if (!late && more) issue_next(fill);
        const char* base = smem + slot * BUF;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(base + a_lo + i * 32 * RB + ((32 * s) ^ a_x6));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(base + b_lo + j * 32 * RB + ((32 * s) ^ b_x6));
#if defined(VAEK_HS_ABLATE) && (VAEK_HS_ABLATE & 1)      // diagnostic (tools/hs_ablate.sh): no MFMAs, operands kept alive
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j]));
#else
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
#endif
        }
        if (late && more) issue_next(fill);
        if (more) ++issued;
        slot = slot + 1 == NS ? 0 : slot + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
        if (++ckt == nt) {                               // this output tile is complete (the next one's loads are already in flight)
            epilogue(cm0, cn0);
            zero_acc();
            ckt = 0; ++cj;
            if (cj < my_tiles) { tile_of(cj, cm0, cn0); load_bias(cn0); }
        }
    }
}

// ---- dW|db ------------------------------------------------------------------------------------------------------------
struct HsDwArgs {
    const __bf16* X; const __bf16* dY; float* slab;   // X [rows, M] (ldx), dY [rows, N] (ldy), slab[split][(M + 1) x N]
    int rows, M, N, ldx, ldy, rows_per_split; long long slab_stride;
    int tiles_m, tiles_n;
    // CONV (the kernel gradient of a 4x4 / stride 2 / pad 1 convolution, conv_bf16.hip): the "rows" are the output pixels of an
    // NHWC image batch, column m = (kh, kw, c) of row p is x[n, 2 i + kh - 1, 2 j + kw - 1, c] -- gathered by the loader, 8
    // channels (16 bytes) per lane, from `zeros` where the tap falls outside the image; Ho, Wo powers of two, Cin a multiple of 8
    int H, W, Cin, sh_hw, sh_w; const __bf16* zeros;
};

// LDS image of a BK-row(k) x 128-feature tile: 256-byte rows, 16-byte chunk ch of row k at chunk position
// ch ^ (((k & 3) << 2) | ((k >> 2) & 3)): the 32 lanes of a ds_read_b64_tr_b16 half (4 k-rows x 2 blocks of 16
// features) then cover the 256-byte bank row exactly.
// NARROW (N <= 64): the four waves stack along M (32 rows x 64 columns each) instead of 2 x 2 -- no wave multiplies columns past N
// BM 256 (eight waves, 4 x 2): the M side is TWO 128-feature images side by side -- 1.5 KB of operands per k-row for twice the
// products of the 128 x 128 tile's 1 KB
// LW > 0 (BM 256): LW extra waves do nothing but issue the LDS-DMA pieces (and wait for them); the eight multiplying waves never
// spend issue slots on loads -- a global_load_lds costs the issuing wave ~100 cycles per KB, and in-order waves add that to their
// MFMA time (the ablation of the forward GEMM: loads 13 us, MFMAs 11 us, stores 12 us of a 48 us launch, nearly additive).
template <int BK, int NS, bool CONV = false, bool NARROW = false, int BM = 128, int LW = 0>
__global__ __launch_bounds__(2 * BM + 64 * LW) void hs_tn_kernel(const HsDwArgs g) {
    constexpr int BN = 128, T_BYTES = BK * 256, NTH = 2 * BM, AT = BM / 128, BUF = (AT + 1) * T_BYTES;
    constexpr int STH = LW ? 64 * LW : NTH;                   // threads that stage
    constexpr int PPT = BK * 16 / STH, P = (AT + 1) * PPT;    // staging passes per 128-feature image; loads per k-tile and (staging) wave
    constexpr int PASSES = AT * PPT;                          // ... of the A side
    static_assert(BK * 16 % STH == 0 && (BM == 128 || (!CONV && !NARROW)) && (LW == 0 || BM == 256), "whole staging passes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned tiles = g.tiles_m * g.tiles_n, nb = gridDim.x;
    unsigned id = blockIdx.x;
    if (nb % 8 == 0) id = (id % 8) * (nb / 8) + id / 8;        // the tiles of one split run on one XCD: they share X / dY rows
    const int split = id / tiles, tile = id % tiles;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * g.rows_per_split, kend = min(g.rows, kbeg + g.rows_per_split);
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int TI = NARROW ? 1 : 2, WROWS = NARROW ? 32 : 64;      // m-fragments per wave, rows of M per wave
    const int wm = NARROW ? wv : wv >> 1, wn = NARROW ? 0 : wv & 1;
    const bool stager = LW == 0 || wv >= NTH / 64;                    // does this wave issue loads?
    const int st = LW ? t - NTH : t, swv = LW ? wv - NTH / 64 : wv;   // its index among the staging threads / waves

    // staging: 16 BK chunks per 128-feature image; position p -> k-row p >> 4, chunk slot p & 15; pass i of the A side belongs to
    // image i / PPT
    int a_col[PASSES], b_col[PPT], s_row[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = i * STH + (st & (STH - 1)), row = p >> 4, ch = (p & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        s_row[i] = row;
        b_col[i] = min(n0 + ch * 8, g.N - 8);        // a partial last tile re-reads valid columns; those outputs are not stored
#pragma unroll
        for (int im = 0; im < AT; ++im) a_col[im * PPT + i] = min(m0 + 128 * im + ch * 8, g.M - 8);
    }
    // CONV: this thread's chunk of 8 columns is 8 channels of ONE tap, fixed for the whole kernel
    int c_dy[PASSES], c_dx[PASSES], c_ch[PASSES];
    if (CONV) {
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int tap = a_col[i] / g.Cin;
            c_dy[i] = (tap >> 2) - 1; c_dx[i] = (tap & 3) - 1; c_ch[i] = a_col[i] % g.Cin;
        }
    }
    auto stage = [&](int k0, int slot) {
        char* base = smem + slot * BUF;
        if constexpr (BM != 128) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const long long krow = min(k0 + s_row[i], g.rows - 1);
#pragma unroll
                for (int im = 0; im < AT; ++im)
                    glds16(g.X + krow * g.ldx + a_col[im * PPT + i], base + im * T_BYTES + (i * STH + swv * 64) * 16);
                glds16(g.dY + krow * g.ldy + b_col[i], base + AT * T_BYTES + (i * STH + swv * 64) * 16);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const long long krow = min(k0 + s_row[i], g.rows - 1);
            if (CONV) {
                const int pix = (int)krow, n = pix >> g.sh_hw, ij = pix & ((1 << g.sh_hw) - 1);
                const int yy = 2 * (ij >> g.sh_w) + c_dy[i], xx = 2 * (ij & ((1 << g.sh_w) - 1)) + c_dx[i];
                const bool in = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                const __bf16* src = in ? g.X + (((long long)n * g.H + yy) * g.W + xx) * g.Cin + c_ch[i] : g.zeros;
                glds16(src, base + (i * 256 + wv * 64) * 16);
            } else {
                glds16(g.X + krow * g.ldx + a_col[i], base + (i * 256 + wv * 64) * 16);
            }
            glds16(g.dY + krow * g.ldy + b_col[i], base + T_BYTES + (i * 256 + wv * 64) * 16);
        }
    };
    // transposed fragment reads: lane 4 q + p of a 16-lane group addresses k-row q, features 4 p .. 4 p + 3 of its block;
    // the group receives features block + (lane & 15), four consecutive k each
    const int h = lane >> 5, g16 = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    int a_addr[2][2], b_addr[2][2];                 // [fragment i / j][u: k rows 4 u .. 4 u + 3 of the lane half's 8]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int krow = 8 * h + 4 * u + q, xr = (q << 2) | ((2 * h + u) & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mfull = wm * WROWS + (i % TI) * 32 + 16 * g16 + 4 * p, mcol = mfull & 127;
            a_addr[i][u] = (mfull >> 7) * T_BYTES + krow * 256 + (((mcol >> 3) ^ xr) << 4) + ((mcol >> 2) & 1) * 8;
            const int ncol = wn * 64 + i * 32 + 16 * g16 + 4 * p;
            b_addr[i][u] = krow * 256 + (((ncol >> 3) ^ xr) << 4) + ((ncol >> 2) & 1) * 8 + AT * T_BYTES;
        }
    }
    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; accb[i][r] = 0.f; }
    }
    const bool do_bias = tm == 0 && wm == 0;         // wave-uniform
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    auto tr_read = [&](int addr) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(smem + addr));
    };
    const int nt = (kend - kbeg + BK - 1) / BK;
    if (stager) {
#pragma unroll
        for (int qq = 0; qq < NS - 1; ++qq) if (qq < nt) stage(kbeg + qq * BK, qq);
    }
    int slot = 0, fill = NS - 1;
    if constexpr (LW > 0) {
        if (stager) {                               // the loading waves' whole life: wait, meet the others, issue the tile NS - 1 ahead
            for (int kt = 0; kt < nt; ++kt) {
                wait_tiles<P, NS - 2>(min(NS - 2, nt - 1 - kt));
                __builtin_amdgcn_s_barrier();
                if (kt + NS - 1 < nt) stage(kbeg + (kt + NS - 1) * BK, fill);
                if (kend - (kbeg + kt * BK) < BK) __syncthreads();       // (the others zero the rows past the end: see below)
                fill = fill + 1 == NS ? 0 : fill + 1;
            }
            return;
        }
    }
    for (int kt = 0; kt < nt; ++kt) {
        if constexpr (LW == 0) wait_tiles<P, NS - 2>(min(NS - 2, nt - 1 - kt));
        __builtin_amdgcn_s_barrier();
        if constexpr (LW == 0) { if (kt + NS - 1 < nt) stage(kbeg + (kt + NS - 1) * BK, fill); }
        const int boff = slot * BUF;
        const int valid = kend - (kbeg + kt * BK);
        if (valid < BK) {                           // last k-tile of the batch: rows past the end must contribute zero
            for (int o = t * 16; o < BUF; o += NTH * 16)
                if (((o & (T_BYTES - 1)) >> 8) >= valid) *reinterpret_cast<uint4*>(smem + boff + o) = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 fa[TI], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i < TI) {
                    const bf16x4 a0 = tr_read(boff + a_addr[i][0] + s * 4096), a1 = tr_read(boff + a_addr[i][1] + s * 4096);
                    fa[i % TI] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                }
                const bf16x4 b0 = tr_read(boff + b_addr[i][0] + s * 4096), b1 = tr_read(boff + b_addr[i][1] + s * 4096);
                fb[i] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < 2; ++j) accb[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, fb[j], accb[j], 0, 0, 0);
            }
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
    }
    // ---- slab store: register r of tile (i, j) = G[m0 + 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 h][n0 + 64 wn + 32 j + (lane & 31)]
    float* C = g.slab + (long long)split * g.slab_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.M) C[(long long)m * g.N + n] = acc[i][j][r];
            }
        if (do_bias && h == 0) C[(long long)g.M * g.N + n] = accb[j][0];      // every row of 1 . dY is the column sum
    }
}

// ---- convolution forward / transposed forward on the same loader (conv_bf16.hip) --------------------------------------------------
// hs_nt_kernel's main loop with the A operand GATHERED: GEMM row m is an output pixel (FWD: of the stride-2 convolution; T: an
// input-grid pixel of one output parity class, blockIdx.y), k = (tap, channel); every 16-byte chunk (8 channels) is one
// global_load_lds from the bf16 NHWC image, or from a page of zeros where the tap falls outside it.  B: FWD the [N][K] transposed
// bf16 copy of the kernel; T the kernel as stored [4][4][C_out][C_in], the class's four [o][c] slices picked per chunk.
// Epilogue: + bias, relu, relu-mask of the layer below, float32 out (16 bytes per lane and register quad), scattered to the class's
// pixels for T.  C_in a power of two (>= 8 FWD, >= 16 T), C_out a multiple of BN.
enum { HC_FWD = 0, HC_T = 1 };
struct HsConvArgs {
    const __bf16* X; const __bf16* Wb; const __bf16* zeros;
    float* out; const float* bias; const float* mask; __bf16* out16;     // out16: optional bf16 copy of out (the next layer's operand); out may be null when out16 is given
    const __bf16* mask16;                     // the relu-mask source as its bf16 copy (instead of mask)
    int relu, H, W, Cin, Cout, M, N, K;       // H, W: the INPUT tensor's spatial size
    int sh_c, hw, wo;                         // log2 C_in; GEMM rows per image and per image row (FWD: Ho Wo, Wo; T: H W, W)
    int sh_hw, sh_wo;                         // their log2 where they are powers of two (else -1): a run-time integer division is ~35 vector
                                              // instructions, and a thread of the 256-row tile did twenty of them in front of four k-tiles
};
template <int MODE, int BN, int WM, int WN, int BK, int NS, int BM = 128>
__global__ __launch_bounds__(64 * WM * WN) void hs_conv_kernel(const HsConvArgs g) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RB = 2 * BK, CPR = BK / 8;
    constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, BUF = A_BYTES + B_BYTES;
    constexpr int A_PASSES = BM * CPR / NTH, B_PASSES = BN * CPR / NTH, P = A_PASSES + B_PASSES;
    static_assert(BM * CPR % NTH == 0 && BN * CPR % NTH == 0 && B_PASSES >= 1, "whole staging passes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // T: the four parity classes of a tile are neighbours in the launch order -- they read the same input pixels (one HBM read, three L2 hits)
    // XCD-aware order (workgroups are dealt round-robin over the 8 XCDs, each with its own L2): every XCD gets a contiguous run of
    // the tile space, so the four classes / the column tiles of one row tile, and the row tiles that share window rows, meet in ONE L2
    // (PMC: the 32-column transposed layers read their input 4x from HBM when the classes were merely adjacent in launch order)
    unsigned lid = blockIdx.x;
    if (gridDim.x % 8 == 0) lid = (lid % 8) * (gridDim.x / 8) + lid / 8;
    const unsigned tiles_n = g.N / BN, bid = MODE == HC_T ? lid >> 2 : lid, cls = MODE == HC_T ? lid & 3 : 0;
    const int m0 = (int)(bid / tiles_n) * BM, n0 = (int)(bid % tiles_n) * BN;
    const int pp = (int)(cls >> 1), qq = (int)(cls & 1);
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wv / WN, wn = wv % WN;
    auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };
    const int cmask = g.Cin - 1;
    const bool pow2 = g.sh_hw >= 0 && g.sh_wo >= 0;      // (uniform: config 5's image sizes)

    int a_n[A_PASSES], a_y[A_PASSES], a_x[A_PASSES], a_c[A_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int p = i * NTH + t, row = p / CPR;
        a_c[i] = ((p % CPR) ^ swz(row)) * 8;
        const int m = min(m0 + row, g.M - 1);
        int n, ij, ii, jj;
        if (pow2) { n = m >> g.sh_hw; ij = m & (g.hw - 1); ii = ij >> g.sh_wo; jj = ij & (g.wo - 1); }
        else { n = m / g.hw; ij = m % g.hw; ii = ij / g.wo; jj = ij % g.wo; }
        a_n[i] = n * g.H * g.W;
        a_y[i] = MODE == HC_FWD ? 2 * ii - 1 : ii + pp;
        a_x[i] = MODE == HC_FWD ? 2 * jj - 1 : jj + qq;
    }
    const __bf16* b_src[B_PASSES]; int b_c[B_PASSES], b_row[B_PASSES];
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
        const int p = i * NTH + t, row = p / CPR;
        b_c[i] = ((p % CPR) ^ swz(row)) * 8;
        b_row[i] = min(n0 + row, g.N - 1);
        b_src[i] = g.Wb + (long long)b_row[i] * g.K + b_c[i];
    }
    auto stage = [&](int kt, int slot) {
        char* base = smem + slot * BUF;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int k = kt * BK + a_c[i], tap = k >> g.sh_c, ch = k & cmask;
            const int yy = MODE == HC_FWD ? a_y[i] + (tap >> 2) : a_y[i] - (tap >> 1);
            const int xx = MODE == HC_FWD ? a_x[i] + (tap & 3) : a_x[i] - (tap & 1);
            const bool in = (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
            const __bf16* src = in ? g.X + (((long long)(a_n[i] + yy * g.W + xx)) << g.sh_c) + ch : g.zeros;
            glds16(src, base + (i * NTH + wv * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const __bf16* src;
            if (MODE == HC_FWD) src = b_src[i] + kt * BK;
            else {
                const int k = kt * BK + b_c[i], tap = k >> g.sh_c, ch = k & cmask;
                const int kh = 1 - pp + 2 * (tap >> 1), kw = 1 - qq + 2 * (tap & 1);
                src = g.Wb + ((long long)((kh * 4 + kw) * g.Cout + b_row[i]) << g.sh_c) + ch;
            }
            glds16(src, base + A_BYTES + (i * NTH + wv * 64) * 16);
        }
    };
    const int r = lane & 31, h = lane >> 5;
    const int arow = wm * (BM / WM) + r, brow = wn * (BN / WN) + r;
    const int ax = swz(arow), bx = swz(brow);
    const int a_lo = arow * RB + ((h ^ ax) & 1) * 16, a_x6 = (ax & 6) * 16;
    const int b_lo = brow * RB + ((h ^ bx) & 1) * 16 + A_BYTES, b_x6 = (bx & 6) * 16;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // the epilogue's row offsets; with a small register tile the relu mask is fetched NOW, under the whole main loop (a workgroup of
    // these shapes sees only 4-16 k-tiles: a mask fetched in the epilogue is one more exposed HBM round trip per tile)
    const int nbase = n0 + wn * (BN / WN);
    constexpr bool kEarlyMask = TM * TN <= 2;
    long long offs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = min(m0 + wm * (BM / WM) + i * 32 + r, g.M - 1);
        if (MODE == HC_FWD) offs[i] = (long long)m * g.N;
        else {
            int n, ij, ii, jj;
            if (pow2) { n = m >> g.sh_hw; ij = m & (g.hw - 1); ii = ij >> g.sh_wo; jj = ij & (g.wo - 1); }
            else { n = m / g.hw; ij = m % g.hw; ii = ij / g.wo; jj = ij % g.wo; }
            offs[i] = (((long long)n * 2 * g.H + 2 * ii + pp) * 2 * g.W + 2 * jj + qq) * g.N;
        }
    }
    float4 mk[kEarlyMask ? TM : 1][kEarlyMask ? TN : 1][4];
    auto mask_at = [&](long long e) {                      // four mask values at element e: float32, or the bf16 copy widened
        if (g.mask16) {
            const uint2 u = *reinterpret_cast<const uint2*>(g.mask16 + e);
            return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                               __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
        }
        return *reinterpret_cast<const float4*>(g.mask + e);
    };
    const bool masked = g.mask || g.mask16;
    if (kEarlyMask && masked) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) mk[i][j][q] = mask_at(offs[i] + nbase + 32 * j + 8 * q + 4 * h);
    }

    const int nt = g.K / BK;
#pragma unroll
    for (int q = 0; q < NS - 1; ++q) if (q < nt) stage(q, q);
    int slot = 0, fill = NS - 1;
    for (int kt = 0; kt < nt; ++kt) {
        wait_tiles<P, NS - 2>(min(NS - 2, nt - 1 - kt));
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nt) stage(kt + NS - 1, fill);
        const char* base = smem + slot * BUF;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(base + a_lo + i * 32 * RB + ((32 * s) ^ a_x6));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(base + b_lo + j * 32 * RB + ((32 * s) ^ b_x6));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
        fill = fill + 1 == NS ? 0 : fill + 1;
    }

    // ---- epilogue: register 4 q + p of tile (i, j) = C[row][nbase + 32 j + 8 q + 4 h + p]
    const float floor_v = g.relu ? 0.f : -__builtin_huge_valf();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (BM / WM) + i * 32 + r;
        if (m >= g.M) continue;
        const long long off = offs[i];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = nbase + 32 * j + 8 * q + 4 * h;
                float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (g.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(g.bias + col);
                    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                }
                v.x = fmaxf(v.x, floor_v); v.y = fmaxf(v.y, floor_v); v.z = fmaxf(v.z, floor_v); v.w = fmaxf(v.w, floor_v);
                if (masked) {
                    const float4 k4 = kEarlyMask ? mk[i][j][q] : mask_at(off + col);
                    v.x = k4.x > 0.f ? v.x : 0.f; v.y = k4.y > 0.f ? v.y : 0.f; v.z = k4.z > 0.f ? v.z : 0.f; v.w = k4.w > 0.f ? v.w : 0.f;
                }
                if (g.out) *reinterpret_cast<float4*>(g.out + off + col) = v;
                if (g.out16) *reinterpret_cast<uint2*>(g.out16 + off + col) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
            }
    }
}

// bf16 [N][K] copy of a float32 [K][N] array (the convolution kernel as the k-contiguous B operand)
__global__ __launch_bounds__(256) void cvt_bf16_t_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int K, int N, __bf16* zero8) {
    if (zero8 && blockIdx.x == 0 && threadIdx.x < 8) zero8[threadIdx.x] = (__bf16)0.f;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < (long long)K * N) {
        const int n = (int)(e / K), k = (int)(e % K);
        dst[e] = (__bf16)src[(long long)k * N + n];
    }
}

// ---- bf16 copies of the wide layers' kernels ------------------------------------------------------------------------
struct CvtArgs {
    int n; int K[24], N[24]; long long w_off[24], out_off[24];     // out: W as stored [K, N], then W^T [N, K] (bf16 elements)
};
__global__ __launch_bounds__(256) void cvt_w_kernel(const float* __restrict__ params, __bf16* __restrict__ out, const CvtArgs a) {
    __shared__ float tile[64][65];
    const int l = blockIdx.y, K = a.K[l], N = a.N[l];
    const int tn = N / 64, tk = K / 64;
    if ((int)blockIdx.x >= tn * tk) return;
    const int k0 = (blockIdx.x / tn) * 64, n0 = (blockIdx.x % tn) * 64;
    const float* w = params + a.w_off[l];
    __bf16* wb = out + a.out_off[l];
    __bf16* wt = wb + (long long)K * N;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const float v = w[(long long)(k0 + r) * N + n0 + tx];
        tile[r][tx] = v;
        wb[(long long)(k0 + r) * N + n0 + tx] = (__bf16)v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) wt[(long long)(n0 + r) * K + k0 + tx] = (__bf16)tile[tx][r];
}

// ---- launchers ----------------------------------------------------------------------------------------------------------
// Tile / ring variants (tools/hs_tune.py times them on the box through vaek_debug_hs_gemm; g_hs_variant_* pick the default).
typedef void (*HsNtKernel)(const HsArgs);
struct HsNtVariant { HsNtKernel fwd, dx; int bm, bn, nth; size_t lds; const char* name; int persistent; };
#define HS_NTP(BM, BN, WM, WN, BK, NS) \
    {hs_nt_p_kernel<HS_FWD, BM, BN, WM, WN, BK, NS>, hs_nt_p_kernel<HS_DX, BM, BN, WM, WN, BK, NS>, BM, BN, 64 * WM * WN, \
     (size_t)NS * (BM + BN) * 2 * BK, #BM "x" #BN "x" #BK " ring" #NS " persistent staggered", 1}
#define HS_NT(BM, BN, WM, WN, BK, NS) \
    {hs_nt_kernel<HS_FWD, BM, BN, WM, WN, BK, NS>, hs_nt_kernel<HS_DX, BM, BN, WM, WN, BK, NS>, BM, BN, 64 * WM * WN, \
     (size_t)NS * (BM + BN) * 2 * BK, #BM "x" #BN "x" #BK " ring" #NS, 0}
static const HsNtVariant kNt[] = {
    HS_NT(128, 128, 2, 2, 64, 2),      // 0: one tile of prefetch, 64 KB, two workgroups per CU (the first version)
    HS_NT(128, 128, 2, 2, 32, 5),      // 1: 16 KB stages, four in flight, 80 KB, two workgroups per CU
    HS_NT(128, 128, 2, 2, 64, 4),      // 2: 32 KB stages, three in flight, 128 KB, one workgroup per CU
    HS_NT(256, 128, 4, 2, 64, 3),      // 3: 48 KB stages, two in flight, 144 KB, 8 waves
    HS_NT(256, 128, 4, 2, 32, 6),      // 4: 24 KB stages, five in flight, 144 KB, 8 waves
    HS_NT(128, 128, 2, 2, 32, 4),      // 5: 16 KB stages, three in flight, 64 KB, two workgroups per CU
    HS_NT(128, 128, 2, 2, 32, 3),      // 6: 48 KB: three workgroups per CU
    // a global_load_lds costs the issuing wave ~100 cycles per KB (in-kernel stamps, tools/hs_stamps.sh): a 128 x 128 tile
    // loads 0.5 KB per 32-cycle MFMA -- issue-bound; 256 x 256 loads 0.25 KB per MFMA
    HS_NT(256, 256, 2, 4, 64, 2),      // 7: 128 KB, one workgroup of 8 waves per CU, wave tile 128 x 64
    HS_NT(256, 256, 4, 2, 64, 2),      // 8: wave tile 64 x 128
    HS_NT(256, 256, 2, 4, 32, 4),      // 9: 32 KB stages, three in flight
    HS_NT(256, 128, 4, 2, 64, 2),      // 10: 96 KB
    HS_NTP(256, 256, 2, 4, 32, 4),     // 11: persistent + staggered, 32 KB stages, three in flight, 128 KB
    HS_NTP(256, 256, 2, 4, 32, 5),     // 12: four in flight, 160 KB
    HS_NTP(256, 256, 4, 2, 32, 4),     // 13: wave tile 64 x 128
    HS_NTP(256, 128, 4, 2, 32, 4),     // 14: 96 KB
};
constexpr int kNtCount = sizeof(kNt) / sizeof(kNt[0]);
typedef void (*HsTnKernel)(const HsDwArgs);
struct HsTnVariant { HsTnKernel fn; size_t lds; const char* name; int bm; int nth = 0; };      // nth: threads per workgroup (0: 2 bm)
static const HsTnVariant kTn[] = {
    {hs_tn_kernel<64, 2>, 2 * 2 * 64 * 256, "128x128x64 ring2", 128},      // 0: the first version
    {hs_tn_kernel<32, 5>, 5 * 2 * 32 * 256, "128x128x32 ring5", 128},      // 1
    {hs_tn_kernel<64, 4>, 4 * 2 * 64 * 256, "128x128x64 ring4", 128},      // 2: one workgroup per CU
    {hs_tn_kernel<32, 4>, 4 * 2 * 32 * 256, "128x128x32 ring4", 128},      // 3
    {hs_tn_kernel<32, 3>, 3 * 2 * 32 * 256, "128x128x32 ring3", 128},      // 4: three workgroups per CU
    {hs_tn_kernel<64, 2, false, false, 256>, 2 * 3 * 64 * 256, "256x128x64 ring2", 256},      // 5: eight waves, one workgroup per CU
    {hs_tn_kernel<32, 4, false, false, 256>, 4 * 3 * 32 * 256, "256x128x32 ring4", 256},      // 6
    {hs_tn_kernel<64, 3, false, false, 256>, 3 * 3 * 64 * 256, "256x128x64 ring3", 256},      // 7: 144 KB
    {hs_tn_kernel<64, 3, false, false, 256, 4>, 3 * 3 * 64 * 256, "256x128x64 ring3 + 4 loading waves", 256, 768},      // 8
    {hs_tn_kernel<64, 3, false, false, 256, 2>, 3 * 3 * 64 * 256, "256x128x64 ring3 + 2 loading waves", 256, 640},      // 9
};
constexpr int kTnCount = sizeof(kTn) / sizeof(kTn[0]);
// -1 = by shape: 256 x 256 tiles where they still give every CU a workgroup (C3 at full size: forward 50 -> 44 us), 128 x 128
// with three workgroups per CU otherwise; measured on the box (tools/hs_tune.py), every variant bitwise identical
// (tests/test_gpu_bf16.py::test_bf16s_variants_are_bitwise_identical)
int g_hs_variant_nt = -1, g_hs_variant_tn = -1;

template <int EPI>
static int hs_launch_nt(const HsArgs& g, const char* label, hipStream_t st) {
    if (g.M <= 0) return VAEK_OK;
    if (g.K % 64 || g.N % 64 || g.lda % 8 || g.ldb % 8 || g.ldc % 8) { set_error("bf16-storage GEMM needs widths that are multiples of 64"); return VAEK_ERR_INVALID; }
    const bool big = (long long)((g.M + 255) / 256) * ((g.N + 255) / 256) >= 256;
    const int vi = g_hs_variant_nt >= 0 ? g_hs_variant_nt : (big ? 7 : 6);
    if (vi < 0 || vi >= kNtCount) { set_error("unknown bf16-storage GEMM variant %d", vi); return VAEK_ERR_INVALID; }
    const HsNtVariant& v = kNt[vi];
    const HsNtKernel fn = EPI == HS_FWD ? v.fwd : v.dx;
    static thread_local PerDeviceOnce attr_set[2][kNtCount];
    if (attr_set[EPI][vi].need()) {
        VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
        attr_set[EPI][vi].mark();
    }
    unsigned grid = (unsigned)(((g.M + v.bm - 1) / v.bm) * ((g.N + v.bn - 1) / v.bn));
    if (v.persistent) grid = std::min(grid, 256u);         // one resident workgroup per CU walks the tiles
    ProfScope ps(label, st);
    launch_k(ps, fn, dim3(grid), dim3(v.nth), v.lds, st, g);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int launch_hs_fwd(const __bf16* x, const __bf16* wT, const float* b, __bf16* y, int rows, int n_in, int n_out, bool relu,
                  hipStream_t st) {
    HsArgs g{};
    g.A = x; g.Bt = wT; g.C = y; g.M = rows; g.N = n_out; g.K = n_in; g.lda = n_in; g.ldb = n_in; g.ldc = n_out;
    g.bias = b; g.relu = relu;
    return hs_launch_nt<HS_FWD>(g, "gemm_bf16s_fwd", st);
}
// dX[rows, n_in] = (dY[rows, n_out] . W^T) * (x_post > 0); W as stored [n_in, n_out] is the k-contiguous Bt operand
int launch_hs_dx(const __bf16* dy, const __bf16* w16, const __bf16* x_post, __bf16* dx, int rows, int n_in, int n_out,
                 hipStream_t st) {
    HsArgs g{};
    g.A = dy; g.Bt = w16; g.C = dx; g.M = rows; g.N = n_in; g.K = n_out; g.lda = n_out; g.ldb = n_out; g.ldc = n_in;
    g.aux = x_post;
    return hs_launch_nt<HS_DX>(g, "gemm_bf16s_dx", st);
}
int launch_hs_dw(const __bf16* x, const __bf16* dy, float* slab0, int64_t slab_stride, int S, int rows_per_split, int rows,
                 int n_in, int n_out, hipStream_t st) {
    if (rows <= 0) return VAEK_OK;
    if (n_in % 64 || n_out % 64 || rows_per_split % 64) { set_error("bf16-storage dW needs widths / splits that are multiples of 64"); return VAEK_ERR_INVALID; }
    HsDwArgs g{};
    g.X = x; g.dY = dy; g.slab = slab0; g.rows = rows; g.M = n_in; g.N = n_out; g.ldx = n_in; g.ldy = n_out;
    g.rows_per_split = rows_per_split; g.slab_stride = slab_stride;
    // -1 = by shape: 256 x 128 tiles (a quarter fewer operand bytes through L2 -> LDS per product: C3's 512 x 512 layers 75 -> 63 us,
    // 45 us with the LDS-DMA in inline assembly and a ring of three)
    // where they still give most CUs a workgroup, 128 x 128 otherwise; every variant bitwise identical for the same splits
    const bool big = n_in % 256 == 0 && (long long)(n_in / 256) * ((n_out + 127) / 128) * S >= 192;
    const int vi = g_hs_variant_tn >= 0 ? g_hs_variant_tn : (big ? 8 : 0);      // (8: ring of three 64-row k-tiles + four loading waves: 43 us; 7 without them: 45; ring of two: 52)
    if (vi < 0 || vi >= kTnCount) { set_error("unknown bf16-storage dW variant %d", vi); return VAEK_ERR_INVALID; }
    const HsTnVariant& v = kTn[vi];
    g.tiles_m = (n_in + v.bm - 1) / v.bm; g.tiles_n = (n_out + 127) / 128;
    static thread_local PerDeviceOnce attr_set[kTnCount];
    if (attr_set[vi].need()) {
        VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds));
        attr_set[vi].mark();
    }
    ProfScope ps("gemm_bf16s_dw", st);
    launch_k(ps, v.fn, dim3((unsigned)(g.tiles_m * g.tiles_n * S)), dim3(v.nth ? v.nth : 2 * v.bm), v.lds, st, g);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

// float32 -> bf16 copies (the convolution's kernel gradient reads its two tensors through the bf16-storage loader)
__global__ __launch_bounds__(256) void cvt_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long long n, __bf16* zero8) {
    if (zero8 && blockIdx.x == 0 && threadIdx.x < 8) zero8[threadIdx.x] = (__bf16)0.f;       // the loader's 16-byte page of zeros
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
        const bf16x8 o = {(__bf16)a.x, (__bf16)a.y, (__bf16)a.z, (__bf16)a.w, (__bf16)b.x, (__bf16)b.y, (__bf16)b.z, (__bf16)b.w};
        *reinterpret_cast<bf16x8*>(dst + i) = o;
    } else {
        for (long long k = i; k < n; ++k) dst[k] = (__bf16)src[k];
    }
}
int launch_cvt_bf16(const float* src, __bf16* dst, int64_t n, __bf16* zero8, hipStream_t st) {
    if (n <= 0 && !zero8) return VAEK_OK;                  // (n == 0 with a zero page: only the page is written)
    if (n < 0) n = 0;
    ProfScope ps("cvt_bf16", st);
    launch_k(ps, cvt_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256 + 1)), dim3(256), 0, st, src, dst, (long long)n, zero8);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

// kernel gradient of a 4x4 / stride 2 / pad 1 convolution on bf16 copies: slab[s][(16 Cin + 1) x Cout] (row 16 Cin: the bias gradient)
int launch_hs_conv_dw(const __bf16* x, const __bf16* dy, const __bf16* zeros, float* slab0, int64_t slab_stride, int S, int rows_per_split,
                      int batch, int H, int W, int Cin, int Cout, hipStream_t st) {
    const int Ho = H / 2, Wo = W / 2, hw = Ho * Wo;
    if (Cin % 8 || Cout % 8 || (hw & (hw - 1)) || (Wo & (Wo - 1))) { set_error("conv dW on the bf16-storage loader: unsupported shape"); return VAEK_ERR_INVALID; }
    HsDwArgs g{};
    g.X = x; g.dY = dy; g.slab = slab0; g.rows = batch * hw; g.M = 16 * Cin; g.N = Cout; g.ldx = 0; g.ldy = Cout;
    g.rows_per_split = rows_per_split; g.slab_stride = slab_stride;
    g.tiles_m = (g.M + 127) / 128; g.tiles_n = (g.N + 127) / 128;
    g.H = H; g.W = W; g.Cin = Cin; g.sh_hw = 31 - __builtin_clz(hw); g.sh_w = 31 - __builtin_clz(Wo); g.zeros = zeros;
    const bool narrow = Cout <= 64;
    const auto fn = narrow ? hs_tn_kernel<64, 2, true, true> : hs_tn_kernel<64, 2, true>;
    const size_t lds = 2 * 2 * 64 * 256;
    static thread_local PerDeviceOnce attr_set[2];
    if (attr_set[narrow].need()) { VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set[narrow].mark(); }
    ProfScope ps("conv_wgrad_bf16s", st);
    launch_k(ps, fn, dim3((unsigned)(g.tiles_m * g.tiles_n * S)), dim3(256), lds, st, g);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

// the convolution's forward (mode 0) / transposed forward (mode 1) on bf16 copies; shapes checked by the caller (conv_bf16.hip)
template <int MODE, int BN, int WM, int WN, int BK, int NS, int BM = 128>
static int hs_conv_launch(const HsConvArgs& g, hipStream_t st) {
    const auto fn = hs_conv_kernel<MODE, BN, WM, WN, BK, NS, BM>;
    constexpr size_t lds = (size_t)NS * (BM + BN) * 2 * BK;
    static thread_local PerDeviceOnce attr_set;
    if (attr_set.need()) { VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set.mark(); }
    ProfScope ps(MODE == HC_FWD ? "conv_fwd_bf16s" : "conv_t_fwd_bf16s", st);
    launch_k(ps, fn, dim3((unsigned)(((g.M + BM - 1) / BM) * (g.N / BN)) * (MODE == HC_T ? 4u : 1u)), dim3(64 * WM * WN), lds, st, g);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}
int launch_hs_conv(int mode, const __bf16* x, const __bf16* wb, const __bf16* zeros, const float* bias, const float* mask, const __bf16* mask16,
                   float* out, __bf16* out16, int batch, int H, int W, int Cin, int Cout, bool relu, hipStream_t st) {
    HsConvArgs g{};
    g.X = x; g.Wb = wb; g.zeros = zeros; g.out = out; g.out16 = out16; g.bias = bias; g.mask = mask; g.mask16 = mask16; g.relu = relu;
    g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.N = Cout; g.sh_c = 31 - __builtin_clz(Cin);
    if (mode == HC_FWD) { g.hw = (H / 2) * (W / 2); g.wo = W / 2; g.K = 16 * Cin; }
    else { g.hw = H * W; g.wo = W; g.K = 4 * Cin; }
    g.sh_hw = g.hw > 0 && (g.hw & (g.hw - 1)) == 0 ? 31 - __builtin_clz(g.hw) : -1;
    g.sh_wo = g.wo > 0 && (g.wo & (g.wo - 1)) == 0 ? 31 - __builtin_clz(g.wo) : -1;
    const long long M = (long long)batch * g.hw;
    if ((Cin & (Cin - 1)) || Cout % 32 || g.K % 64 || M > 0x7fffffffll || (long long)batch * H * W > 0x7fffffffll) {
        set_error("convolution on the bf16-storage loader: unsupported shape");
        return VAEK_ERR_INVALID;
    }
    g.M = (int)M;
    if (mode == HC_FWD) {
        if (Cout % 128 == 0) return hs_conv_launch<HC_FWD, 128, 2, 2, 64, 2>(g, st);
        if (Cout % 64 == 0) return hs_conv_launch<HC_FWD, 64, 2, 2, 64, 3>(g, st);
        return hs_conv_launch<HC_FWD, 32, 4, 1, 64, 2, 256>(g, st);     // short reductions, thin tiles: 256 rows per workgroup halve the per-tile latencies
    }
    if (Cout % 128 == 0) return hs_conv_launch<HC_T, 128, 2, 2, 64, 2>(g, st);
    if (Cout % 64 == 0) return hs_conv_launch<HC_T, 64, 2, 2, 64, 3>(g, st);
    return hs_conv_launch<HC_T, 32, 4, 1, 64, 2, 256>(g, st);
}
int launch_cvt_bf16_t(const float* src, __bf16* dst, int K, int N, __bf16* zero8, hipStream_t st) {
    ProfScope ps("cvt_bf16_t", st);
    launch_k(ps, cvt_bf16_t_kernel, dim3((unsigned)(((long long)K * N + 255) / 256)), dim3(256), 0, st, src, dst, K, N, zero8);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int launch_cvt_weights(const float* params, __bf16* out, const int* K, const int* N, const int64_t* w_off, const int64_t* out_off,
                       int n, hipStream_t st) {
    if (n <= 0) return VAEK_OK;
    if (n > 24) { set_error("too many wide layers"); return VAEK_ERR_INVALID; }
    CvtArgs a{};
    a.n = n;
    int maxt = 0;
    for (int i = 0; i < n; ++i) {
        a.K[i] = K[i]; a.N[i] = N[i]; a.w_off[i] = w_off[i]; a.out_off[i] = out_off[i];
        maxt = std::max(maxt, (K[i] / 64) * (N[i] / 64));
    }
    ProfScope ps("cvt_weights_bf16", st);
    launch_k(ps, cvt_w_kernel, dim3((unsigned)maxt, (unsigned)n), dim3(256), 0, st, params, out, a);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

}  // namespace vaek

// Diagnostic hook (not part of include/vaek.h; tools/hs_tune.py): pick the tile / ring variant of the bf16-storage GEMMs
// for this process (-1 keeps the current one) and report how many exist.
extern "C" int vaek_debug_hs_variant(int nt, int tn, int* n_nt, int* n_tn) {
    if (nt >= vaek::kNtCount || tn >= vaek::kTnCount) return VAEK_ERR_INVALID;
    if (nt >= 0) vaek::g_hs_variant_nt = nt;
    if (nt == -2) vaek::g_hs_variant_nt = -1;            // back to the by-shape default
    if (tn >= 0) vaek::g_hs_variant_tn = tn;
    if (n_nt) *n_nt = vaek::kNtCount;
    if (n_tn) *n_tn = vaek::kTnCount;
    return VAEK_OK;
}

#ifdef VAEK_HS_STAMPS
extern "C" int vaek_debug_hs_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(vaek::g_hs_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#endif
