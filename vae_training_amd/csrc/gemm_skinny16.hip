// The skinny ends of a bf16-storage stack (api.hip, Net::b16): the first layer d -> H and the last layer H -> d of an MLP
// whose hidden width H is wide and whose data / latent dimension d is a handful of features (every shipped experiment:
// d = 6, 7, 12, 20; networks.py:26-44 via vae.py:53-54).  Each of these layers touches ONE big tensor, the [B, H] bf16
// activation or gradient (67 MB at C3), so its roofline is that tensor's HBM stream; on the general f32 matrix-core kernel
// (gemm_f32.hip, d padded to a 32-wide MFMA tile, f32 MFMA = 1/16 of the bf16 rate) they ran 2-3x over it and together
// made up a third of the C3 bf16 step.  Four kernels, all reading / writing the big tensor in 16-byte pieces, fully coalesced:
//
//   sk_first_fwd    y[B,H] = relu(x[B,d] W[d,H] + b) as bf16.  VALU: a lane owns 8 output columns, the d x 8 weights live
//                   in its registers, x rows are broadcast loads.                                   (one 67 MB write)
//   sk_rows_mfma    C[B,d] = A[B,H] . Wp^T on v_mfma_f32_32x32x16_bf16 with A fragments straight from global memory (each
//                   element is used by exactly one MFMA: staging it through LDS would buy nothing) and Wp = the weights
//                   padded to 32 rows, bf16, k-contiguous.  Epilogues: bias (+ accumulate), reparameterisation
//                   (networks.py:73-74), the ELBO's elementwise pass (networks.py:81-83, 94-98).  Serves the last layer's
//                   forward AND the first layer's dX (A = dY).                                       (one 67 MB read)
//   sk_last_bwd     last layer backward, dX and dW|db FUSED: dh[B,H] = (dy[B,d] W^T) * (h > 0) as bf16 and G[(H+1),d] =
//                   [h | 1]^T dy from ONE pass over h.                                    (one 67 MB read, one 67 MB write)
//   sk_first_bwd    first layer dW|db: G[(d+1),H] = [x | 1]^T dY.                                    (one 67 MB read)
//
// Batch reductions (G) are per-workgroup register sums -> one partial image per workgroup -> sk_partials_reduce sums fixed
// groups of 64 partials in fixed order into the layer's S slabs (flat-gradient layout, as every other dW kernel writes them):
// no atomics, bitwise repeatable.
#include "vaek_internal.h"

namespace vaek {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ unsigned sk_pack(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void sk_unpack8(const uint4 u, float (&f)[8]) {      // bf16 -> f32 is a 16-bit shift
    f[0] = __builtin_bit_cast(float, u.x << 16); f[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, u.y << 16); f[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
    f[4] = __builtin_bit_cast(float, u.z << 16); f[5] = __builtin_bit_cast(float, u.z & 0xffff0000u);
    f[6] = __builtin_bit_cast(float, u.w << 16); f[7] = __builtin_bit_cast(float, u.w & 0xffff0000u);
}

// ---- first layer forward ------------------------------------------------------------------------------------------------
struct SkFwdArgs { const float* x; const float* w; const float* b; __bf16* y; int rows, d, H, relu; };

template <int DMAX>
__global__ __launch_bounds__(256) void sk_first_fwd_kernel(const SkFwdArgs a) {
    const int cprw = a.H >> 3, rpp = 256 / cprw, t = threadIdx.x;
    if (t >= rpp * cprw) return;
    const int chunk = t % cprw, rsub = t / cprw;
    float w[DMAX][8], b8[8];
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {                    // unconditional loads (clamped row), zeroed by a select afterwards
        const float4 lo = *reinterpret_cast<const float4*>(a.w + (long long)min(i, a.d - 1) * a.H + 8 * chunk);
        const float4 hi = *reinterpret_cast<const float4*>(a.w + (long long)min(i, a.d - 1) * a.H + 8 * chunk + 4);
        const bool in = i < a.d;
        w[i][0] = in ? lo.x : 0.f; w[i][1] = in ? lo.y : 0.f; w[i][2] = in ? lo.z : 0.f; w[i][3] = in ? lo.w : 0.f;
        w[i][4] = in ? hi.x : 0.f; w[i][5] = in ? hi.y : 0.f; w[i][6] = in ? hi.z : 0.f; w[i][7] = in ? hi.w : 0.f;
    }
    {
        const float4 lo = *reinterpret_cast<const float4*>(a.b + 8 * chunk), hi = *reinterpret_cast<const float4*>(a.b + 8 * chunk + 4);
        b8[0] = lo.x; b8[1] = lo.y; b8[2] = lo.z; b8[3] = lo.w; b8[4] = hi.x; b8[5] = hi.y; b8[6] = hi.z; b8[7] = hi.w;
    }
    const float floor_v = a.relu ? 0.f : -__builtin_huge_valf();
    constexpr int U = 4;                                  // rows per trip: all x loads first
    const long long stride = (long long)gridDim.x * rpp;
    for (long long row = (long long)blockIdx.x * rpp + rsub; row < a.rows; row += U * stride) {
        float xi[U][DMAX];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long rr = min(row + u * stride, (long long)a.rows - 1);
#pragma unroll
            for (int i = 0; i < DMAX; ++i) xi[u][i] = a.x[rr * a.d + min(i, a.d - 1)];   // rows i >= d meet zero weights
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long rr = row + u * stride;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = b8[j];
#pragma unroll
            for (int i = 0; i < DMAX; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaf(xi[u][i], w[i][j], o[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], floor_v);
            if (rr < a.rows)
                *reinterpret_cast<uint4*>(a.y + rr * a.H + 8 * chunk) =
                    make_uint4(sk_pack(o[0], o[1]), sk_pack(o[2], o[3]), sk_pack(o[4], o[5]), sk_pack(o[6], o[7]));
        }
    }
}

// ---- rows x small matrix on the bf16 matrix cores ---------------------------------------------------------------------------
enum { SK_PLAIN = 0, SK_REPARAM = 1, SK_ELBO = 2 };
struct SkRowsArgs {
    const __bf16* A; const __bf16* Bt; int M, K, d, lda;      // A [M, K]; Bt [32, K] (rows >= d are zero)
    float* C; int ldc; const float* bias; int accumulate;
    float* C2; const float* z1; const float* lv;              // REPARAM: C = mu, C2 = samples
    const float* x; const float* z2; const float* eps_param; float eps_cli, inv_bt; float* part;   // ELBO: C = dL/dx_hat
};

template <int EPI>
__global__ __launch_bounds__(256) void sk_rows_mfma_kernel(const SkRowsArgs a) {
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const long long m = (long long)blockIdx.x * 128 + wave * 32 + r;
    const bool row_ok = m < a.M;
    const __bf16* ap = a.A + (row_ok ? m : (long long)a.M - 1) * a.lda + 8 * h;
    const __bf16* bp = a.Bt + (long long)r * a.K + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += 64) {           // K is a multiple of 64: four 16-deep steps, eight loads in flight
        bf16x8 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = *reinterpret_cast<const bf16x8*>(ap + k0 + 16 * u);
            bv[u] = *reinterpret_cast<const bf16x8*>(bp + k0 + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)                  // swapped operands: D[n][m] -- lane & 31 = row m, registers = columns n
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[u], av[u], acc, 0, 0, 0);
    }
    float e_mse = 0.f, e_deps = 0.f, e_inv_var = 0.f, e_sigma = 0.f, e_dscale = 0.f;
    if (EPI == SK_ELBO) {
        const float eps = a.eps_param ? a.eps_param[0] * a.eps_cli : a.eps_cli;
        e_inv_var = expf(-eps); e_sigma = expf(0.5f * eps); e_dscale = e_inv_var * a.inv_bt;
    }
    // register q = column (q & 3) + 8 (q >> 2) + 4 h of row m
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int n = (q & 3) + 8 * (q >> 2) + 4 * h;
        if (n >= a.d || !row_ok) continue;
        const long long o = m * a.ldc + n;
        float v = acc[q] + (a.bias ? a.bias[n] : 0.f);
        if (EPI == SK_PLAIN) {
            if (a.accumulate) v += a.C[o];
            a.C[o] = v;
        } else if (EPI == SK_REPARAM) {
            a.C[o] = v;
            a.C2[o] = v + expf(0.5f * a.lv[n]) * a.z1[o];
        } else {
            const float z = a.z2[o];
            const float rr = v + e_sigma * z - a.x[o];               // x_hat - x, x_hat = y + z2 e^{eps/2}
            const float qq = rr * rr * e_inv_var;
            e_mse += 0.5f * qq;
            e_deps += -0.5f * qq + 0.5f * e_sigma * z * rr * e_inv_var;
            a.C[o] = rr * e_dscale;
        }
    }
    if (EPI == SK_ELBO) {      // this workgroup's two sums, fixed order: lanes by xor-shuffle, then the four waves in order
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) { e_mse += __shfl_xor(e_mse, s, 64); e_deps += __shfl_xor(e_deps, s, 64); }
        if (lane == 0) { red[2 * wave] = e_mse; red[2 * wave + 1] = e_deps; }
        __syncthreads();
        if (threadIdx.x == 0) {
            a.part[2 * (long long)blockIdx.x] = ((red[0] + red[2]) + red[4]) + red[6];
            a.part[2 * (long long)blockIdx.x + 1] = ((red[1] + red[3]) + red[5]) + red[7];
        }
    }
}

// ---- last layer backward: dX and dW|db from one pass over h ----------------------------------------------------------------
struct SkBwdArgs {
    const __bf16* big;      // last: h [rows, H] (the layer's bf16 input); first: dY [rows, H]
    const float* small;     // last: dy [rows, d]; first: x [rows, d]
    const float* w;         // last: W [H, d]
    __bf16* dh;             // last: [rows, H]
    float* partial;         // [gridDim.x][(H + 1) d] resp. [(d + 1) H]
    int rows, d, H, rows_per_wg;
};

template <int DMAX>
__global__ __launch_bounds__(256) void sk_last_bwd_kernel(const SkBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];
    const int cprw = a.H >> 3, rpp = 256 / cprw, t = threadIdx.x;
    const bool active = t < rpp * cprw;
    const int chunk = active ? t % cprw : 0, rsub = active ? t / cprw : 0;
    // pairs of latent / data columns as float2: v_pk_fma_f32 does two of this kernel's ~130 FMAs per row and lane at once
    // (it is VALU-, not HBM-bound: 216 vector instructions per KB of h)
    constexpr int DP = DMAX / 2;
    f32x2 w[8][DP], g[8][DP], gb[DP];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int n = 0; n < DMAX; ++n) {
            const float v = a.w[(long long)(8 * chunk + j) * a.d + min(n, a.d - 1)];
            w[j][n >> 1][n & 1] = n < a.d ? v : 0.f;
            g[j][n >> 1][n & 1] = 0.f;
        }
#pragma unroll
    for (int n = 0; n < DP; ++n) gb[n] = f32x2{0.f, 0.f};
    const long long r0 = (long long)blockIdx.x * a.rows_per_wg, r1 = min((long long)a.rows, r0 + a.rows_per_wg);
    // U rows per trip, every load of the trip issued before the first use (unconditional, clamped row; the surplus rows of the
    // last trip are masked out of the sums and not stored): one row per trip was one exposed HBM round trip per KB and wave
    constexpr int U = 4;
    if (active)
        for (long long row = r0 + rsub; row < r1; row += (long long)U * rpp) {
            uint4 hv[U]; f32x2 dyv[U][DP];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long rr = min(row + (long long)u * rpp, r1 - 1);
                hv[u] = *reinterpret_cast<const uint4*>(a.big + rr * a.H + 8 * chunk);
#pragma unroll
                for (int n = 0; n < DMAX; ++n) dyv[u][n >> 1][n & 1] = a.small[rr * a.d + min(n, a.d - 1)];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long rr = row + (long long)u * rpp;
                const bool in = rr < r1;
                float hf[8], o[8];
                sk_unpack8(hv[u], hf);
#pragma unroll
                for (int n = 0; n < DMAX; ++n) dyv[u][n >> 1][n & 1] = (in && n < a.d) ? dyv[u][n >> 1][n & 1] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x2 s2 = {0.f, 0.f};
                    const f32x2 hj = {hf[j], hf[j]};
#pragma unroll
                    for (int n = 0; n < DP; ++n) { s2 = dyv[u][n] * w[j][n] + s2; g[j][n] = hj * dyv[u][n] + g[j][n]; }
                    o[j] = hf[j] > 0.f ? s2[0] + s2[1] : 0.f;                  // relu'
                }
#pragma unroll
                for (int n = 0; n < DP; ++n) gb[n] += dyv[u][n];
                if (in)
                    *reinterpret_cast<uint4*>(a.dh + rr * a.H + 8 * chunk) =
                        make_uint4(sk_pack(o[0], o[1]), sk_pack(o[2], o[3]), sk_pack(o[4], o[5]), sk_pack(o[6], o[7]));
            }
        }
    // the rpp row phases of a column chunk: phases 1.. park their sums in LDS, phase 0 adds them in order
    const int per = 8 * a.d + a.d;                          // floats a thread contributes: its 8 x d block, then gb
    if (active && rsub > 0) {
        float* p = sk_lds + ((long long)(rsub - 1) * cprw + chunk) * per;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int n = 0; n < DMAX; ++n) if (n < a.d) p[j * a.d + n] = g[j][n >> 1][n & 1];
#pragma unroll
        for (int n = 0; n < DMAX; ++n) if (n < a.d) p[8 * a.d + n] = gb[n >> 1][n & 1];
    }
    __syncthreads();
    if (active && rsub == 0) {
        for (int s = 1; s < rpp; ++s) {
            const float* p = sk_lds + ((long long)(s - 1) * cprw + chunk) * per;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int n = 0; n < DMAX; ++n) if (n < a.d) g[j][n >> 1][n & 1] += p[j * a.d + n];
#pragma unroll
            for (int n = 0; n < DMAX; ++n) if (n < a.d) gb[n >> 1][n & 1] += p[8 * a.d + n];
        }
        float* out = a.partial + (long long)blockIdx.x * ((long long)(a.H + 1) * a.d);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int n = 0; n < DMAX; ++n) if (n < a.d) out[(long long)(8 * chunk + j) * a.d + n] = g[j][n >> 1][n & 1];
        if (chunk == 0) {
#pragma unroll
            for (int n = 0; n < DMAX; ++n) if (n < a.d) out[(long long)a.H * a.d + n] = gb[n >> 1][n & 1];
        }
    }
}

// ---- last layer backward on the f32 matrix cores (d <= 8, H = 64 NT = 512: a row of h is one 1 KB piece) ------------------------------------------------------------
// The per-lane loops above are vector-ALU-bound (216 vector instructions per KB of h: 54 us for C3's 64 MB + 64 MB).  Here a
// workgroup walks its rows in blocks of 16 through a ring of FOUR LDS slots filled by LDS-DMA three blocks ahead (a row of h is
// one 1 KB global_load_lds, row stride 1056 bytes; the block's dy rows are contiguous in memory: one more piece) with counted
// waits -- a workgroup has only 8 blocks, so nothing else would hide the memory round trips -- and both products run on
// v_mfma_f32_16x16x4_f32: exact float32 products of the bf16 h and the float32 dy, as before:
//   dh^T[hcol][row] = W[hcol][:] . dy[row][:]      M = 16 columns of H, N = the 16 rows, K = d in two k-steps.  The M index of a
//       PAIR of tiles is dealt so that a lane ends up with 8 consecutive columns of one row (M index 4 g + r of the pair's first
//       tile = column 8 g + r, of its second = 8 g + 4 + r): one 16-byte LDS read of h for the relu mask, one 16-byte store.
//   G[hcol][n] += sum_rows h[row][hcol] dy[row][n]  M = 16 columns of H, N = n (d of 16 used), K = the 16 rows in four k-steps;
//       accumulators live across the workgroup's rows.  A from LDS: one ds_read_u16 per product.
// Wave w owns columns 16 NT w .. of H.  Rows past the workgroup's range: dy reads as zero, nothing is stored.
// rows_per_wg a multiple of 16 (the host checks): every piece starts on a 16-byte boundary.
// FIRST: the first layer's [x | 1]^T dY instead -- the same G product with big = dY, small = x and a ones column behind x's d
// (its row of the image is the bias gradient); no dh.
// One 1 KB LDS-DMA piece in inline assembly: through the builtin hipcc knows that LDS is being written and puts s_waitcnt vmcnt(0)
// in front of every later LDS read it cannot tell apart -- the ring below was drained four times per block (ISA; 36 us per launch).
// The waits are this kernel's own (counted) ones.
__device__ __forceinline__ void sk_glds16(const void* gsrc, void* lds_wave_base) {
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_void_t*)lds_wave_base);
    _Pragma("clang diagnostic push") _Pragma("clang diagnostic ignored \"-Winline-asm\"")      // (M0 on the clobber list: it is what the LDS-DMA takes its LDS address from)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "memory", "m0");
    _Pragma("clang diagnostic pop")
}
template <int NT, bool FIRST = false>
__global__ __launch_bounds__(256) void sk_last_bwd_mfma_kernel(const SkBwdArgs a) {
    extern __shared__ __attribute__((aligned(1024))) float sk_lds[];
    static_assert(NT == 8, "a row of h = one LDS-DMA piece");
    constexpr int H = 64 * NT, RS = 2 * H + 32, SLOT = 16 * RS + 1024, NSL = 4, RPW = 16 * (2 * H) / 1024 / 4;   // h pieces per wave and block
    char* const ring = reinterpret_cast<char*>(sk_lds);                     // [NSL][16 rows of h | the block's dy: a whole 1 KB piece lands there]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), g = lane >> 4, c = lane & 15, d = a.d;
    const long long r0 = (long long)blockIdx.x * a.rows_per_wg, r1 = min((long long)a.rows, r0 + a.rows_per_wg);
    const int nblk = r1 > r0 ? (int)((r1 - r0 + 15) / 16) : 0;
    const int col0 = 16 * NT * wave;
    // n in {0, 1, 2} blocks of this wave's pieces may stay in flight (what it stored since counts too, but a last block may store
    // less than a whole one: the stores are left out of the allowance -- a wait for some of them is the price)
    auto wait_vm = [&](int n) {
        if (wave == 0) { if (n >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RPW + 1))); else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RPW + 1)); else asm volatile("s_waitcnt vmcnt(0)"); }
        else { if (n >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RPW)); else if (n == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RPW)); else asm volatile("s_waitcnt vmcnt(0)"); }
    };
    auto issue_block = [&](int blk) {                          // rows of h: piece q of the block = row q's (H / 512) th KB ...
        char* slot = ring + (blk & (NSL - 1)) * SLOT;
        const long long rb = r0 + 16ll * blk;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int piece = wave * RPW + i, row = piece / (2 * H / 1024), part = piece % (2 * H / 1024);
            const long long rr = min(rb + row, r1 - 1);
            sk_glds16(reinterpret_cast<const char*>(a.big) + rr * (2 * H) + part * 1024 + lane * 16, slot + row * RS + part * 1024);
        }
        if (wave == 0) {                                       // ... and the 16 d floats of dy behind them (clamped inside the tensor)
            const long long off = min(rb * d * 4 + lane * 16, (long long)a.rows * d * 4 - 16);
            sk_glds16(reinterpret_cast<const char*>(a.small) + off, slot + 16 * RS);
        }
    };
    // A of the dh product: this wave's W rows in the paired order, k = n
    [[maybe_unused]] float wa[NT][2];
    if constexpr (!FIRST) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
            const int hcol = col0 + 32 * (tl >> 1) + 8 * (c >> 2) + 4 * (tl & 1) + (c & 3);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int n = g + 4 * s2;
                const float v = a.w[(long long)hcol * d + min(n, d - 1)];
                wa[tl][s2] = n < d ? v : 0.f;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the counted waits below count LDS-DMA pieces only)
    f32x4 G[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) G[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gb = 0.f;
#pragma unroll
    for (int q = 0; q < NSL - 1; ++q) if (q < nblk) issue_block(q);
    for (int blk = 0; blk < nblk; ++blk) {
        const long long rb = r0 + 16ll * blk;
        wait_vm(min(NSL - 2, nblk - 1 - blk));                 // this wave's pieces of block blk have landed ...
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... everybody's; and everybody has left slot (blk - 1) & 3
        if (blk + NSL - 1 < nblk) issue_block(blk + NSL - 1);
        const char* hblk = ring + (blk & (NSL - 1)) * SLOT;
        const float* dyl = reinterpret_cast<const float*>(hblk + 16 * RS);          // [16][d]
        [[maybe_unused]] float dyT[2];
        float dyB[4];
        {
            float rawT[2] = {0.f, 0.f}, rawB[4];               // all six LDS reads first, the selects behind them
            if constexpr (!FIRST) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) rawT[s2] = dyl[c * d + min(g + 4 * s2, d - 1)];
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) rawB[s4] = dyl[(4 * s4 + g) * d + min(c, d - 1)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) dyT[s2] = (rb + c < r1 && g + 4 * s2 < d) ? rawT[s2] : 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                dyB[s4] = rb + 4 * s4 + g < r1 ? (c < d ? rawB[s4] : (FIRST && c == d) ? 1.f : 0.f) : 0.f;
                if constexpr (!FIRST) gb += dyB[s4];
            }
        }
        // ---- G += h^T dy
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
            unsigned short hv[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) hv[s4] = *reinterpret_cast<const unsigned short*>(hblk + (4 * s4 + g) * RS + 2 * (col0 + 16 * tl + c));
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
                G[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, (unsigned)hv[s4] << 16), dyB[s4], G[tl], 0, 0, 0);
        }
        // ---- dh = (W dy^T) * (h > 0), a pair of tiles = 8 consecutive columns of row c per lane
        if constexpr (!FIRST)
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
            f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2 * pr][s2], dyT[s2], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2 * pr + 1][s2], dyT[s2], o1, 0, 0, 0);
            }
            const int hc = col0 + 32 * pr + 8 * g;
            float hf[8];
            sk_unpack8(*reinterpret_cast<const uint4*>(hblk + c * RS + 2 * hc), hf);
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = hf[r] > 0.f ? o0[r] : 0.f; o[4 + r] = hf[4 + r] > 0.f ? o1[r] : 0.f; }
            if (rb + c < r1) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 pk = {sk_pack(o[0], o[1]), sk_pack(o[2], o[3]), sk_pack(o[4], o[5]), sk_pack(o[6], o[7])};
                // (a store the compiler does not count: its own waits must not drain the LDS-DMA pieces in flight)
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(a.dh + (rb + c) * H + hc), "v"(pk) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- the workgroup's partial image: G[hcol][n] (lane: n = c, rows 4 g + r of each tile), then the bias row
    if constexpr (FIRST) {                                     // [(d + 1)][H]: row n = c, four consecutive columns per tile and lane
        float* out = a.partial + (long long)blockIdx.x * ((long long)(d + 1) * H);
        if (c <= d) {
#pragma unroll
            for (int tl = 0; tl < NT; ++tl)
                *reinterpret_cast<float4*>(out + (long long)c * H + col0 + 16 * tl + 4 * g) = make_float4(G[tl][0], G[tl][1], G[tl][2], G[tl][3]);
        }
        return;
    }
    float* out = a.partial + (long long)blockIdx.x * ((long long)(H + 1) * d);
    if (c < d) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long long)(col0 + 16 * tl + 4 * g + r) * d + c] = G[tl][r];
    }
    gb += __shfl_xor(gb, 16, 64);
    gb += __shfl_xor(gb, 32, 64);
    if (wave == 0 && lane < d) out[(long long)H * d + lane] = gb;
}

// ---- first layer dW|db ---------------------------------------------------------------------------------------------------
template <int DMAX>
__global__ __launch_bounds__(256) void sk_first_bwd_kernel(const SkBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];
    const int cprw = a.H >> 3, rpp = 256 / cprw, t = threadIdx.x;
    const bool active = t < rpp * cprw;
    const int chunk = active ? t % cprw : 0, rsub = active ? t / cprw : 0;
    f32x2 g[DMAX + 1][4];               // column pairs as float2 (v_pk_fma_f32)
#pragma unroll
    for (int i = 0; i <= DMAX; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) g[i][j] = f32x2{0.f, 0.f};
    const long long r0 = (long long)blockIdx.x * a.rows_per_wg, r1 = min((long long)a.rows, r0 + a.rows_per_wg);
    constexpr int U = 4;                                  // rows per trip, loads first (see sk_last_bwd_kernel)
    if (active)
        for (long long row = r0 + rsub; row < r1; row += (long long)U * rpp) {
            uint4 dv[U]; float xi[U][DMAX];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long rr = min(row + (long long)u * rpp, r1 - 1);
                dv[u] = *reinterpret_cast<const uint4*>(a.big + rr * a.H + 8 * chunk);
#pragma unroll
                for (int i = 0; i < DMAX; ++i) xi[u][i] = a.small[rr * a.d + min(i, a.d - 1)];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = row + (long long)u * rpp < r1;
                float df[8];
                sk_unpack8(dv[u], df);
#pragma unroll
                for (int j = 0; j < 8; ++j) df[j] = in ? df[j] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 d2 = {df[2 * j], df[2 * j + 1]};
#pragma unroll
                    for (int i = 0; i < DMAX; ++i) { const float xv = i < a.d ? xi[u][i] : 0.f; g[i][j] = f32x2{xv, xv} * d2 + g[i][j]; }
                    g[DMAX][j] += d2;                                         // the ones row of [x | 1]^T dY
                }
            }
        }
    const int per = (a.d + 1) * 8;
    if (active && rsub > 0) {
        float* p = sk_lds + ((long long)(rsub - 1) * cprw + chunk) * per;
#pragma unroll
        for (int i = 0; i < DMAX; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) if (i < a.d) p[i * 8 + j] = g[i][j >> 1][j & 1];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[a.d * 8 + j] = g[DMAX][j >> 1][j & 1];
    }
    __syncthreads();
    if (active && rsub == 0) {
        for (int s = 1; s < rpp; ++s) {
            const float* p = sk_lds + ((long long)(s - 1) * cprw + chunk) * per;
#pragma unroll
            for (int i = 0; i < DMAX; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) if (i < a.d) g[i][j >> 1][j & 1] += p[i * 8 + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) g[DMAX][j >> 1][j & 1] += p[a.d * 8 + j];
        }
        float* out = a.partial + (long long)blockIdx.x * ((long long)(a.d + 1) * a.H);
#pragma unroll
        for (int i = 0; i < DMAX; ++i)
            if (i < a.d) {
                *reinterpret_cast<float4*>(out + (long long)i * a.H + 8 * chunk) = make_float4(g[i][0][0], g[i][0][1], g[i][1][0], g[i][1][1]);
                *reinterpret_cast<float4*>(out + (long long)i * a.H + 8 * chunk + 4) = make_float4(g[i][2][0], g[i][2][1], g[i][3][0], g[i][3][1]);
            }
        *reinterpret_cast<float4*>(out + (long long)a.d * a.H + 8 * chunk) = make_float4(g[DMAX][0][0], g[DMAX][0][1], g[DMAX][1][0], g[DMAX][1][1]);
        *reinterpret_cast<float4*>(out + (long long)a.d * a.H + 8 * chunk + 4) = make_float4(g[DMAX][2][0], g[DMAX][2][1], g[DMAX][3][0], g[DMAX][3][1]);
    }
}

// slab[s][i] = sum of partial[s * G + w][i], w = 0 .. G - 1 in order, 32 loads in flight (8 until round 3: a group of 64 images
// was eight dependent round trips, 5 us for a kernel that moves 6 MB)
__global__ __launch_bounds__(256) void sk_partials_reduce_kernel(const float* partial, long long n, int G, float* slab0, long long slab_stride) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = partial + (long long)blockIdx.y * G * n + i;
    float s = 0.f;
    constexpr int UF = 32;
    for (int w0 = 0; w0 < G; w0 += UF) {
        float tv[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) tv[u] = p[(long long)min(w0 + u, G - 1) * n];
#pragma unroll
        for (int u = 0; u < UF; ++u) s += w0 + u < G ? tv[u] : 0.f;
    }
    slab0[(long long)blockIdx.y * slab_stride + i] = s;
}

// bf16, 32-row padded, k-contiguous copies of the skinny kernels for sk_rows_mfma: transposed = 1: src W [H, d] (last layer
// forward), out[n][k] = W[k][n]; transposed = 0: src W [d, H] (first layer dX), out[n][k] = W[n][k]
struct SkPrepArgs { int n; int H[16], d[16], transposed[16]; long long w_off[16], out_off[16]; };
__global__ __launch_bounds__(256) void sk_prep_kernel(const float* __restrict__ params, __bf16* __restrict__ out, const SkPrepArgs a) {
    const int l = blockIdx.y, H = a.H[l], d = a.d[l];
    const float* w = params + a.w_off[l];
    __bf16* o = out + a.out_off[l];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 32 * H; e += gridDim.x * 256) {
        const int n = e / H, k = e % H;
        const float v = n < d ? (a.transposed[l] ? w[(long long)k * d + n] : w[(long long)n * H + k]) : 0.f;
        o[e] = (__bf16)v;
    }
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------
bool sk_supported(int d, int H) { return d >= 1 && d <= 16 && H % 64 == 0 && H >= 64 && H <= 2048; }

int launch_sk_first_fwd(const float* x, const float* w, const float* b, __bf16* y, int rows, int d, int H, bool relu, hipStream_t st) {
    SkFwdArgs a{x, w, b, y, rows, d, H, relu ? 1 : 0};
    const int rpp = 256 / (H / 8);
    const unsigned grid = (unsigned)std::min<long long>(((long long)rows + rpp - 1) / rpp, 2048);
    ProfScope ps("sk16_first_fwd", st);
    if (d <= 8) launch_k(ps, sk_first_fwd_kernel<8>, dim3(grid), dim3(256), 0, st, a);
    else launch_k(ps, sk_first_fwd_kernel<16>, dim3(grid), dim3(256), 0, st, a);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

static int sk_rows(const SkRowsArgs& a, int epi, const char* label, hipStream_t st) {
    const unsigned grid = (unsigned)((a.M + 127) / 128);
    ProfScope ps(label, st);
    if (epi == SK_PLAIN) launch_k(ps, sk_rows_mfma_kernel<SK_PLAIN>, dim3(grid), dim3(256), 0, st, a);
    else if (epi == SK_REPARAM) launch_k(ps, sk_rows_mfma_kernel<SK_REPARAM>, dim3(grid), dim3(256), 0, st, a);
    else launch_k(ps, sk_rows_mfma_kernel<SK_ELBO>, dim3(grid), dim3(256), 0, st, a);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}
int launch_sk_last_fwd(const __bf16* h, const __bf16* wp, const float* b, float* y, int rows, int H, int d, hipStream_t st) {
    SkRowsArgs a{};
    a.A = h; a.Bt = wp; a.M = rows; a.K = H; a.d = d; a.lda = H; a.C = y; a.ldc = d; a.bias = b;
    return sk_rows(a, SK_PLAIN, "sk16_last_fwd", st);
}
int launch_sk_last_fwd_reparam(const __bf16* h, const __bf16* wp, const float* b, float* mu, float* samples, const float* z1,
                               const float* lv, int rows, int H, int d, hipStream_t st) {
    SkRowsArgs a{};
    a.A = h; a.Bt = wp; a.M = rows; a.K = H; a.d = d; a.lda = H; a.C = mu; a.ldc = d; a.bias = b; a.C2 = samples; a.z1 = z1; a.lv = lv;
    return sk_rows(a, SK_REPARAM, "sk16_last_fwd_reparam", st);
}
int launch_sk_last_fwd_elbo(const __bf16* h, const __bf16* wp, const float* b, float* d_out, const float* x, const float* z2,
                            const float* eps_param, float eps_cli, float inv_bt, float* part, int rows, int H, int d, int* bm,
                            int* nbx, hipStream_t st) {
    SkRowsArgs a{};
    a.A = h; a.Bt = wp; a.M = rows; a.K = H; a.d = d; a.lda = H; a.C = d_out; a.ldc = d; a.bias = b;
    a.x = x; a.z2 = z2; a.eps_param = eps_param; a.eps_cli = eps_cli; a.inv_bt = inv_bt; a.part = part;
    *bm = 128; *nbx = 1;
    return sk_rows(a, SK_ELBO, "sk16_last_fwd_elbo", st);
}
// first layer dX (the decoder's first layer): dx[rows, d] (+)= dY[rows, H] . W^T, W [d, H] padded to wp [32, H]
int launch_sk_first_dx(const __bf16* dy, const __bf16* wp, float* dx, int rows, int H, int d, bool accumulate, hipStream_t st) {
    SkRowsArgs a{};
    a.A = dy; a.Bt = wp; a.M = rows; a.K = H; a.d = d; a.lda = H; a.C = dx; a.ldc = d; a.accumulate = accumulate ? 1 : 0;
    return sk_rows(a, SK_PLAIN, "sk16_first_dx", st);
}

constexpr int kSkGroup = 64;       // partial images summed into one slab

// S slabs <- S * 64 workgroups; rows_per_wg = ceil(rows / (S * 64))
static int sk_reduce(const float* partial, long long n, int S, float* slab0, int64_t slab_stride, hipStream_t st) {
    ProfScope ps("sk16_partials_reduce", st);
    launch_k(ps, sk_partials_reduce_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)S), dim3(256), 0, st, partial, n, kSkGroup,
             slab0, (long long)slab_stride);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}
size_t sk_partial_bytes(int d, int H, int S) { return (size_t)S * kSkGroup * (size_t)std::max((H + 1) * d, (d + 1) * H) * sizeof(float); }

int launch_sk_last_bwd(const __bf16* h, const float* dy, const float* w, __bf16* dh, float* partial, float* slab0, int64_t slab_stride,
                       int S, int rows, int H, int d, hipStream_t st) {
    const int nwg = S * kSkGroup;
    SkBwdArgs a{h, dy, w, dh, partial, rows, d, H, (rows + nwg - 1) / nwg};
    const int cprw = H / 8, rpp = 256 / cprw;
    const size_t lds = (size_t)std::max(rpp - 1, 0) * cprw * (9 * d) * sizeof(float);
    if (d <= 8 && H == 512 && a.rows_per_wg % 16 == 0 && (long long)rows * d * 4 >= 16) {
        // the matrix-core form: rows in blocks of 16 through a ring of four LDS images of h (+ dy)
        ProfScope ps("sk16_last_bwd", st);
        const size_t lds_m = (size_t)4 * (16 * (2 * H + 32) + 1024);
        static thread_local PerDeviceOnce setm;
        if (setm.need()) { VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)sk_last_bwd_mfma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); setm.mark(); }
        launch_k(ps, sk_last_bwd_mfma_kernel<8>, dim3(nwg), dim3(256), lds_m, st, a);
        VAEK_HIP_CHECK(hipGetLastError());
    } else {
        ProfScope ps("sk16_last_bwd", st);
        if (d <= 8) {
            static thread_local PerDeviceOnce set8;
            if (set8.need()) { VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)sk_last_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set8.mark(); }
            launch_k(ps, sk_last_bwd_kernel<8>, dim3(nwg), dim3(256), lds, st, a);
        } else {
            static thread_local PerDeviceOnce set16;
            if (set16.need()) { VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)sk_last_bwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set16.mark(); }
            launch_k(ps, sk_last_bwd_kernel<16>, dim3(nwg), dim3(256), lds, st, a);
        }
        VAEK_HIP_CHECK(hipGetLastError());
    }
    return sk_reduce(partial, (long long)(H + 1) * d, S, slab0, slab_stride, st);
}

int launch_sk_first_bwd(const float* x, const __bf16* dy, float* partial, float* slab0, int64_t slab_stride, int S, int rows, int H,
                        int d, hipStream_t st) {
    const int nwg = S * kSkGroup;
    SkBwdArgs a{dy, x, nullptr, nullptr, partial, rows, d, H, (rows + nwg - 1) / nwg};
    const int cprw = H / 8, rpp = 256 / cprw;
    const size_t lds = (size_t)std::max(rpp - 1, 0) * cprw * ((d + 1) * 8) * sizeof(float);
    if (d <= 8 && H == 512 && a.rows_per_wg % 16 == 0 && (long long)rows * d * 4 >= 16) {
        // the matrix-core form (sk_last_bwd_mfma_kernel's G product with a ones column)
        ProfScope ps("sk16_first_bwd", st);
        const size_t lds_m = (size_t)4 * (16 * (2 * H + 32) + 1024);
        static thread_local PerDeviceOnce setm;
        if (setm.need()) { VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)sk_last_bwd_mfma_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); setm.mark(); }
        launch_k(ps, sk_last_bwd_mfma_kernel<8, true>, dim3(nwg), dim3(256), lds_m, st, a);
        VAEK_HIP_CHECK(hipGetLastError());
    } else {
        ProfScope ps("sk16_first_bwd", st);
        if (d <= 8) {
            static thread_local PerDeviceOnce set8;
            if (set8.need()) { VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)sk_first_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set8.mark(); }
            launch_k(ps, sk_first_bwd_kernel<8>, dim3(nwg), dim3(256), lds, st, a);
        } else {
            static thread_local PerDeviceOnce set16;
            if (set16.need()) { VAEK_HIP_CHECK(hipFuncSetAttribute((const void*)sk_first_bwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set16.mark(); }
            launch_k(ps, sk_first_bwd_kernel<16>, dim3(nwg), dim3(256), lds, st, a);
        }
        VAEK_HIP_CHECK(hipGetLastError());
    }
    return sk_reduce(partial, (long long)(d + 1) * H, S, slab0, slab_stride, st);
}

int launch_sk_prep(const float* params, __bf16* out, const int* H, const int* d, const int* transposed, const int64_t* w_off,
                   const int64_t* out_off, int n, hipStream_t st) {
    if (n <= 0) return VAEK_OK;
    if (n > 16) { set_error("too many skinny layers"); return VAEK_ERR_INVALID; }
    SkPrepArgs a{};
    a.n = n;
    for (int i = 0; i < n; ++i) { a.H[i] = H[i]; a.d[i] = d[i]; a.transposed[i] = transposed[i]; a.w_off[i] = w_off[i]; a.out_off[i] = out_off[i]; }
    ProfScope ps("sk16_prep", st);
    launch_k(ps, sk_prep_kernel, dim3(16, (unsigned)n), dim3(256), 0, st, params, out, a);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

}  // namespace vaek
