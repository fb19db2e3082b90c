// Convolution layers of the convolutional VAE (BASELINE config 5; NO reference counterpart -- the architecture is this
// repository's own specification, DESIGN.md 3.4, and oracle/conv_vae_oracle.py is what these kernels are checked against):
// 4x4 / stride 2 / pad 1 convolutions on NHWC float32 tensors as IMPLICIT GEMMs on v_mfma_f32_32x32x16_bf16 (f32 accumulate,
// operands rounded to bf16 while they are staged into LDS -- the arithmetic of gemm_bf16.hip, whose tile machinery this file
// shares: 128 x 128 output tile, 4 waves x 2 x 2 MFMA tiles, 32-deep k-tiles, [row][k] LDS images at an 80-byte row stride,
// next k-tile fetched into registers under the MFMAs).
//
//   forward   y[n, i, j, o] = act(b[o] + sum_{kh, kw, c} x[n, 2 i + kh - 1, 2 j + kw - 1, c] K[kh, kw, c, o])
//             GEMM rows = output pixels (n, i, j), columns = o, inner index k = (kh, kw, c): the A operand is GATHERED from the
//             input image (zeros outside it), never materialised; the HWIO kernel array IS the row-major [k][o] B operand;
//             the row-major [pixel][o] result IS the NHWC output.
//
// Built so far: the forward convolution (the four encoder layers of config 5) and the transposed convolution = the input gradient
// (four parity-phase GEMMs with 2 x 2 taps each) and the kernel gradient (batch-split GEMM with the gather transposed): every
// product of both layer kinds' forward and backward passes (oracle: conv_fwd / conv_bwd / conv_t_fwd / conv_t_bwd).  The
// convolutional VAE's train step (bottleneck Dense layers + ELBO around them) is not assembled yet.
#include "vaek_internal.h"

namespace vaek {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

constexpr int CBM = 128, CBN = 128, CNT = 256, CBK = 32, CSTR = CBK + 8, CKU = CBK / 8;     // (64-deep k-tiles measured slower: forward 1.1 -> 2.0 ms, kernel gradient unchanged)

struct ConvArgs {
    const float* x; const float* w; const float* bias; const float* mask; float* y;
    int B, H, W, Cin, Cout, Ho, Wo;
    int M, N, K;                        // M = B Ho Wo, N = Cout, K = 16 Cin
    int relu;
};

// The shared main loop: registers -> LDS (A: thread = row t / 8 + 32 u, 4 consecutive k; B either the same form, B_KCONT, or
// thread = column t & 127, 16 consecutive k), barrier, next k-tile's fetch into registers, 2 x 2 MFMA tiles per wave.
template <bool B_KCONT, bool A_KCONT = true, typename FA, typename FB>
__device__ __forceinline__ void conv_mainloop(__bf16* As, __bf16* Bs, int K, float (&ra)[CKU][4], float (&rb)[CKU][4], FA fetch_a,
                                              FB fetch_b, f32x16 (&acc)[2][2], int kbeg = 0) {
    constexpr int TPR = CBK / 4, RPP = CNT / TPR;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    if (kbeg < K) { fetch_a(kbeg); fetch_b(kbeg); }
    for (int k0 = kbeg; k0 < K; k0 += CBK) {
        __syncthreads();
        if (A_KCONT) {
#pragma unroll
            for (int u = 0; u < CKU; ++u) {
                const bf16x4 h = {(__bf16)ra[u][0], (__bf16)ra[u][1], (__bf16)ra[u][2], (__bf16)ra[u][3]};
                *reinterpret_cast<bf16x4*>(&As[(t / TPR + RPP * u) * CSTR + (t % TPR) * 4]) = h;
            }
        } else {
            __bf16* rowp = As + (t & 127) * CSTR + (CBK / 2) * (t >> 7);
#pragma unroll
            for (int h = 0; h < CKU / 2; ++h) {
                const bf16x8 w8 = {(__bf16)ra[2 * h][0], (__bf16)ra[2 * h][1], (__bf16)ra[2 * h][2], (__bf16)ra[2 * h][3],
                                   (__bf16)ra[2 * h + 1][0], (__bf16)ra[2 * h + 1][1], (__bf16)ra[2 * h + 1][2], (__bf16)ra[2 * h + 1][3]};
                *reinterpret_cast<bf16x8*>(rowp + 8 * h) = w8;
            }
        }
        if (B_KCONT) {
#pragma unroll
            for (int u = 0; u < CKU; ++u) {
                const bf16x4 h = {(__bf16)rb[u][0], (__bf16)rb[u][1], (__bf16)rb[u][2], (__bf16)rb[u][3]};
                *reinterpret_cast<bf16x4*>(&Bs[(t / TPR + RPP * u) * CSTR + (t % TPR) * 4]) = h;
            }
        } else {
            __bf16* rowp = Bs + (t & 127) * CSTR + (CBK / 2) * (t >> 7);
#pragma unroll
            for (int h = 0; h < CKU / 2; ++h) {
                const bf16x8 w8 = {(__bf16)rb[2 * h][0], (__bf16)rb[2 * h][1], (__bf16)rb[2 * h][2], (__bf16)rb[2 * h][3],
                                   (__bf16)rb[2 * h + 1][0], (__bf16)rb[2 * h + 1][1], (__bf16)rb[2 * h + 1][2], (__bf16)rb[2 * h + 1][3]};
                *reinterpret_cast<bf16x8*>(rowp + 8 * h) = w8;
            }
        }
        __syncthreads();
        if (k0 + CBK < K) { fetch_a(k0 + CBK); fetch_b(k0 + CBK); }
        // fragments: lane (row = lane & 31, half = lane >> 5) holds k = 16 s + 8 half .. + 7 of its row
        const __bf16* pa = As + (wm * 64 + (lane & 31)) * CSTR + 8 * (lane >> 5);
        const __bf16* pb = Bs + (wn * 64 + (lane & 31)) * CSTR + 8 * (lane >> 5);
#pragma unroll
        for (int s = 0; s < CBK / 16; ++s) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * CSTR + 16 * s);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * CSTR + 16 * s);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(CNT) void conv_fwd_kernel(const ConvArgs g) {
    __shared__ __attribute__((aligned(16))) __bf16 As[CBM * CSTR];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[CBN * CSTR];
    const int m0 = blockIdx.y * CBM, n0 = blockIdx.x * CBN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- A operand: thread -> rows (t / 8) + 32 u, u < 4, and 4 consecutive k = k0 + 4 (t % 8) ..; the rows' pixels are fixed for
    // the whole kernel: image base offset and the top-left input coordinate of the 4 x 4 window, decoded once
    constexpr int TPR = CBK / 4, RPP = CNT / TPR;
    long long a_img[CKU]; int a_y0[CKU], a_x0[CKU]; bool a_ok[CKU];
#pragma unroll
    for (int u = 0; u < CKU; ++u) {
        const int row = m0 + t / TPR + RPP * u;
        a_ok[u] = row < g.M;
        const int rc = a_ok[u] ? row : 0, n = rc / (g.Ho * g.Wo), ij = rc % (g.Ho * g.Wo);
        a_img[u] = (long long)n * g.H * g.W * g.Cin;
        a_y0[u] = 2 * (ij / g.Wo) - 1; a_x0[u] = 2 * (ij % g.Wo) - 1;
    }
    const bool cin4 = g.Cin % 4 == 0 && (reinterpret_cast<uintptr_t>(g.x) & 15) == 0;
    float ra[CKU][4], rb[CKU][4];
    auto fetch_a = [&](int k0) {
        const int k = k0 + (t % TPR) * 4;
        if (cin4) {                                        // the 4 k share a tap and are 4 consecutive channels: one 16-byte load
            const int tap = k / g.Cin, c = k % g.Cin, kh = tap >> 2, kw = tap & 3;
#pragma unroll
            for (int u = 0; u < CKU; ++u) {                // unconditional at a clamped address, selected afterwards
                const int yy = a_y0[u] + kh, xx = a_x0[u] + kw;
                const bool in = a_ok[u] && k < g.K && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                const int yc = min(max(yy, 0), g.H - 1), xc = min(max(xx, 0), g.W - 1), cc = min(c, g.Cin - 4);
                const float4 f = *reinterpret_cast<const float4*>(g.x + a_img[u] + ((long long)yc * g.W + xc) * g.Cin + cc);
                ra[u][0] = in ? f.x : 0.f; ra[u][1] = in ? f.y : 0.f; ra[u][2] = in ? f.z : 0.f; ra[u][3] = in ? f.w : 0.f;
            }
        } else {                                           // any channel count (the first layer: 1 channel, K = 16)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = min(k + e, g.K - 1), tap = ke / g.Cin, c = ke % g.Cin, kh = tap >> 2, kw = tap & 3;
#pragma unroll
                for (int u = 0; u < CKU; ++u) {
                    const int yy = a_y0[u] + kh, xx = a_x0[u] + kw;
                    const bool in = a_ok[u] && k + e < g.K && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                    const int yc = min(max(yy, 0), g.H - 1), xc = min(max(xx, 0), g.W - 1);
                    const float f = g.x[a_img[u] + ((long long)yc * g.W + xc) * g.Cin + c];
                    ra[u][e] = in ? f : 0.f;
                }
            }
        }
    };
    // ---- B operand: the HWIO kernel array as [k][o]: thread -> ONE column o = n0 + (t & 127) and 16 consecutive k (gemm_bf16.hip)
    auto fetch_b = [&](int k0) {
        const int col = n0 + (t & 127), kb = k0 + (CBK / 2) * (t >> 7), colc = min(col, g.N - 1);
#pragma unroll
        for (int u = 0; u < CKU; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = kb + 4 * u + c;
                const float f = g.w[(long long)min(k, g.K - 1) * g.N + colc];
                rb[u][c] = (k < g.K && col < g.N) ? f : 0.f;
            }
    };
    conv_mainloop<false>(As, Bs, g.K, ra, rb, fetch_a, fetch_b, acc);
    // ---- epilogue: bias, relu; the [pixel][o] tile is the NHWC output
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= g.N) continue;
        const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                float v = acc[i][j][r] + bias;
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.mask) v = g.mask[(long long)row * g.N + col] > 0.f ? v : 0.f;
                g.y[(long long)row * g.N + col] = v;
            }
    }
}

// ---- transposed convolution = the adjoint of the convolution above with the same kernel array ------------------------------------
//   out[n, P, Q, o] = act(b[o] + sum_{kh, kw, c : (P + 1 - kh), (Q + 1 - kw) even} y[n, (P + 1 - kh) / 2, (Q + 1 - kw) / 2, c] K[kh, kw, o, c])
// (K [4, 4, C_out, C_in]: the HWIO kernel of the convolution it is the adjoint of; oracle: conv_t_fwd.)  For an output pixel of
// parity (pp, qq) = (P % 2, Q % 2) exactly 2 x 2 taps contribute: kh = 1 - pp + 2 th reads input row P' + pp - th (P = 2 P' + pp),
// likewise in x.  So the layer is FOUR GEMMs, one per parity class (blockIdx.z): rows = (n, P', Q'), inner index k = (th, tw, c),
// 4 C_in deep, the A operand gathered from y, the B operand the class's four [o][c] slices of K (k-contiguous: the row form of
// the loader), the result scattered to the class's pixels.  The same kernel is the convolution's INPUT GRADIENT (y := dL/d out
// of the convolution, no bias; `mask`: multiply by [mask > 0], the relu of the layer below).
struct ConvTArgs {
    const float* y; const float* w; const float* bias; const float* mask; float* out;
    int B, h, w_in, Cin, Cout;          // input [B, h, w_in, Cin] -> output [B, 2 h, 2 w_in, Cout]
    int M, N, K;                        // per parity class: M = B h w_in, N = Cout, K = 4 Cin
    int relu;
};

__global__ __launch_bounds__(CNT) void conv_t_fwd_kernel(const ConvTArgs g) {
    __shared__ __attribute__((aligned(16))) __bf16 As[CBM * CSTR];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[CBN * CSTR];
    __shared__ long long rowoff[CBM];                      // output offset of the tile's rows (-1: past the end)
    const int m0 = blockIdx.y * CBM, n0 = blockIdx.x * CBN, pp = blockIdx.z >> 1, qq = blockIdx.z & 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (t < CBM) {
        const int row = m0 + t;
        long long off = -1;
        if (row < g.M) {
            const int n = row / (g.h * g.w_in), ij = row % (g.h * g.w_in), P = 2 * (ij / g.w_in) + pp, Q = 2 * (ij % g.w_in) + qq;
            off = (((long long)n * 2 * g.h + P) * 2 * g.w_in + Q) * g.Cout;
        }
        rowoff[t] = off;
    }
    constexpr int TPR = CBK / 4, RPP = CNT / TPR;
    long long a_img[CKU]; int a_i[CKU], a_j[CKU]; bool a_ok[CKU];
#pragma unroll
    for (int u = 0; u < CKU; ++u) {
        const int row = m0 + t / TPR + RPP * u;
        a_ok[u] = row < g.M;
        const int rc = a_ok[u] ? row : 0, n = rc / (g.h * g.w_in), ij = rc % (g.h * g.w_in);
        a_img[u] = (long long)n * g.h * g.w_in * g.Cin;
        a_i[u] = ij / g.w_in + pp; a_j[u] = ij % g.w_in + qq;         // input pixel = (a_i - th, a_j - tw)
    }
    const bool c4 = g.Cin % 4 == 0 && (reinterpret_cast<uintptr_t>(g.y) & 15) == 0 && (reinterpret_cast<uintptr_t>(g.w) & 15) == 0;
    float ra[CKU][4], rb[CKU][4];
    auto fetch_a = [&](int k0) {
        const int k = k0 + (t % TPR) * 4;
#pragma unroll
        for (int e = 0; e < 4; e += 4) {
            if (c4) {
                const int tap = min(k / g.Cin, 3), c = k % g.Cin, th = tap >> 1, tw = tap & 1;
#pragma unroll
                for (int u = 0; u < CKU; ++u) {            // unconditional at a clamped address, selected afterwards
                    const int yy = a_i[u] - th, xx = a_j[u] - tw;
                    const bool in = a_ok[u] && k < g.K && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w_in;
                    const int yc = min(max(yy, 0), g.h - 1), xc = min(max(xx, 0), g.w_in - 1);
                    const float4 f = *reinterpret_cast<const float4*>(g.y + a_img[u] + ((long long)yc * g.w_in + xc) * g.Cin + c);
                    ra[u][0] = in ? f.x : 0.f; ra[u][1] = in ? f.y : 0.f; ra[u][2] = in ? f.z : 0.f; ra[u][3] = in ? f.w : 0.f;
                }
            } else {
#pragma unroll
                for (int ee = 0; ee < 4; ++ee) {
                    const int ke = min(k + ee, g.K - 1), tap = ke / g.Cin, c = ke % g.Cin, th = tap >> 1, tw = tap & 1;
#pragma unroll
                    for (int u = 0; u < CKU; ++u) {
                        const int yy = a_i[u] - th, xx = a_j[u] - tw;
                        const bool in = a_ok[u] && k + ee < g.K && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w_in;
                        const int yc = min(max(yy, 0), g.h - 1), xc = min(max(xx, 0), g.w_in - 1);
                        const float f = g.y[a_img[u] + ((long long)yc * g.w_in + xc) * g.Cin + c];
                        ra[u][ee] = in ? f : 0.f;
                    }
                }
            }
        }
    };
    // B operand, row form: thread -> output channels o = n0 + t / 8 + 32 u and 4 consecutive k of tap (th, tw): K[kh, kw, o, c ..]
    auto fetch_b = [&](int k0) {
        const int k = k0 + (t % TPR) * 4;
        if (c4) {
            const int tap = min(k / g.Cin, 3), c = k % g.Cin, kh = 1 - pp + 2 * (tap >> 1), kw = 1 - qq + 2 * (tap & 1);
            const float* base = g.w + (long long)(kh * 4 + kw) * g.Cout * g.Cin + c;
#pragma unroll
            for (int u = 0; u < CKU; ++u) {
                const int o = n0 + t / TPR + RPP * u;
                const float4 f = *reinterpret_cast<const float4*>(base + (long long)min(o, g.N - 1) * g.Cin);
                const bool in = o < g.N && k < g.K;
                rb[u][0] = in ? f.x : 0.f; rb[u][1] = in ? f.y : 0.f; rb[u][2] = in ? f.z : 0.f; rb[u][3] = in ? f.w : 0.f;
            }
        } else {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
                const int ke = min(k + ee, g.K - 1), tap = ke / g.Cin, c = ke % g.Cin, kh = 1 - pp + 2 * (tap >> 1), kw = 1 - qq + 2 * (tap & 1);
                const float* base = g.w + (long long)(kh * 4 + kw) * g.Cout * g.Cin + c;
#pragma unroll
                for (int u = 0; u < CKU; ++u) {
                    const int o = n0 + t / TPR + RPP * u;
                    const float f = base[(long long)min(o, g.N - 1) * g.Cin];
                    rb[u][ee] = (o < g.N && k + ee < g.K) ? f : 0.f;
                }
            }
        }
    };
    conv_mainloop<true>(As, Bs, g.K, ra, rb, fetch_a, fetch_b, acc);
    // (rowoff was written before the main loop's first barrier)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= g.N) continue;
        const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long off = rowoff[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
                if (off < 0) continue;
                float v = acc[i][j][r] + bias;
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.mask) v = g.mask[off + col] > 0.f ? v : 0.f;
                g.out[off + col] = v;
            }
    }
}

// ---- kernel gradient -------------------------------------------------------------------------------------------------------------
//   dK[kh, kw, c, o] = sum_{n, i, j} x[n, 2 i + kh - 1, 2 j + kw - 1, c] dy[n, i, j, o];   db[o] = sum_{n, i, j} dy[n, i, j, o]
// GEMM rows m = (kh, kw, c) plus one row of ones (db), columns o, inner index = the output pixels, split over the batch into S
// slabs that a fixed-order sum adds afterwards (no float atomics: bitwise repeatable, like the Dense dW|db).  Both operands have
// the inner index strided in memory: thread = one row (resp. column) and 16 consecutive pixels, staged transposed with 16-byte
// LDS writes.  With (x := dL/d out, dy := the layer's input) the same kernel is the TRANSPOSED layer's kernel gradient
// [kh, kw, C_out, C_in].
struct ConvWArgs {
    const float* x; const float* dy; float* slab;
    int B, H, W, Cin, Cout, Ho, Wo;
    int M, N, K;                        // M = 16 Cin + 1, N = Cout, K = B Ho Wo
    int k_per_split; long long slab_stride;
};

__global__ __launch_bounds__(CNT) void conv_wgrad_kernel(const ConvWArgs g) {
    __shared__ __attribute__((aligned(16))) __bf16 As[CBM * CSTR];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[CBN * CSTR];
    const int m0 = blockIdx.y * CBM, n0 = blockIdx.x * CBN;
    const int kbeg = blockIdx.z * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // A: this thread's row m = (kh, kw, c) -- or the row of ones
    const int m = m0 + (t & 127), mreal = g.M - 1;
    const bool a_row = m < mreal, a_one = m == mreal;
    const int mc = a_row ? m : 0, tap = mc / g.Cin, ch = mc % g.Cin, kh = tap >> 2, kw = tap & 3;
    const int col = n0 + (t & 127), colc = min(col, g.N - 1);
    float ra[CKU][4], rb[CKU][4];
    // pixel -> (image, i, j): shifts when the output sizes are powers of two (config 5: 32 / 16 / 8 / 4), divisions otherwise -- 16
    // pixels per thread and k-tile made the divisions the most expensive thing in this kernel
    const int hw = g.Ho * g.Wo;
    const bool pow2 = (hw & (hw - 1)) == 0 && (g.Wo & (g.Wo - 1)) == 0;
    const int sh_hw = 31 - __builtin_clz(hw), sh_w = 31 - __builtin_clz(g.Wo);
    auto fetch_a = [&](int k0) {
        const int pb = k0 + (CBK / 2) * (t >> 7);
#pragma unroll
        for (int u = 0; u < CKU; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {                  // unconditional at a clamped address, selected afterwards
                const int p = pb + 4 * u + c, pc = min(p, g.K - 1);
                const int n = pow2 ? pc >> sh_hw : pc / hw, ij = pow2 ? pc & (hw - 1) : pc % hw;
                const int yy = 2 * (pow2 ? ij >> sh_w : ij / g.Wo) + kh - 1, xx = 2 * (pow2 ? ij & (g.Wo - 1) : ij % g.Wo) + kw - 1;
                const bool in = a_row && p < kend && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                const int yc = min(max(yy, 0), g.H - 1), xc = min(max(xx, 0), g.W - 1);
                const float f = g.x[(((long long)n * g.H + yc) * g.W + xc) * g.Cin + ch];
                ra[u][c] = in ? f : ((a_one && p < kend) ? 1.f : 0.f);
            }
    };
    auto fetch_b = [&](int k0) {
        const int pb = k0 + (CBK / 2) * (t >> 7);
#pragma unroll
        for (int u = 0; u < CKU; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int p = pb + 4 * u + c;
                const float f = g.dy[(long long)min(p, g.K - 1) * g.N + colc];
                rb[u][c] = (p < kend && col < g.N) ? f : 0.f;
            }
    };
    conv_mainloop<false, false>(As, Bs, kend, ra, rb, fetch_a, fetch_b, acc, kbeg);
    float* C = g.slab + (long long)blockIdx.z * g.slab_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cc = n0 + wn * 64 + j * 32 + (lane & 31);
        if (cc >= g.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M) C[(long long)row * g.N + cc] = acc[i][j][r];
            }
    }
}

// column sums of dy [pixels][C] (the bias gradient of a transposed layer: its dL/d out summed over the pixels): block s sums its
// rows per column, a fixed-order sum over the blocks follows (launch_sum_slabs)
__global__ __launch_bounds__(256) void conv_colsum_kernel(const float* dy, float* partial, long long pixels, int C, long long rows_per_split) {
    extern __shared__ float sh[];   // 256 floats
    const long long r0 = blockIdx.x * rows_per_split, r1 = min(pixels, r0 + rows_per_split);
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int W = min(256, C - c0), G = 256 / W, col = threadIdx.x % W, grp = threadIdx.x / W;
        float acc = 0.f;
        if (grp < G)
            for (long long r = r0 + grp; r < r1; r += G) acc += dy[r * C + c0 + col];
        sh[threadIdx.x] = grp < G ? acc : 0.f;
        __syncthreads();
        if ((int)threadIdx.x < W) {
            float tsum = 0.f;
            for (int g2 = 0; g2 < G; ++g2) tsum += sh[g2 * W + threadIdx.x];
            partial[(long long)blockIdx.x * C + c0 + threadIdx.x] = tsum;
        }
        __syncthreads();
    }
}

// the same sums as a flat 16-byte stream (C a power of two <= 1024, pixels * C a multiple of 4): thread t of a block always sees the
// same 4 columns (4 t mod C); four independent accumulators, then the block's threads of one column quad meet through LDS
__global__ __launch_bounds__(256) void conv_colsum4_kernel(const float4* dy4, float* partial, long long n4, int C, long long chunk) {
    __shared__ float4 sh[256];
    const long long f0 = blockIdx.x * chunk, f1 = min(n4, f0 + chunk);
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    long long f = f0 + threadIdx.x;
    for (; f + 768 < f1; f += 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 v = dy4[f + 256 * u];
            a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
        }
    }
    for (; f < f1; f += 256) {
        const float4 v = dy4[f];
        a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
    }
    sh[threadIdx.x] = make_float4((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y),
                                  (a[0].z + a[1].z) + (a[2].z + a[3].z), (a[0].w + a[1].w) + (a[2].w + a[3].w));
    __syncthreads();
    const int Q = C >= 4 ? C / 4 : 1;                    // column quads; Q divides 256
    if ((int)threadIdx.x < Q) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g2 = threadIdx.x; g2 < 256; g2 += Q) { const float4 v = sh[g2]; t4.x += v.x; t4.y += v.y; t4.z += v.z; t4.w += v.w; }
        float* o = partial + (long long)blockIdx.x * C;
        if (C >= 4) *reinterpret_cast<float4*>(o + 4 * threadIdx.x) = t4;
        else if (C == 2) { o[0] = t4.x + t4.z; o[1] = t4.y + t4.w; }
        else o[0] = (t4.x + t4.y) + (t4.z + t4.w);
    }
}

// column sums of a bf16 tensor as a flat 16-byte stream (C a power of two >= 8, <= 2048; pixels * C a multiple of 8): thread t of a
// block always sees the same 8 columns (8 t mod C); two independent accumulator sets, then the block's threads of one column
// octet meet through LDS.  float32 sums of the bf16 values, fixed order.
__global__ __launch_bounds__(256) void conv_colsum8_bf16_kernel(const uint4* dy8, float* partial, long long n8, int C, long long chunk) {
    __shared__ float sh[256][9];
    const long long f0 = blockIdx.x * chunk, f1 = min(n8, f0 + chunk);
    float a[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[u][e] = 0.f;
    auto add8 = [&](float (&acc)[8], const uint4 v) {
        acc[0] += __builtin_bit_cast(float, v.x << 16); acc[1] += __builtin_bit_cast(float, v.x & 0xffff0000u);
        acc[2] += __builtin_bit_cast(float, v.y << 16); acc[3] += __builtin_bit_cast(float, v.y & 0xffff0000u);
        acc[4] += __builtin_bit_cast(float, v.z << 16); acc[5] += __builtin_bit_cast(float, v.z & 0xffff0000u);
        acc[6] += __builtin_bit_cast(float, v.w << 16); acc[7] += __builtin_bit_cast(float, v.w & 0xffff0000u);
    };
    long long f = f0 + threadIdx.x;
    for (; f + 768 < f1; f += 1024) {
        const uint4 v0 = dy8[f], v1 = dy8[f + 256], v2 = dy8[f + 512], v3 = dy8[f + 768];
        add8(a[0], v0); add8(a[1], v1); add8(a[0], v2); add8(a[1], v3);
    }
    for (; f < f1; f += 256) add8(a[0], dy8[f]);
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[threadIdx.x][e] = a[0][e] + a[1][e];
    __syncthreads();
    const int Q = C / 8;                                 // column octets; Q divides 256
    if ((int)threadIdx.x < Q) {
        float t8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t8[e] = 0.f;
        for (int g2 = threadIdx.x; g2 < 256; g2 += Q)
#pragma unroll
            for (int e = 0; e < 8; ++e) t8[e] += sh[g2][e];
        float* o = partial + (long long)blockIdx.x * C + 8 * threadIdx.x;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = t8[e];
    }
}

static int conv_wgrad_splits(long long pixels, int M, int N) {
    const long long tiles = (long long)((M + CBM - 1) / CBM) * ((N + CBN - 1) / CBN);
    long long S = std::max(1ll, 1024 / tiles);
    S = std::min(S, std::max(1ll, pixels / 256));          // at least 256 pixels (8 k-tiles) per split
    return (int)std::min(S, 4096ll);
}

// ---- the one-channel layers (the first convolution, the last transposed convolution and their gradients): streaming kernels --------
// 16 or 128 multiply-adds per output value against 4 to 128 bytes of traffic: HBM-bound, nothing for the matrix cores.  Exact f32.
struct ThinArgs {
    const float* x; const float* w; const float* bias; const float* mask; float* y; __bf16* y16;
    int B, H, W, C, Ho, Wo; long long pixels;          // C: the wide side's channel count
    // lean forms (round 3): the WIDE tensor as its bf16 copy instead -- x16: the transposed forward's input (4s kernel); mask16: the
    // forward's relu-mask source / the kernel gradient's dy (matrix-core forms)
    const __bf16* x16; const __bf16* mask16;
};
// forward, C_in = 1: y[p, o] = act(b[o] + sum_taps x_window[p, tap] K[tap, o]).  Thread = (pixel lane, o); C | 256.
__global__ __launch_bounds__(256) void thin_conv_fwd_kernel(const ThinArgs g, const int relu) {
    const int C = g.C, o = threadIdx.x % C, pl = threadIdx.x / C, ppb = 256 / C;
    float wk[16];
#pragma unroll
    for (int tp = 0; tp < 16; ++tp) wk[tp] = g.w[tp * C + o];
    const float b = g.bias ? g.bias[o] : 0.f;
    for (long long p = (long long)blockIdx.x * ppb + pl; p < g.pixels; p += (long long)gridDim.x * ppb) {
        const int n = (int)(p / (g.Ho * g.Wo)), ij = (int)(p % (g.Ho * g.Wo)), y0 = 2 * (ij / g.Wo) - 1, x0 = 2 * (ij % g.Wo) - 1;
        const float* img = g.x + (long long)n * g.H * g.W;
        float acc = b;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int yy = y0 + kh, xx = x0 + kw;
                const bool in = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                const float v = img[(long long)min(max(yy, 0), g.H - 1) * g.W + min(max(xx, 0), g.W - 1)];
                acc = fmaf(in ? v : 0.f, wk[kh * 4 + kw], acc);
            }
        if (relu) acc = fmaxf(acc, 0.f);
        if (g.mask) acc = g.mask[p * C + o] > 0.f ? acc : 0.f;
        g.y[p * C + o] = acc;
    }
}
// transposed forward, C_out = 1: out[n, P, Q] = act(b + sum over the 2 x 2 contributing taps and c of y[n, i, j, c] K[kh, kw, 0, c]).
// Thread = one output pixel; the 16 x C kernel slice in LDS; C % 4 == 0, C <= 256.
__global__ __launch_bounds__(256) void thin_conv_t_fwd_kernel(const ThinArgs g, const int relu) {
    __shared__ __attribute__((aligned(16))) float ks[16 * 256];
    const int C = g.C;
    for (int e = threadIdx.x; e < 16 * C; e += 256) ks[e] = g.w[e];
    __syncthreads();
    const float b = g.bias ? g.bias[0] : 0.f;
    const int Hout = 2 * g.H, Wout = 2 * g.W;           // here H, W are the INPUT sizes
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < g.pixels; p += (long long)gridDim.x * 256) {
        const int n = (int)(p / ((long long)Hout * Wout)), PQ = (int)(p % ((long long)Hout * Wout)), P = PQ / Wout, Q = PQ % Wout;
        const int pp = P & 1, qq = Q & 1;
        float acc = b;
#pragma unroll
        for (int th = 0; th < 2; ++th)
#pragma unroll
            for (int tw = 0; tw < 2; ++tw) {
                const int i = (P >> 1) + pp - th, j = (Q >> 1) + qq - tw, kh = 1 - pp + 2 * th, kw = 1 - qq + 2 * tw;
                const bool in = i >= 0 && i < g.H && j >= 0 && j < g.W;
                const float4* src = reinterpret_cast<const float4*>(g.x + (((long long)n * g.H + min(max(i, 0), g.H - 1)) * g.W + min(max(j, 0), g.W - 1)) * C);
                const float4* kk = reinterpret_cast<const float4*>(ks + (kh * 4 + kw) * C);
                float part = 0.f;
                for (int c4 = 0; c4 < C / 4; ++c4) {
                    const float4 v = src[c4], k4 = kk[c4];
                    part = fmaf(v.x, k4.x, part); part = fmaf(v.y, k4.y, part); part = fmaf(v.z, k4.z, part); part = fmaf(v.w, k4.w, part);
                }
                acc += in ? part : 0.f;
            }
        if (relu) acc = fmaxf(acc, 0.f);
        if (g.mask) acc = g.mask[p] > 0.f ? acc : 0.f;
        g.y[p] = acc;
    }
}
// The same layer for C a power of two in [4, 256]: C / 4 lanes per INPUT-grid pixel, each with its 4 channels of the 16 kernel taps
// in registers; the 3 x 3 input neighbourhood (9 coalesced 16-byte loads) feeds the pixel's 2 x 2 output block, the lanes of a
// pixel meet by a butterfly, lane 0 stores the four values.
__global__ __launch_bounds__(256) void thin_conv_t_fwd4_kernel(const ThinArgs g, const int relu) {
    const int C = g.C, LP = C / 4;
    unsigned lb = blockIdx.x;                            // XCD-aware: each XCD (own L2) walks a contiguous pixel range -- the 3 x 3
    if (gridDim.x % 8 == 0) lb = (lb % 8) * (gridDim.x / 8) + lb / 8;      // neighbourhoods then re-read from L2, not 3x from HBM (PMC)
    const long long gid = (long long)lb * 256 + threadIdx.x, groups = (long long)gridDim.x * 256 / LP;
    const int lg = (int)(gid % LP);
    float4 kk[16];
#pragma unroll
    for (int tp = 0; tp < 16; ++tp) kk[tp] = *reinterpret_cast<const float4*>(g.w + tp * C + 4 * lg);
    const float b = g.bias ? g.bias[0] : 0.f;
    const long long in_px = (long long)g.B * g.H * g.W, rounds = (in_px + groups - 1) / groups;
    for (long long it = 0; it < rounds; ++it) {           // every lane runs every round: the butterfly needs the whole group
        const long long p = gid / LP + it * groups;
        const bool live = p < in_px;
        const long long pc = live ? p : in_px - 1;
        const int n = (int)(pc / (g.H * g.W)), ij = (int)(pc % (g.H * g.W)), i = ij / g.W, j = ij % g.W;
        float4 v[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = i - 1 + dy, xx = j - 1 + dx;
                const bool in = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                const float4 f = *reinterpret_cast<const float4*>(g.x + (((long long)n * g.H + min(max(yy, 0), g.H - 1)) * g.W + min(max(xx, 0), g.W - 1)) * C + 4 * lg);
                v[dy][dx] = in ? f : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float o[2][2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                float acc = 0.f;
#pragma unroll
                for (int th = 0; th < 2; ++th)
#pragma unroll
                    for (int tw = 0; tw < 2; ++tw) {      // input pixel (i + pp - th, j + qq - tw) = v[1 + pp - th][1 + qq - tw]
                        const float4 f = v[1 + pp - th][1 + qq - tw], k4 = kk[(1 - pp + 2 * th) * 4 + (1 - qq + 2 * tw)];
                        acc = fmaf(f.x, k4.x, acc); acc = fmaf(f.y, k4.y, acc); acc = fmaf(f.z, k4.z, acc); acc = fmaf(f.w, k4.w, acc);
                    }
                for (int sft = LP >> 1; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
                o[pp][qq] = acc + b;
            }
        if (live && lg == 0) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const long long off = ((long long)n * 2 * g.H + 2 * i + pp) * 2 * g.W + 2 * j;
                float2 r2 = make_float2(o[pp][0], o[pp][1]);
                if (relu) { r2.x = fmaxf(r2.x, 0.f); r2.y = fmaxf(r2.y, 0.f); }
                if (g.mask) { const float2 m2 = *reinterpret_cast<const float2*>(g.mask + off); r2.x = m2.x > 0.f ? r2.x : 0.f; r2.y = m2.y > 0.f ? r2.y : 0.f; }
                *reinterpret_cast<float2*>(g.y + off) = r2;
            }
        }
    }
}
// The strip form (input width a multiple of 4): C / 4 lanes take FOUR input pixels of a row at once -- 3 x 6 neighbourhood loads for 16
// outputs instead of 4 x 9, one index decode and one set of row / column bounds per strip.  (The one-pixel form above spent 290
// instructions per lane and pixel, most of them addresses and bounds: it ran at 1.7 TB/s with its traffic already at the minimum.)
// X16: the input is read from its bf16 copy (a compile-time choice: a load under a run-time condition is a wait where it is issued)
template <bool X16>
__global__ __launch_bounds__(256) void thin_conv_t_fwd4s_kernel(const ThinArgs g, const int relu) {
    const int C = g.C, LP = C / 4, SW = g.W / 4;         // strips per input row
    unsigned lb = blockIdx.x;
    if (gridDim.x % 8 == 0) lb = (lb % 8) * (gridDim.x / 8) + lb / 8;      // XCD-contiguous, as above
    const long long gid = (long long)lb * 256 + threadIdx.x, groups = (long long)gridDim.x * 256 / LP;
    const int lg = (int)(gid % LP);
    float4 kk[16];
#pragma unroll
    for (int tp = 0; tp < 16; ++tp) kk[tp] = *reinterpret_cast<const float4*>(g.w + tp * C + 4 * lg);
    const float b = g.bias ? g.bias[0] : 0.f;
    const long long strips = (long long)g.B * g.H * SW, rounds = (strips + groups - 1) / groups;
    for (long long it = 0; it < rounds; ++it) {
        const long long sidx = gid / LP + it * groups;
        const bool live = sidx < strips;
        const long long sc = live ? sidx : strips - 1;
        const int n = (int)(sc / (g.H * SW)), rem = (int)(sc % (g.H * SW)), i = rem / SW, j0 = 4 * (rem % SW);
        float4 v[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = i - 1 + r;
            const bool rv = yy >= 0 && yy < g.H;
            const long long rowo = ((long long)n * g.H + min(max(yy, 0), g.H - 1)) * g.W * C + 4 * lg;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int xx = j0 - 1 + c;
                const bool in = rv && xx >= 0 && xx < g.W;
                const long long eo = rowo + (long long)min(max(xx, 0), g.W - 1) * C;
                float4 f;
                if constexpr (X16) {                        // (four bf16 widened)
                    const uint2 u = *reinterpret_cast<const uint2*>(g.x16 + eo);
                    f = make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                                    __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
                } else f = *reinterpret_cast<const float4*>(g.x + eo);
                v[r][c] = in ? f : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float o[2][8];                                    // [output row parity][8 consecutive output columns]
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    float acc = 0.f;
#pragma unroll
                    for (int th = 0; th < 2; ++th)
#pragma unroll
                        for (int tw = 0; tw < 2; ++tw) {
                            const float4 f = v[1 + pp - th][1 + px + qq - tw], k4 = kk[(1 - pp + 2 * th) * 4 + (1 - qq + 2 * tw)];
                            acc = fmaf(f.x, k4.x, acc); acc = fmaf(f.y, k4.y, acc); acc = fmaf(f.z, k4.z, acc); acc = fmaf(f.w, k4.w, acc);
                        }
                    for (int sft = LP >> 1; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
                    o[pp][2 * px + qq] = acc + b;
                }
        if (live && lg == 0) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const long long off = ((long long)n * 2 * g.H + 2 * i + pp) * 2 * g.W + 2 * j0;       // 8 floats: 32-byte aligned
                float r8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) r8[e] = relu ? fmaxf(o[pp][e], 0.f) : o[pp][e];
                if (g.mask) {
                    const float4 m0 = *reinterpret_cast<const float4*>(g.mask + off), m1 = *reinterpret_cast<const float4*>(g.mask + off + 4);
                    r8[0] = m0.x > 0.f ? r8[0] : 0.f; r8[1] = m0.y > 0.f ? r8[1] : 0.f; r8[2] = m0.z > 0.f ? r8[2] : 0.f; r8[3] = m0.w > 0.f ? r8[3] : 0.f;
                    r8[4] = m1.x > 0.f ? r8[4] : 0.f; r8[5] = m1.y > 0.f ? r8[5] : 0.f; r8[6] = m1.z > 0.f ? r8[6] : 0.f; r8[7] = m1.w > 0.f ? r8[7] : 0.f;
                }
                *reinterpret_cast<float4*>(g.y + off) = make_float4(r8[0], r8[1], r8[2], r8[3]);
                *reinterpret_cast<float4*>(g.y + off + 4) = make_float4(r8[4], r8[5], r8[6], r8[7]);
            }
        }
    }
}
// kernel gradient with a one-channel gathered tensor: part[block][tap or 16 = bias][o] = sum over the block's pixels of
// x_window[p, tap] dy[p, o]; thread = (pixel lane, o), the lanes of a block meet through LDS, a fixed-order sum over the blocks follows
__global__ __launch_bounds__(256) void thin_conv_wgrad_kernel(const ThinArgs g, float* part) {
    __shared__ float red[256 * 17];
    const int C = g.C, o = threadIdx.x % C, pl = threadIdx.x / C, ppb = 256 / C;
    float acc[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) acc[k] = 0.f;
    const long long per = (g.pixels + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = min(g.pixels, p0 + per);
    for (long long p = p0 + pl; p < p1; p += ppb) {
        const int n = (int)(p / (g.Ho * g.Wo)), ij = (int)(p % (g.Ho * g.Wo)), y0 = 2 * (ij / g.Wo) - 1, x0 = 2 * (ij % g.Wo) - 1;
        const float* img = g.x + (long long)n * g.H * g.W;
        const float d = g.mask[p * C + o];               // (mask: the C-channel tensor here)
        acc[16] += d;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int yy = y0 + kh, xx = x0 + kw;
                const bool in = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                const float v = img[(long long)min(max(yy, 0), g.H - 1) * g.W + min(max(xx, 0), g.W - 1)];
                acc[kh * 4 + kw] = fmaf(in ? v : 0.f, d, acc[kh * 4 + kw]);
            }
    }
#pragma unroll
    for (int k = 0; k < 17; ++k) red[(k * ppb + pl) * C + o] = acc[k];
    __syncthreads();
    for (int e = threadIdx.x; e < 17 * C; e += 256) {
        const int k = e / C, oo = e % C;
        float sum = 0.f;
        for (int l = 0; l < ppb; ++l) sum += red[(k * ppb + l) * C + oo];
        part[(long long)blockIdx.x * 17 * C + e] = sum;
    }
}
// The same two kernels for C a multiple of 8 that divides 256: thread = (pixel, group of 8 channels) -- the 16 window loads feed 128
// multiply-adds instead of 16, 32-byte runs per lane on the wide tensor.
__global__ __launch_bounds__(256) void thin_conv_fwd8_kernel(const ThinArgs g, const int relu) {
    const int C = g.C, G = C / 8, cg = threadIdx.x % G, pl = threadIdx.x / G, ppb = 256 / G, o0 = cg * 8;
    float wk[16][8], b[8];
#pragma unroll
    for (int tp = 0; tp < 16; ++tp) {
        const float4 u = *reinterpret_cast<const float4*>(g.w + tp * C + o0), v = *reinterpret_cast<const float4*>(g.w + tp * C + o0 + 4);
        wk[tp][0] = u.x; wk[tp][1] = u.y; wk[tp][2] = u.z; wk[tp][3] = u.w; wk[tp][4] = v.x; wk[tp][5] = v.y; wk[tp][6] = v.z; wk[tp][7] = v.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) b[e] = g.bias ? g.bias[o0 + e] : 0.f;
    for (long long p = (long long)blockIdx.x * ppb + pl; p < g.pixels; p += (long long)gridDim.x * ppb) {
        const int n = (int)(p / (g.Ho * g.Wo)), ij = (int)(p % (g.Ho * g.Wo)), y0 = 2 * (ij / g.Wo) - 1, x0 = 2 * (ij % g.Wo) - 1;
        const float* img = g.x + (long long)n * g.H * g.W;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = b[e];
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int yy = y0 + kh, xx = x0 + kw;
                const bool in = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                float v = img[(long long)min(max(yy, 0), g.H - 1) * g.W + min(max(xx, 0), g.W - 1)];
                v = in ? v : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(v, wk[kh * 4 + kw][e], acc[e]);
            }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
        }
        if (g.mask) {
            const float4 u = *reinterpret_cast<const float4*>(g.mask + p * C + o0), v = *reinterpret_cast<const float4*>(g.mask + p * C + o0 + 4);
            acc[0] = u.x > 0.f ? acc[0] : 0.f; acc[1] = u.y > 0.f ? acc[1] : 0.f; acc[2] = u.z > 0.f ? acc[2] : 0.f; acc[3] = u.w > 0.f ? acc[3] : 0.f;
            acc[4] = v.x > 0.f ? acc[4] : 0.f; acc[5] = v.y > 0.f ? acc[5] : 0.f; acc[6] = v.z > 0.f ? acc[6] : 0.f; acc[7] = v.w > 0.f ? acc[7] : 0.f;
        }
        *reinterpret_cast<float4*>(g.y + p * C + o0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(g.y + p * C + o0 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        if (g.y16) {
            typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
            const bf16x8_t o8 = {(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3], (__bf16)acc[4], (__bf16)acc[5], (__bf16)acc[6], (__bf16)acc[7]};
            *reinterpret_cast<bf16x8_t*>(g.y16 + p * C + o0) = o8;
        }
    }
}
// part[block][17][C] as above; the lanes of a wave that share a channel group meet by DPP-free shuffles, the 4 waves through LDS
__global__ __launch_bounds__(256) void thin_conv_wgrad8_kernel(const ThinArgs g, float* part) {
    extern __shared__ float red[];                       // [4 waves][17][C]
    const int C = g.C, G = C / 8, cg = threadIdx.x % G, pl = threadIdx.x / G, ppb = 256 / G, o0 = cg * 8;
    float acc[17][8];
#pragma unroll
    for (int k = 0; k < 17; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    const long long per = (g.pixels + gridDim.x - 1) / gridDim.x, p0 = blockIdx.x * per, p1 = min(g.pixels, p0 + per);
    for (long long p = p0 + pl; p < p1; p += ppb) {
        const int n = (int)(p / (g.Ho * g.Wo)), ij = (int)(p % (g.Ho * g.Wo)), y0 = 2 * (ij / g.Wo) - 1, x0 = 2 * (ij % g.Wo) - 1;
        const float* img = g.x + (long long)n * g.H * g.W;
        const float4 u = *reinterpret_cast<const float4*>(g.mask + p * C + o0), v4 = *reinterpret_cast<const float4*>(g.mask + p * C + o0 + 4);
        const float d[8] = {u.x, u.y, u.z, u.w, v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[16][e] += d[e];
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int yy = y0 + kh, xx = x0 + kw;
                const bool in = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
                float v = img[(long long)min(max(yy, 0), g.H - 1) * g.W + min(max(xx, 0), g.W - 1)];
                v = in ? v : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[kh * 4 + kw][e] = fmaf(v, d[e], acc[kh * 4 + kw][e]);
            }
    }
    // lanes l, l + G, l + 2 G, ... of a wave hold the same channel group (G divides 32): a fixed butterfly over the lane bits above it
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 17; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[k][e];
            for (int sft = 32; sft >= G; sft >>= 1) v += __shfl_xor(v, sft, 64);
            acc[k][e] = v;
        }
    if (lane < G) {
#pragma unroll
        for (int k = 0; k < 17; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(wave * 17 + k) * C + o0 + e] = acc[k][e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 17 * C; e += 256)
        part[(long long)blockIdx.x * 17 * C + e] = (red[e] + red[17 * C + e]) + (red[2 * 17 * C + e] + red[3 * 17 * C + e]);
}
// The kernel gradient again, on the f32 matrix cores (C a multiple of 32, power-of-two output sizes): it IS a GEMM with M = 16 taps + a
// row of ones, N = C, K = pixels; v_mfma_f32_32x32x2_f32 takes two pixels per instruction -- lane (i, k) supplies x at tap i of pixel k
// (one gathered dword), lane (j, k) supplies dy[pixel k][j] (a coalesced row) -- and the 17 x 8 accumulators per thread of the VALU
// form become 16 registers per lane, so many more waves stream at once.  blockIdx.y = the block of 32 channels.
template <bool DY16>                                     // dy read from its bf16 copy (compile-time, as above)
__global__ __launch_bounds__(256) void thin_conv_wgrad_mfma_kernel(const ThinArgs g, float* part, const int sh_hw, const int sh_w) {
    __shared__ float red[4][17 * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, k = lane >> 5, c0 = blockIdx.y * 32;
    const int kh = (i >> 2) & 3, kw = i & 3;
    f32x16 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    const long long pairs = (g.pixels + 1) / 2, per = (pairs + gridDim.x - 1) / gridDim.x;
    const long long q0 = blockIdx.x * per, q1 = min(pairs, q0 + per);
    auto a_val = [&](long long p, bool live) -> float {     // x at tap i of pixel p (ones row 16, zero rows above)
        const int n = (int)(p >> sh_hw), ij = (int)(p & ((1 << sh_hw) - 1)), yy = 2 * (ij >> sh_w) - 1 + kh, xx = 2 * (ij & ((1 << sh_w) - 1)) - 1 + kw;
        const bool in = live && i < 16 && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        const float v = g.x[((long long)n * g.H + min(max(yy, 0), g.H - 1)) * g.W + min(max(xx, 0), g.W - 1)];
        return in ? v : (live && i == 16 ? 1.f : 0.f);
    };
    for (long long q = q0 + wave; q < q1; q += 32) {       // 4 waves x 8 pairs in flight per wave (2 KB of dy: HBM needs ~16 MB in flight chip-wide)
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long qq = q + 4 * u, p = min(2 * qq + k, g.pixels - 1);
            const bool live = qq < q1 && 2 * qq + k < g.pixels;
            av[u] = a_val(p, live);
            // (bf16: a DWORD per lane -- channel pair i & ~1 -- and this lane's half of it: 2-byte loads measured 60 % slower than the float32 ones)
            float d;
            if constexpr (DY16) {
                const unsigned w2 = reinterpret_cast<const unsigned*>(g.mask16)[(p * g.C + c0 + i) >> 1];
                d = __builtin_bit_cast(float, (i & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
            } else d = g.mask[p * g.C + c0 + i];
            bv[u] = live ? d : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc[u & 1], 0, 0, 0);
    }
    // register r of a lane: row (r & 3) + 8 (r >> 2) + 4 k (the tap), column i (the channel)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * k;
        if (row < 17) red[wave][row * 32 + i] = acc[0][r] + acc[1][r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 17 * 32; e += 256) {
        const int row = e >> 5, col = e & 31;
        part[(long long)blockIdx.x * 17 * g.C + row * g.C + c0 + col] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}
// The C_in = 1 forward on the f32 matrix cores (C a multiple of 32, power-of-two output sizes): rows = 32 pixels, k = the 16 taps,
// columns = 32 channels; 8 gathered dwords per lane feed 8 v_mfma_f32_32x32x2_f32, the 16 x 32 weights of the channel block sit in 8
// registers, and a result register is one pixel's 32 channels across 32 lanes -- 128-byte rows for the stores, the mask and the bf16
// copy.  ~50 VGPRs instead of 190: four times the waves, i.e. the bytes in flight this HBM-bound layer was missing.
template <bool M16>                                      // the relu-mask source read from its bf16 copy (compile-time, as above)
__global__ __launch_bounds__(256) void thin_conv_fwd_mfma_kernel(const ThinArgs g, const int relu, const int sh_hw, const int sh_w) {
    const int lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5, c0 = blockIdx.y * 32;
    float bw[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) bw[s2] = g.w[(2 * s2 + kk) * g.C + c0 + i];
    const float bias = g.bias ? g.bias[c0 + i] : 0.f;
    const long long tiles = (g.pixels + 31) / 32, wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long tile = wave_id; tile < tiles; tile += nwaves) {
        const long long p0 = tile * 32, p = min(p0 + i, g.pixels - 1);
        const int n = (int)(p >> sh_hw), ij = (int)(p & ((1 << sh_hw) - 1)), y0 = 2 * (ij >> sh_w) - 1, x0 = 2 * (ij & ((1 << sh_w) - 1)) - 1 + kk;
        const float* img = g.x + (long long)n * g.H * g.W;
        float av[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {                   // tap 2 s2 + kk: kh = s2 >> 1, kw = 2 (s2 & 1) + kk
            const int yy = y0 + (s2 >> 1), xx = x0 + 2 * (s2 & 1);
            const bool in = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
            const float v = img[(long long)min(max(yy, 0), g.H - 1) * g.W + min(max(xx, 0), g.W - 1)];
            av[s2] = in ? v : 0.f;
        }
        float mk[16];
        const bool masked = g.mask || g.mask16;
        if constexpr (M16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned w2 = reinterpret_cast<const unsigned*>(g.mask16)[(min(p0 + (r & 3) + 8 * (r >> 2) + 4 * kk, g.pixels - 1) * g.C + c0 + i) >> 1];
                mk[r] = __builtin_bit_cast(float, (i & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
            }
        } else if (g.mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mk[r] = g.mask[min(p0 + (r & 3) + 8 * (r >> 2) + 4 * kk, g.pixels - 1) * g.C + c0 + i];
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bw[s2], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long q = p0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            float v = acc[r];
            if (relu) v = fmaxf(v, 0.f);
            if (masked) v = mk[r] > 0.f ? v : 0.f;
            const float nb = __shfl_down(v, 1, 64);        // the odd neighbour's channel: even lanes store bf16 pairs
            if (q < g.pixels) {
                if (g.y) g.y[q * g.C + c0 + i] = v;
                if (g.y16 && !(i & 1)) {
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
                    const bf16x2_t pr = {(__bf16)v, (__bf16)nb};
                    *reinterpret_cast<bf16x2_t*>(g.y16 + q * g.C + c0 + i) = pr;
                }
            }
        }
    }
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// the gathering loaders' 16 bytes of zeros (taps outside the image): a module-scope device array, zero from load time on -- nothing to
// launch per call
__device__ __attribute__((aligned(256))) __bf16 g_conv_zero_page[128];
static const __bf16* conv_zero_page() {
    static thread_local const __bf16* p = nullptr;
    static thread_local int dev_of_p = -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (!p || dev != dev_of_p) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_conv_zero_page)) != hipSuccess) return nullptr;
        p = static_cast<const __bf16*>(q); dev_of_p = dev;
    }
    return p;
}
static bool thin_channels_ok(int c) { return c >= 1 && c <= 256 && 256 % c == 0; }
static bool thin_groups_ok(int c) { return c % 8 == 0 && c <= 256 && 256 % c == 0; }     // c / 8 divides 32: lanes of a group stay in a wave
constexpr int kThinWgradBlocks = 2048;

}  // namespace vaek

using namespace vaek;

// The fast form (channel counts multiples of 8, power-of-two output sizes): bf16 copies of the two tensors, then the bf16-storage dW
// kernel of gemm_bf16s.hip with its loader gathering the (kh, kw, c) columns from the image (LDS-DMA, transposed LDS reads, a
// ring of k-tiles): 2.2 ms -> 0.4 ms per call at config 5's layer shapes.  Workspace: [256 B unused | x bf16 | dy bf16 | slabs].
struct ConvWFast { bool ok; int S, rps; size_t off_x, off_dy, off_slab, bytes; };
static ConvWFast conv_wgrad_fast(long long batch, int H, int W, int Cin, int Cout) {
    ConvWFast f{};
    const int Ho = H / 2, Wo = W / 2, hw = Ho * Wo;
    const long long pixels = batch * hw;
    f.ok = Cin % 8 == 0 && Cout % 8 == 0 && (hw & (hw - 1)) == 0 && (Wo & (Wo - 1)) == 0 && pixels >= 64 && pixels < 0x7fffffffll;
    if (!f.ok) return f;
    const long long tiles = (long long)((16 * Cin + 127) / 128) * ((Cout + 127) / 128);
    long long S = std::max(1ll, 1024 / tiles);            // ~1024 workgroups; more splits only add slab traffic (2 x S x slab bytes)
    S = std::min(S, std::max(1ll, pixels / 512));
    f.rps = (int)(((pixels + S - 1) / S + 63) / 64 * 64);
    f.S = (int)((pixels + f.rps - 1) / f.rps);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    f.off_x = 256; f.off_dy = f.off_x + up((size_t)batch * H * W * Cin * 2); f.off_slab = f.off_dy + up((size_t)pixels * Cout * 2);
    f.bytes = f.off_slab + (size_t)f.S * (16 * Cin + 1) * Cout * sizeof(float);
    return f;
}

// The fast form of the forward / transposed forward (C_in a power of two >= 8 / 16, C_out a multiple of 32, a workspace given):
// bf16 copies of the tensor and the kernel, then gemm_bf16s.hip's LDS-DMA GEMM with the gather in its loader (hs_conv_kernel).
// Workspace: [256 B unused | x bf16 | kernel bf16].
struct ConvFFast { bool ok; size_t off_x, off_w, bytes; };
static ConvFFast conv_fwd_fast(int mode, long long batch, int H, int W, int Cin, int Cout) {
    ConvFFast f{};
    const long long in_px = batch * H * W, M = mode == 0 ? in_px / 4 : in_px;
    f.ok = (Cin & (Cin - 1)) == 0 && Cin >= (mode == 0 ? 8 : 16) && Cout % 32 == 0 && in_px < 0x7fffffffll && M >= 1 && (M + 127) / 128 * (Cout / 32) < 0x7fffffffll;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    f.off_x = 256; f.off_w = f.off_x + up((size_t)in_px * Cin * 2); f.bytes = f.off_w + up((size_t)16 * Cin * Cout * 2);
    return f;
}

extern "C" int vaek_conv2d_forward_workspace(int32_t batch, int32_t height, int32_t width, int32_t c_in, int32_t c_out, int32_t transposed, size_t* bytes) {
    if (!bytes || batch < 1 || height < 1 || width < 1 || c_in < 1 || c_out < 1) { set_error("vaek_conv2d_forward_workspace: invalid argument"); return VAEK_ERR_INVALID; }
    const ConvFFast f = conv_fwd_fast(transposed ? 1 : 0, batch, height, width, c_in, c_out);
    *bytes = f.ok ? f.bytes : 0;
    return VAEK_OK;
}
static int conv_forward_fast(int mode, const ConvFFast& f, const float* x, const float* w, const float* bias, const float* mask, const void* mask16,
                             float* out, void* workspace, const void* x16, void* out16, int batch, int H, int W, int Cin, int Cout, bool relu, hipStream_t st) {
    char* ws = static_cast<char*>(workspace);
    const __bf16* zeros = conv_zero_page();
    if (!zeros) { set_error("convolution: no page of zeros (hipGetSymbolAddress)"); return VAEK_ERR_HIP; }
    const __bf16* xb = static_cast<const __bf16*>(x16);
    __bf16* wb = reinterpret_cast<__bf16*>(ws + f.off_w);
    int rc = VAEK_OK;
    if (!xb) {                                            // no copy handed in: make one
        __bf16* mine = reinterpret_cast<__bf16*>(ws + f.off_x);
        rc = launch_cvt_bf16(x, mine, (int64_t)batch * H * W * Cin, nullptr, st);
        xb = mine;
    }
    if (rc == VAEK_OK) rc = mode == 0 ? launch_cvt_bf16_t(w, wb, 16 * Cin, Cout, nullptr, st) : launch_cvt_bf16(w, wb, (int64_t)16 * Cin * Cout, nullptr, st);
    if (rc == VAEK_OK) rc = launch_hs_conv(mode, xb, wb, zeros, bias, mask, static_cast<const __bf16*>(mask16), out, static_cast<__bf16*>(out16), batch, H, W, Cin, Cout, relu, st);
    return rc;
}

extern "C" int vaek_conv2d_weight_grad_workspace(int32_t batch, int32_t height, int32_t width, int32_t c_in, int32_t c_out, size_t* bytes) {
    if (!bytes || batch < 1 || height < 2 || width < 2 || c_in < 1 || c_out < 1) { set_error("vaek_conv2d_weight_grad_workspace: invalid argument"); return VAEK_ERR_INVALID; }
    const long long pixels = (long long)batch * (height / 2) * (width / 2);
    const int M = 16 * c_in + 1;
    const ConvWFast f = conv_wgrad_fast(batch, height, width, c_in, c_out);
    *bytes = f.ok ? f.bytes : (size_t)conv_wgrad_splits(pixels, M, c_out) * M * c_out * sizeof(float);
    if (c_in == 1 && thin_channels_ok(c_out)) *bytes = std::max(*bytes, (size_t)(kThinWgradBlocks + 1) * 17 * c_out * sizeof(float));
    return VAEK_OK;
}

extern "C" int vaek_conv2d_weight_grad(const float* x, const float* dy, float* dw, float* dbias, void* workspace, int32_t batch,
                                       int32_t height, int32_t width, int32_t c_in, int32_t c_out, const void* x_bf16, const void* dy_bf16,
                                       void* stream) {
    // (x / dy may be null beside their bf16 copies where the LDS-DMA form applies: that form reads nothing else)
    if ((!x && !x_bf16) || (!dy && !dy_bf16) || !dw || !workspace || batch < 1 || height < 2 || width < 2 || (height & 1) || (width & 1) ||
        c_in < 1 || c_out < 1 || !aligned16(x_bf16) || !aligned16(dy_bf16)) {
        set_error("vaek_conv2d_weight_grad: invalid argument");
        return VAEK_ERR_INVALID;
    }
    {
        const int hw_ = (height / 2) * (width / 2), wo_ = width / 2;
        const bool thin = c_in == 1 && thin_channels_ok(c_out);
        const bool thin_mfma = thin && c_out % 32 == 0 && (hw_ & (hw_ - 1)) == 0 && (wo_ & (wo_ - 1)) == 0;
        const bool ok = thin ? (x && (dy || thin_mfma)) : (conv_wgrad_fast(batch, height, width, c_in, c_out).ok || (x && dy));
        if (!ok) { set_error("vaek_conv2d_weight_grad: bf16-only tensors need the LDS-DMA (or, one channel: the matrix-core) kernel's shapes"); return VAEK_ERR_INVALID; }
    }
    ConvWArgs g{};
    g.x = x; g.dy = dy; g.slab = static_cast<float*>(workspace);
    g.B = batch; g.H = height; g.W = width; g.Cin = c_in; g.Cout = c_out; g.Ho = height / 2; g.Wo = width / 2;
    const long long pixels = (long long)batch * g.Ho * g.Wo;
    if (pixels > 0x7fffffffll) { set_error("vaek_conv2d_weight_grad: too many pixels"); return VAEK_ERR_INVALID; }
    hipStream_t st0 = (hipStream_t)stream;
    if (c_in == 1 && thin_channels_ok(c_out)) {           // one-channel gathered tensor: a streaming kernel, exact f32
        ThinArgs ta{};
        ta.x = x; ta.mask = dy; ta.mask16 = dy ? nullptr : static_cast<const __bf16*>(dy_bf16);       // (dy NULL: its bf16 copy, matrix-core form)
        ta.B = batch; ta.H = height; ta.W = width; ta.C = c_out; ta.Ho = g.Ho; ta.Wo = g.Wo; ta.pixels = pixels;
        float* part = static_cast<float*>(workspace);
        const int nb = (int)std::min<long long>(kThinWgradBlocks, (pixels + 255) / 256);
        {
            ProfScope ps("conv_wgrad_thin", st0);
            const int hw = g.Ho * g.Wo;
            if (c_out % 32 == 0 && (hw & (hw - 1)) == 0 && (g.Wo & (g.Wo - 1)) == 0)
                launch_k(ps, ta.mask16 ? thin_conv_wgrad_mfma_kernel<true> : thin_conv_wgrad_mfma_kernel<false>, dim3(nb, c_out / 32), dim3(256), 0, st0, ta, part, 31 - __builtin_clz(hw), 31 - __builtin_clz(g.Wo));
            else if (thin_groups_ok(c_out) && aligned16(dy)) launch_k(ps, thin_conv_wgrad8_kernel, dim3(nb), dim3(256), (size_t)4 * 17 * c_out * sizeof(float), st0, ta, part);
            else launch_k(ps, thin_conv_wgrad_kernel, dim3(nb), dim3(256), 0, st0, ta, part);
            VAEK_HIP_CHECK(hipGetLastError());
        }
        float* dwb = part + (size_t)kThinWgradBlocks * 17 * c_out;      // [17][c_out] behind the block partials: one grouped sum for kernel and bias
        int rc = launch_sum_slabs_inplace(part, (int64_t)17 * c_out, nb, dwb, (int64_t)17 * c_out, st0);
        if (rc == VAEK_OK) rc = launch_sum_slabs(dwb, 0, 1, dw, (int64_t)16 * c_out, st0);                  // (a "sum" of one slab: the copy)
        if (rc == VAEK_OK && dbias) rc = launch_sum_slabs(dwb + 16 * c_out, 0, 1, dbias, c_out, st0);
        return rc;
    }
    const ConvWFast f = conv_wgrad_fast(batch, height, width, c_in, c_out);
    if (f.ok) {
        char* ws = static_cast<char*>(workspace);
        const __bf16* zeros = conv_zero_page();
        if (!zeros) { set_error("convolution: no page of zeros (hipGetSymbolAddress)"); return VAEK_ERR_HIP; }
        const __bf16* xb = static_cast<const __bf16*>(x_bf16);
        const __bf16* dyb = static_cast<const __bf16*>(dy_bf16);
        float* slab = reinterpret_cast<float*>(ws + f.off_slab);
        const long long slab_stride = (long long)(16 * c_in + 1) * c_out;
        int rc = VAEK_OK;
        if (!xb) {
            __bf16* mine = reinterpret_cast<__bf16*>(ws + f.off_x);
            rc = launch_cvt_bf16(x, mine, (int64_t)batch * height * width * c_in, nullptr, st0);
            xb = mine;
        }
        if (rc == VAEK_OK && !dyb) {
            __bf16* mine = reinterpret_cast<__bf16*>(ws + f.off_dy);
            rc = launch_cvt_bf16(dy, mine, (int64_t)pixels * c_out, nullptr, st0);
            dyb = mine;
        }
        if (rc == VAEK_OK) rc = launch_hs_conv_dw(xb, dyb, zeros, slab, slab_stride, f.S, f.rps, batch, height, width, c_in, c_out, st0);
        const int64_t nk = (int64_t)16 * c_in * c_out;
        if (rc == VAEK_OK && dbias == dw + nk) return launch_sum_slabs(slab, slab_stride, f.S, dw, nk + c_out, st0);     // [kernel | bias] contiguous: one sum
        if (rc == VAEK_OK) rc = launch_sum_slabs(slab, slab_stride, f.S, dw, nk, st0);
        if (rc == VAEK_OK && dbias) rc = launch_sum_slabs(slab + nk, slab_stride, f.S, dbias, c_out, st0);
        return rc;
    }
    g.M = 16 * c_in + 1; g.N = c_out; g.K = (int)pixels;
    const int S = conv_wgrad_splits(pixels, g.M, g.N);
    g.k_per_split = (int)(((pixels + S - 1) / S + CBK - 1) / CBK * CBK);
    g.slab_stride = (long long)g.M * g.N;
    hipStream_t st = (hipStream_t)stream;
    {
        ProfScope ps("conv_wgrad_bf16", st);
        launch_k(ps, conv_wgrad_kernel, dim3((g.N + CBN - 1) / CBN, (g.M + CBM - 1) / CBM, S), dim3(CNT), 0, st, g);
        VAEK_HIP_CHECK(hipGetLastError());
    }
    int rc = launch_sum_slabs(g.slab, g.slab_stride, S, dw, (int64_t)16 * c_in * c_out, st);
    if (rc == VAEK_OK && dbias) rc = launch_sum_slabs(g.slab + (long long)16 * c_in * c_out, g.slab_stride, S, dbias, c_out, st);
    return rc;
}

extern "C" int vaek_conv2d_bias_grad(const float* dy, float* dbias, void* workspace, int64_t pixels, int32_t c, void* stream) {
    if (!dy || !dbias || !workspace || pixels < 1 || c < 1) { set_error("vaek_conv2d_bias_grad: invalid argument"); return VAEK_ERR_INVALID; }
    const int S = (int)std::min<long long>(512, (pixels + 255) / 256);
    const long long rps = (pixels + S - 1) / S;
    hipStream_t st = (hipStream_t)stream;
    const long long n = pixels * c;
    if ((c & (c - 1)) == 0 && c <= 1024 && n % 4 == 0 && aligned16(dy) && aligned16(workspace)) {
        const long long n4 = n / 4;
        const int S4 = (int)std::min<long long>(512, (n4 + 1023) / 1024);
        const long long chunk = ((n4 + S4 - 1) / S4 + 255) / 256 * 256;          // a multiple of 256 float4s: the column phase of a thread is fixed
        {
            ProfScope ps("conv_bias_grad", st);
            launch_k(ps, conv_colsum4_kernel, dim3(S4), dim3(256), 0, st, reinterpret_cast<const float4*>(dy), static_cast<float*>(workspace), n4, (int)c, chunk);
            VAEK_HIP_CHECK(hipGetLastError());
        }
        return launch_sum_slabs_inplace(static_cast<float*>(workspace), c, S4, dbias, c, st);
    }
    {
        ProfScope ps("conv_bias_grad", st);
        launch_k(ps, conv_colsum_kernel, dim3(S), dim3(256), 256 * sizeof(float), st, dy, static_cast<float*>(workspace), (long long)pixels, (int)c, rps);
        VAEK_HIP_CHECK(hipGetLastError());
    }
    return launch_sum_slabs(static_cast<const float*>(workspace), c, S, dbias, c, st);
}

extern "C" int vaek_conv2d_bias_grad_bf16(const void* dy_bf16, float* dbias, void* workspace, int64_t pixels, int32_t c, void* stream) {
    const long long n = pixels * (long long)c;
    if (!dy_bf16 || !dbias || !workspace || pixels < 1 || c < 8 || (c & (c - 1)) || c > 2048 || n % 8 || !aligned16(dy_bf16)) {
        set_error("vaek_conv2d_bias_grad_bf16: invalid argument (c a power of two in 8 .. 2048, 16-byte aligned)");
        return VAEK_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long n8 = n / 8;
    const int S8 = (int)std::min<long long>(512, (n8 + 1023) / 1024);
    const long long chunk = ((n8 + S8 - 1) / S8 + 255) / 256 * 256;              // a multiple of 256 chunks: the column phase of a thread is fixed
    {
        ProfScope ps("conv_bias_grad", st);
        launch_k(ps, conv_colsum8_bf16_kernel, dim3(S8), dim3(256), 0, st, static_cast<const uint4*>(dy_bf16), static_cast<float*>(workspace), n8, (int)c, chunk);
        VAEK_HIP_CHECK(hipGetLastError());
    }
    return launch_sum_slabs_inplace(static_cast<float*>(workspace), c, S8, dbias, c, st);
}

extern "C" int vaek_conv2d_transpose_forward(const float* y, const float* w, const float* bias, const float* mask, float* out,
                                             int32_t batch, int32_t height, int32_t width, int32_t c_in, int32_t c_out, int32_t relu,
                                             void* workspace, const void* y_bf16, void* out_bf16, void* stream) {
    // lean forms (see vaek.h): relu bit 1 = `mask` is the bf16 copy of the mask source; y / out may be null beside their bf16 copies
    const bool mask_b16 = (relu & 2) != 0 && mask;
    relu &= 1;
    const void* mask16 = mask_b16 ? static_cast<const void*>(mask) : nullptr;
    if (mask_b16) mask = nullptr;
    const bool lean = !y || !out || mask_b16;
    if ((!y && !y_bf16) || !w || (!out && !out_bf16) || batch < 1 || height < 1 || width < 1 || c_in < 1 || c_out < 1 || !aligned16(y_bf16) ||
        !aligned16(out_bf16) || !aligned16(mask16)) {
        set_error("vaek_conv2d_transpose_forward: invalid argument");
        return VAEK_ERR_INVALID;
    }
    const bool thin4s = c_out == 1 && c_in % 4 == 0 && c_in <= 256 && (c_in & (c_in - 1)) == 0 && width % 4 == 0 && aligned16(w) && aligned16(out) && aligned16(mask);
    if (lean && thin4s && out && !mask_b16) {             // the one-channel layer reading the bf16 copy of its input (the 4s kernel only)
        const long long M = (long long)batch * height * width;
        ThinArgs ta{};
        ta.x = y; ta.x16 = static_cast<const __bf16*>(y_bf16); ta.w = w; ta.bias = bias; ta.mask = mask; ta.y = out;
        ta.B = batch; ta.H = height; ta.W = width; ta.C = c_in; ta.pixels = 4 * M;
        {
            ProfScope ps("conv_t_fwd_thin", (hipStream_t)stream);
            launch_k(ps, ta.x16 ? thin_conv_t_fwd4s_kernel<true> : thin_conv_t_fwd4s_kernel<false>, dim3((unsigned)std::min<long long>(8192, (M / 4 * (c_in / 4) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ta, (int)relu);
            VAEK_HIP_CHECK(hipGetLastError());
        }
        return out_bf16 ? launch_cvt_bf16(out, static_cast<__bf16*>(out_bf16), 4 * M, nullptr, (hipStream_t)stream) : VAEK_OK;
    }
    if (lean) {
        const ConvFFast f = conv_fwd_fast(1, batch, height, width, c_in, c_out);
        if (!(f.ok && workspace && aligned16(workspace) && aligned16(y) && aligned16(w) && aligned16(out) && aligned16(bias) && c_out != 1)) {
            set_error("vaek_conv2d_transpose_forward: bf16-only tensors need the LDS-DMA kernel's shapes");
            return VAEK_ERR_INVALID;
        }
        return conv_forward_fast(1, f, y, w, bias, mask, mask16, out, workspace, y_bf16, out_bf16, batch, height, width, c_in, c_out, relu != 0, (hipStream_t)stream);
    }
    ConvTArgs g{};
    g.y = y; g.w = w; g.bias = bias; g.mask = mask; g.out = out;
    g.B = batch; g.h = height; g.w_in = width; g.Cin = c_in; g.Cout = c_out;
    const long long M = (long long)batch * height * width;
    if (M > 0x7fffffffll || (M + CBM - 1) / CBM > 65535) { set_error("vaek_conv2d_transpose_forward: too many pixels"); return VAEK_ERR_INVALID; }
    if (c_out == 1 && c_in % 4 == 0 && c_in <= 256 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        ThinArgs ta{};                                    // the last layer's shape: a streaming kernel, exact f32
        ta.x = y; ta.w = w; ta.bias = bias; ta.mask = mask; ta.y = out;
        ta.B = batch; ta.H = height; ta.W = width; ta.C = c_in; ta.pixels = 4 * M;
        {
            ProfScope ps("conv_t_fwd_thin", (hipStream_t)stream);
            if ((c_in & (c_in - 1)) == 0 && width % 4 == 0 && aligned16(out) && aligned16(mask))
                launch_k(ps, thin_conv_t_fwd4s_kernel<false>, dim3((unsigned)std::min<long long>(8192, (M / 4 * (c_in / 4) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ta, (int)relu);
            else if ((c_in & (c_in - 1)) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && (reinterpret_cast<uintptr_t>(mask) & 7) == 0)
                launch_k(ps, thin_conv_t_fwd4_kernel, dim3((unsigned)std::min<long long>(8192, (M * (c_in / 4) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ta, (int)relu);
            else
                launch_k(ps, thin_conv_t_fwd_kernel, dim3((unsigned)std::min<long long>(16384, (4 * M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ta, (int)relu);
            VAEK_HIP_CHECK(hipGetLastError());
        }
        return out_bf16 ? launch_cvt_bf16(out, static_cast<__bf16*>(out_bf16), 4 * M, nullptr, (hipStream_t)stream) : VAEK_OK;
    }
    if (workspace && aligned16(workspace) && aligned16(y) && aligned16(w) && aligned16(out) && aligned16(mask) && aligned16(bias)) {
        const ConvFFast f = conv_fwd_fast(1, batch, height, width, c_in, c_out);
        if (f.ok) return conv_forward_fast(1, f, y, w, bias, mask, nullptr, out, workspace, y_bf16, out_bf16, batch, height, width, c_in, c_out, relu != 0, (hipStream_t)stream);
    }
    g.M = (int)M; g.N = c_out; g.K = 4 * c_in; g.relu = relu;
    {
        ProfScope ps("conv_t_fwd_bf16", (hipStream_t)stream);
        launch_k(ps, conv_t_fwd_kernel, dim3((g.N + CBN - 1) / CBN, (g.M + CBM - 1) / CBM, 4), dim3(CNT), 0, (hipStream_t)stream, g);
        VAEK_HIP_CHECK(hipGetLastError());
    }
    return out_bf16 ? launch_cvt_bf16(out, static_cast<__bf16*>(out_bf16), 4 * M * c_out, nullptr, (hipStream_t)stream) : VAEK_OK;
}

extern "C" int vaek_to_bf16(const float* src, void* dst_bf16, int64_t n, void* stream) {
    if (!src || !dst_bf16 || n < 0 || !aligned16(src) || !aligned16(dst_bf16)) { set_error("vaek_to_bf16: invalid argument"); return VAEK_ERR_INVALID; }
    return launch_cvt_bf16(src, static_cast<__bf16*>(dst_bf16), n, nullptr, (hipStream_t)stream);
}

extern "C" int vaek_conv2d_forward(const float* x, const float* w, const float* bias, const float* mask, float* y, int32_t batch,
                                   int32_t height, int32_t width, int32_t c_in, int32_t c_out, int32_t relu, void* workspace,
                                   const void* x_bf16, void* y_bf16, void* stream) {
    // lean forms (see vaek.h): relu bit 1 = `mask` is the bf16 copy of the mask source; x / y may be null beside their bf16 copies
    const bool mask_b16 = (relu & 2) != 0 && mask;
    relu &= 1;
    const void* mask16 = mask_b16 ? static_cast<const void*>(mask) : nullptr;
    if (mask_b16) mask = nullptr;
    if ((!x && !x_bf16) || !w || (!y && !y_bf16) || batch < 1 || height < 2 || width < 2 || (height & 1) || (width & 1) || c_in < 1 || c_out < 1 ||
        !aligned16(x_bf16) || !aligned16(y_bf16) || !aligned16(mask16)) {
        set_error("vaek_conv2d_forward: invalid argument");
        return VAEK_ERR_INVALID;
    }
    const bool lean = !x || !y || mask_b16;
    if (lean && !(c_in == 1 && thin_channels_ok(c_out))) {
        const ConvFFast f = conv_fwd_fast(0, batch, height, width, c_in, c_out);
        if (!(f.ok && workspace && aligned16(workspace) && aligned16(x) && aligned16(w) && aligned16(y) && aligned16(bias))) {
            set_error("vaek_conv2d_forward: bf16-only tensors need the LDS-DMA kernel's shapes");
            return VAEK_ERR_INVALID;
        }
        return conv_forward_fast(0, f, x, w, bias, mask, mask16, y, workspace, x_bf16, y_bf16, batch, height, width, c_in, c_out, relu != 0, (hipStream_t)stream);
    }
    ConvArgs g{};
    g.x = x; g.w = w; g.bias = bias; g.mask = mask; g.y = y;
    g.B = batch; g.H = height; g.W = width; g.Cin = c_in; g.Cout = c_out; g.Ho = height / 2; g.Wo = width / 2;
    const long long M = (long long)batch * g.Ho * g.Wo;
    if (M > 0x7fffffffll || (M + CBM - 1) / CBM > 65535) { set_error("vaek_conv2d_forward: too many output pixels"); return VAEK_ERR_INVALID; }
    if (c_in == 1 && thin_channels_ok(c_out)) {           // the first layer's shape: a streaming kernel, exact f32
        ThinArgs ta{};
        ta.x = x; ta.w = w; ta.bias = bias; ta.mask = mask; ta.y = y;
        ta.B = batch; ta.H = height; ta.W = width; ta.C = c_out; ta.Ho = g.Ho; ta.Wo = g.Wo; ta.pixels = M;
        const bool by8 = thin_groups_ok(c_out) && aligned16(w) && aligned16(y) && aligned16(mask);
        {
            ProfScope ps("conv_fwd_thin", (hipStream_t)stream);
            const int hw = g.Ho * g.Wo;
            const bool mfma_form = by8 && c_out % 32 == 0 && (hw & (hw - 1)) == 0 && (g.Wo & (g.Wo - 1)) == 0;
            if (lean && (!x || !mfma_form)) {                   // (the one-channel layer: x stays float32; the lean forms on the matrix-core form only)
                set_error("vaek_conv2d_forward: the one-channel layer reads a float32 x, and its lean forms need the matrix-core kernel's shapes");
                return VAEK_ERR_INVALID;
            }
            if (mfma_form) {
                ta.y16 = static_cast<__bf16*>(y_bf16);
                ta.mask16 = static_cast<const __bf16*>(mask16);
                launch_k(ps, ta.mask16 ? thin_conv_fwd_mfma_kernel<true> : thin_conv_fwd_mfma_kernel<false>, dim3((unsigned)std::min<long long>(8192, (M / 32 + 3) / 4 + 1), c_out / 32), dim3(256), 0, (hipStream_t)stream,
                         ta, (int)relu, 31 - __builtin_clz(hw), 31 - __builtin_clz(g.Wo));
            } else if (by8) {
                ta.y16 = static_cast<__bf16*>(y_bf16);
                launch_k(ps, thin_conv_fwd8_kernel, dim3((unsigned)std::min<long long>(8192, (M * (c_out / 8) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ta, (int)relu);
            } else {
                launch_k(ps, thin_conv_fwd_kernel, dim3((unsigned)std::min<long long>(8192, (M * c_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ta, (int)relu);
            }
            VAEK_HIP_CHECK(hipGetLastError());
        }
        return y_bf16 && !by8 ? launch_cvt_bf16(y, static_cast<__bf16*>(y_bf16), M * c_out, nullptr, (hipStream_t)stream) : VAEK_OK;
    }
    if (workspace && aligned16(workspace) && aligned16(x) && aligned16(w) && aligned16(y) && aligned16(mask) && aligned16(bias)) {
        const ConvFFast f = conv_fwd_fast(0, batch, height, width, c_in, c_out);
        if (f.ok) return conv_forward_fast(0, f, x, w, bias, mask, nullptr, y, workspace, x_bf16, y_bf16, batch, height, width, c_in, c_out, relu != 0, (hipStream_t)stream);
    }
    g.M = (int)M; g.N = c_out; g.K = 16 * c_in; g.relu = relu;
    {
        ProfScope ps("conv_fwd_bf16", (hipStream_t)stream);
        launch_k(ps, conv_fwd_kernel, dim3((g.N + CBN - 1) / CBN, (g.M + CBM - 1) / CBM), dim3(CNT), 0, (hipStream_t)stream, g);
        VAEK_HIP_CHECK(hipGetLastError());
    }
    return y_bf16 ? launch_cvt_bf16(y, static_cast<__bf16*>(y_bf16), M * c_out, nullptr, (hipStream_t)stream) : VAEK_OK;
}
