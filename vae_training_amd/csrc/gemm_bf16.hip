// Dense kernels with bf16 matrix-core arithmetic (BASELINE configs C2/C3: "MLP ... bf16 ... MFMA Dense
// path"): v_mfma_f32_32x32x16_bf16, f32 accumulate, f32 storage.  Same three uses and the same epilogues
// as gemm_f32.hip (forward act(XW+b) / reparameterisation epilogue, dX = (dY W^T) * relu', dW|db =
// [X|1]^T dY split over the batch), selected with vaek_config.dtype = VAEK_BF16.  Operands are rounded
// f32 -> bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are staged into LDS, so HBM keeps
// the reference's float32 tensors and only the products lose precision (measured deviation from the
// float64 oracle: tests/test_gpu_bf16.py; the 1e-5 ELBO contract stays with the f32 path).
//
// Block = 256 threads = 4 waves (2x2), 128x128 output tile, each wave 64x64 = 2x2 MFMA tiles of 32x32,
// BK = 32 or 64 (16-deep MFMA steps).  LDS tiles are [row][k] with k contiguous and an 80- or 144-byte row stride
// (BK bf16 + 8 pad): the 16-byte fragment reads (lane = row, 8 consecutive k) are conflict-free, and an
// operand whose k runs contiguously in memory is staged with 8-byte writes.  The next K-tile is fetched
// into registers while the MFMAs of the current one run.
#include "vaek_internal.h"

namespace vaek {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

constexpr int HBM_ = 128, HBN_ = 128, HNT = 256;
// The k-tile depth HBK is a template parameter; every use runs 32.  64 was measured: the batch-split dW|db looked
// faster in isolation (909 -> 678 us under per-kernel timestamps) but the C3 step replayed from a hipGraph went from
// 2.73 to 2.94 ms with it, and dX got 14 % slower even in isolation -- at K = layer width a workgroup has 8-16 k-tiles
// and occupancy (VGPRs) matters more than barriers.  LDS rows are HBK + 8 bf16 (80 bytes at 32; 144 at 64: both
// conflict-free for the 16-byte fragment reads); HKU = float4 units per thread and operand tile.

enum { HEPI_FWD = 0, HEPI_REPARAM = 1, HEPI_DX = 2, HEPI_DW = 3 };

struct HGemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    int lda, ldb, ldc;
    int a_mem;
    const float* bias; int relu;
    const float* aux; float* C2; const float* lv;
    int accumulate;
    int k_per_split; long long slab_stride;
};

// tile = 128 (mn) x HBK (k).  k-contiguous source p[mn*ld + k]: thread -> rows (t / (HBK/4)) + (1024/HBK) u, 4 consecutive k.
template <int HBK, int HKU = HBK / 8>
__device__ __forceinline__ void hfetch_kcont(const float* __restrict__ p, int ld, int mn0, int MN, int k0, int kend,
                                             bool vec_ok, float (&v)[HKU][4]) {
    constexpr int TPR = HBK / 4, RPP = HNT / TPR;       // threads per row, rows per pass
    const int t = threadIdx.x;
    const int k = k0 + (t % TPR) * 4;
#pragma unroll
    for (int u = 0; u < HKU; ++u) {
        const int mn = mn0 + t / TPR + RPP * u;
        v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0.f;
        if (mn < MN) {
            const float* q = p + (long long)mn * ld + k;
            if (vec_ok && k + 3 < kend) {
                const float4 f = *reinterpret_cast<const float4*>(q);
                v[u][0] = f.x; v[u][1] = f.y; v[u][2] = f.z; v[u][3] = f.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (k + c < kend) v[u][c] = q[c];
            }
        }
    }
}
template <int HBK, int HKU = HBK / 8, int HSTR = HBK + 8>
__device__ __forceinline__ void hstore_kcont(__bf16* s, const float (&v)[HKU][4]) {
    constexpr int TPR = HBK / 4, RPP = HNT / TPR;
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < HKU; ++u) {
        bf16x4 h = {(__bf16)v[u][0], (__bf16)v[u][1], (__bf16)v[u][2], (__bf16)v[u][3]};
        *reinterpret_cast<bf16x4*>(&s[(t / TPR + RPP * u) * HSTR + (t % TPR) * 4]) = h;
    }
}
// mn-contiguous source p[k*ld + mn] (the operand must be TRANSPOSED into the [row][k] image): thread ->
// ONE row mn = t & 127 and HBK/2 consecutive k = (HBK/2)(t >> 7) ..: the dword loads are coalesced along mn
// (256 B per wave and k), and the thread then owns HBK contiguous bytes of its LDS row -> 16-byte
// writes at the conflict-free row stride (a 2-byte scatter would be a 16-way bank conflict).
template <int HBK, int HKU = HBK / 8>
__device__ __forceinline__ void hfetch_mncont(const float* __restrict__ p, int ld, int mn0, int mem, bool aug, int k0,
                                              int kend, bool /*vec_ok*/, float (&v)[HKU][4]) {
    const int t = threadIdx.x;
    const int mn = mn0 + (t & 127);
    const int kb = k0 + (HBK / 2) * (t >> 7);
    const float* q = p + (long long)kb * ld + mn;
#pragma unroll
    for (int u = 0; u < HKU; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = kb + 4 * u + c;
            float x = 0.f;
            if (k < kend) {
                if (mn < mem) x = q[(long long)(4 * u + c) * ld];
                else if (aug && mn == mem) x = 1.f;
            }
            v[u][c] = x;
        }
}
template <int HBK, int HKU = HBK / 8, int HSTR = HBK + 8>
__device__ __forceinline__ void hstore_mncont(__bf16* s, const float (&v)[HKU][4]) {
    const int t = threadIdx.x;
    __bf16* row = s + (t & 127) * HSTR + (HBK / 2) * (t >> 7);
#pragma unroll
    for (int h = 0; h < HKU / 2; ++h) {
        bf16x8 w = {(__bf16)v[2 * h][0], (__bf16)v[2 * h][1], (__bf16)v[2 * h][2], (__bf16)v[2 * h][3],
                    (__bf16)v[2 * h + 1][0], (__bf16)v[2 * h + 1][1], (__bf16)v[2 * h + 1][2], (__bf16)v[2 * h + 1][3]};
        *reinterpret_cast<bf16x8*>(row + 8 * h) = w;
    }
}

// Full-tile forms (every load unconditional): see gemm_f32.hip -- a load under `if (mn < MN)` is waited for where it is issued.
template <int HBK, int HKU = HBK / 8>
__device__ __forceinline__ void hfetch_kcont_full(const float* __restrict__ p, int ld, int mn0, int k0, float (&v)[HKU][4]) {
    constexpr int TPR = HBK / 4, RPP = HNT / TPR;
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < HKU; ++u) {
        const float4 f = *reinterpret_cast<const float4*>(p + (long long)(mn0 + t / TPR + RPP * u) * ld + k0 + (t % TPR) * 4);
        v[u][0] = f.x; v[u][1] = f.y; v[u][2] = f.z; v[u][3] = f.w;
    }
}
template <int HBK, int HKU = HBK / 8>
__device__ __forceinline__ void hfetch_mncont_full(const float* __restrict__ p, int ld, int mn0, int k0, float (&v)[HKU][4]) {
    const int t = threadIdx.x;
    const float* q = p + (long long)(k0 + (HBK / 2) * (t >> 7)) * ld + mn0 + (t & 127);
#pragma unroll
    for (int u = 0; u < HKU; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[u][c] = q[(long long)(4 * u + c) * ld];
}

template <bool A_KCONT, bool B_KCONT, int EPI, int HBK>
__global__ __launch_bounds__(HNT) void gemm_bf16_kernel(const HGemmArgs g) {
    constexpr int HSTR = HBK + 8, HKU = HBK / 8;
    __shared__ __attribute__((aligned(16))) __bf16 As[HBM_ * HSTR];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[HBN_ * HSTR];
    // XCD-aware tile order (see gemm_f32.hip): each XCD walks a contiguous run of the x-fastest tile space
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, nb = gx * gy * gridDim.z;
        if (nb % 8 == 0) {
            const unsigned id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
            const unsigned sw = (id % 8) * (nb / 8) + id / 8;
            bx = sw % gx; by = (sw / gx) % gy; bz = sw / (gx * gy);
        }
    }
    const int m0 = by * HBM_, n0 = bx * HBN_;
    int kbeg = 0, kend = g.K;
    if (EPI == HEPI_DW) {
        kbeg = bz * g.k_per_split;
        kend = min(g.K, kbeg + g.k_per_split);
    }
    const bool a_vec = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) && (A_KCONT ? (kbeg % 4 == 0) : true);
    const bool b_vec = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0) && (B_KCONT ? (kbeg % 4 == 0) : true);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[HKU][4], rb[HKU][4];
    const bool a_in = A_KCONT ? (a_vec && m0 + HBM_ <= g.M) : m0 + HBM_ <= g.a_mem;
    const bool b_in = B_KCONT ? (b_vec && n0 + HBN_ <= g.N) : n0 + HBN_ <= g.N;
    auto fetch = [&](int k0) {
        const bool k_in = k0 + HBK <= kend;                // uniform: a scalar branch
        if (a_in && k_in) { if (A_KCONT) hfetch_kcont_full<HBK>(g.A, g.lda, m0, k0, ra); else hfetch_mncont_full<HBK>(g.A, g.lda, m0, k0, ra); }
        else if (A_KCONT) hfetch_kcont<HBK>(g.A, g.lda, m0, g.M, k0, kend, a_vec, ra);
        else hfetch_mncont<HBK>(g.A, g.lda, m0, g.a_mem, EPI == HEPI_DW, k0, kend, a_vec, ra);
        if (b_in && k_in) { if (B_KCONT) hfetch_kcont_full<HBK>(g.B, g.ldb, n0, k0, rb); else hfetch_mncont_full<HBK>(g.B, g.ldb, n0, k0, rb); }
        else if (B_KCONT) hfetch_kcont<HBK>(g.B, g.ldb, n0, g.N, k0, kend, b_vec, rb);
        else hfetch_mncont<HBK>(g.B, g.ldb, n0, g.N, false, k0, kend, b_vec, rb);
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += HBK) {
        __syncthreads();
        if (A_KCONT) hstore_kcont<HBK>(As, ra); else hstore_mncont<HBK>(As, ra);
        if (B_KCONT) hstore_kcont<HBK>(Bs, rb); else hstore_mncont<HBK>(Bs, rb);
        __syncthreads();
        if (k0 + HBK < kend) fetch(k0 + HBK);
        // fragments: lane (row = lane&31, half = lane>>5) holds k = 16s + 8*half .. +7 of its row
        const __bf16* pa = As + (wm * 64 + (lane & 31)) * HSTR + 8 * (lane >> 5);
        const __bf16* pb = Bs + (wn * 64 + (lane & 31)) * HSTR + 8 * (lane >> 5);
#pragma unroll
        for (int s = 0; s < HBK / 16; ++s) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * HSTR + 16 * s);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * HSTR + 16 * s);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    float* C = g.C;
    if (EPI == HEPI_DW) C += (long long)bz * g.slab_stride;
    // epilogue inputs (relu mask source / z1) into registers before the first store: see gemm_f32.hip
    constexpr bool kAux = EPI == HEPI_REPARAM || EPI == HEPI_DX;
    float auxv[kAux ? 2 : 1][kAux ? 2 : 1][16];
    if (kAux && (EPI != HEPI_DX || g.relu)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const bool in = col < g.N && row < g.M;
                    auxv[kAux ? i : 0][kAux ? j : 0][r] = g.aux[in ? (long long)row * g.ldc + col : 0];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + (lane & 31);
        if (col >= g.N) continue;
        float bias = 0.f, sdev = 0.f;
        if (EPI == HEPI_FWD || EPI == HEPI_REPARAM) bias = g.bias ? g.bias[col] : 0.f;
        if (EPI == HEPI_REPARAM) sdev = expf(0.5f * g.lv[col]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= g.M) continue;
                const long long o = (long long)row * g.ldc + col;
                float v = acc[i][j][r];
                const float ax = auxv[kAux ? i : 0][kAux ? j : 0][r];
                if (EPI == HEPI_FWD) {
                    v += bias;
                    if (g.relu) v = fmaxf(v, 0.f);
                    C[o] = v;
                } else if (EPI == HEPI_REPARAM) {
                    v += bias;
                    C[o] = v;
                    g.C2[o] = v + sdev * ax;
                } else if (EPI == HEPI_DX) {
                    if (g.relu) v = ax > 0.f ? v : 0.f;
                    if (g.accumulate) v += C[o];
                    C[o] = v;
                } else {
                    C[o] = v;
                }
            }
    }
}

template <bool A_KCONT, bool B_KCONT, int EPI>
static int hlaunch(const HGemmArgs& g, int splits, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return VAEK_OK;
    ProfScope ps(EPI == HEPI_FWD ? "gemm_bf16_fwd" : EPI == HEPI_REPARAM ? "gemm_bf16_fwd_reparam"
                 : EPI == HEPI_DX ? "gemm_bf16_dx" : "gemm_bf16_dw", st);
    dim3 grid((g.N + HBN_ - 1) / HBN_, (g.M + HBM_ - 1) / HBM_, splits);
    if (grid.y > 65535u || grid.z > 65535u) { set_error("gemm grid too large (M=%d N=%d splits=%d)", g.M, g.N, splits); return VAEK_ERR_INVALID; }
    launch_k(ps, (gemm_bf16_kernel<A_KCONT, B_KCONT, EPI, 32>), grid, dim3(HNT), 0, st, g);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int launch_dense_fwd_bf16(const float* x, const float* w, const float* b, float* y, int rows, int n_in, int n_out,
                          bool relu, hipStream_t st) {
    HGemmArgs g{};
    g.A = x; g.B = w; g.C = y; g.M = rows; g.N = n_out; g.K = n_in;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.bias = b; g.relu = relu;
    return hlaunch<true, false, HEPI_FWD>(g, 1, st);
}
int launch_dense_fwd_reparam_bf16(const float* x, const float* w, const float* b, float* mu, float* samples,
                                  const float* z1, const float* lv, int rows, int n_in, int n_out, hipStream_t st) {
    HGemmArgs g{};
    g.A = x; g.B = w; g.C = mu; g.C2 = samples; g.aux = z1; g.lv = lv;
    g.M = rows; g.N = n_out; g.K = n_in; g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.bias = b;
    return hlaunch<true, false, HEPI_REPARAM>(g, 1, st);
}
int launch_dense_bwd_dx_bf16(const float* dy, const float* w, const float* x_post, float* dx, int rows, int n_in,
                             int n_out, bool relu, bool accumulate, hipStream_t st) {
    HGemmArgs g{};
    g.A = dy; g.B = w; g.C = dx; g.M = rows; g.N = n_in; g.K = n_out;
    g.lda = n_out; g.ldb = n_out; g.ldc = n_in; g.aux = x_post; g.relu = relu && x_post != nullptr; g.accumulate = accumulate;
    return hlaunch<true, true, HEPI_DX>(g, 1, st);
}
int launch_dense_bwd_dw_bf16(const float* x, const float* dy, float* slab0, int64_t slab_stride, int S, int rows_per_split,
                             int rows, int n_in, int n_out, hipStream_t st) {
    HGemmArgs g{};
    g.A = x; g.B = dy; g.C = slab0; g.M = n_in + 1; g.N = n_out; g.K = rows;
    g.lda = n_in; g.ldb = n_out; g.ldc = n_out; g.a_mem = n_in; g.k_per_split = rows_per_split; g.slab_stride = slab_stride;
    return hlaunch<false, false, HEPI_DW>(g, S, st);
}

}  // namespace vaek
