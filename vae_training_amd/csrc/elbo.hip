// Elementwise / reduction kernels of the ELBO train step (K4-K6 of SURVEY.md 7.1), float32.
//   elbo_kernel         networks.py:81-83 (decoder noise) + :94-98 (Dkl, mse, mean) and dL/dx_hat
//   reparam_bwd_kernel  backward of networks.py:73-74 (samples = mu + exp(lv/2) z1) + KL's mu term
//   finalize_kernel     fixed-order sum of the per-split partial slabs -> flat gradient (+ fused Adam)
//   adam_kernel         flax.optim.Adam.apply_gradient, networks.py:100
// All batch reductions are two-stage (per-split partials, then a fixed-order sum): no float
// atomics, so results are bitwise repeatable run to run.
#include "vaek_internal.h"

namespace vaek {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over a 256-thread block; result valid in thread 0.  red: >= 4 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += red[i];
    }
    return r;
}
__device__ __forceinline__ float sigmoidf_(float a) { return 1.f / (1.f + expf(-a)); }

template <bool SIG, bool GRADS>
__global__ __launch_bounds__(256) void elbo_kernel(const ElboArgs a) {
    __shared__ float red[8];
    const int s = blockIdx.x;
    const long long r0 = (long long)s * a.rows_per_split;
    const long long r1 = min((long long)a.rows, r0 + a.rows_per_split);
    const float eps = a.eps_param ? a.eps_param[0] * a.eps_cli : a.eps_cli;
    const float inv_var = expf(-eps), sigma = expf(0.5f * eps);
    const float dscale = inv_var * a.inv_bt;
    float mse = 0.f, deps = 0.f, musq = 0.f;
    if (r0 < r1) {
        const long long e0 = r0 * a.D, e1 = r1 * a.D;
        auto elem = [&](long long e, float xv, float yl, float ys, float z, float& d_l, float& d_s) {
            float xh = yl + sigma * z;
            float sg = 0.f;
            if (SIG) { sg = sigmoidf_(ys); xh += sg; }
            const float r = xh - xv;
            const float q = r * r * inv_var;
            mse += 0.5f * q;
            deps += -0.5f * q + 0.5f * sigma * z * r * inv_var;
            d_l = r * dscale;
            d_s = SIG ? d_l * sg * (1.f - sg) : 0.f;
        };
        // 16-byte path: a split starts at a multiple of 64 rows, so e0 is 16-byte aligned whenever the base
        // pointers are; the (< 4)-element tail of the last split goes through the scalar loop
        const bool vec = (e0 % 4 == 0) && (((uintptr_t)a.x | (uintptr_t)a.y_lin | (uintptr_t)a.z2 | (uintptr_t)a.y_sig |
                                             (uintptr_t)a.d_lin | (uintptr_t)a.d_sig) & 15) == 0;
        long long ev = e0;
        if (vec) {
            ev = e0 + ((e1 - e0) & ~3ll);
            for (long long e = e0 + 4ll * threadIdx.x; e < ev; e += 4ll * blockDim.x) {
                const float4 xv = *reinterpret_cast<const float4*>(a.x + e), yl = *reinterpret_cast<const float4*>(a.y_lin + e);
                const float4 z = *reinterpret_cast<const float4*>(a.z2 + e);
                float4 ys = make_float4(0.f, 0.f, 0.f, 0.f);
                if (SIG) ys = *reinterpret_cast<const float4*>(a.y_sig + e);
                float4 dl, ds;
                elem(e, xv.x, yl.x, ys.x, z.x, dl.x, ds.x); elem(e + 1, xv.y, yl.y, ys.y, z.y, dl.y, ds.y);
                elem(e + 2, xv.z, yl.z, ys.z, z.z, dl.z, ds.z); elem(e + 3, xv.w, yl.w, ys.w, z.w, dl.w, ds.w);
                if (GRADS) {
                    *reinterpret_cast<float4*>(a.d_lin + e) = dl;
                    if (SIG) *reinterpret_cast<float4*>(a.d_sig + e) = ds;
                }
            }
        }
        for (long long e = ev + threadIdx.x; e < e1; e += blockDim.x) {
            float dl, ds;
            elem(e, a.x[e], a.y_lin[e], SIG ? a.y_sig[e] : 0.f, a.z2[e], dl, ds);
            if (GRADS) {
                a.d_lin[e] = dl;
                if (SIG) a.d_sig[e] = ds;
            }
        }
        const long long l0 = r0 * a.L, l1 = r1 * a.L;
        for (long long e = l0 + threadIdx.x; e < l1; e += blockDim.x) {
            const float m = a.mu[e];
            musq += m * m;
        }
    }
    const float t_mse = block_sum(mse, red);
    const float t_deps = block_sum(deps, red);
    const float t_musq = block_sum(musq, red);
    if (threadIdx.x == 0) {
        float* p = a.partial + (long long)s * 4;
        p[0] = t_mse; p[1] = t_musq; p[2] = t_deps; p[3] = 0.f;
        if (s == 0 && a.step_dev) a.step_dev[0] += 1;
    }
}

// Second half of the ELBO when its elementwise pass ran in the decoder's last GEMM (gemm_f32.hip, EPI_ELBO): split s sums
// the {mse, d eps} pairs of the output tiles whose rows it owns (tile rows of `bm` rows, `nbx` tiles each; a split starts
// on a tile row) and the mu^2 of its rows, and writes the same partial[s][4] elbo_kernel would have.  Fixed order.
__global__ __launch_bounds__(256) void elbo_reduce_kernel(const float* part, int bm, int nbx, const float* mu, float* partial,
                                                         int rows, int L, int rows_per_split, int32_t* step_dev) {
    __shared__ float red[8];
    const int s = blockIdx.x;
    const long long r0 = (long long)s * rows_per_split, r1 = min((long long)rows, r0 + rows_per_split);
    float mse = 0.f, deps = 0.f, musq = 0.f;
    if (r0 < r1) {
        const long long t0 = r0 / bm * nbx, t1 = (r1 + bm - 1) / bm * nbx;
        for (long long t = t0 + threadIdx.x; t < t1; t += blockDim.x) { mse += part[2 * t]; deps += part[2 * t + 1]; }
        for (long long e = r0 * L + threadIdx.x; e < r1 * L; e += blockDim.x) { const float m = mu[e]; musq += m * m; }
    }
    const float t_mse = block_sum(mse, red), t_deps = block_sum(deps, red), t_musq = block_sum(musq, red);
    if (threadIdx.x == 0) {
        float* p = partial + (long long)s * 4;
        p[0] = t_mse; p[1] = t_musq; p[2] = t_deps; p[3] = 0.f;
        if (s == 0 && step_dev) step_dev[0] += 1;
    }
}

int launch_elbo_reduce(const float* part, int bm, int nbx, const float* mu, float* partial, int rows, int L, int S,
                       int rows_per_split, int32_t* step_dev, hipStream_t st) {
    if (rows_per_split % bm != 0) { set_error("elbo reduce: split of %d rows does not start on %d-row tiles", rows_per_split, bm); return VAEK_ERR_INVALID; }
    ProfScope ps("elbo_reduce", st);
    launch_k(ps, elbo_reduce_kernel, dim3(S), dim3(256), 0, st, part, bm, nbx, mu, partial, rows, L, rows_per_split, step_dev);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int launch_elbo(const ElboArgs& a, hipStream_t st) {
    const bool sig = a.y_sig != nullptr, grads = a.d_lin != nullptr;
    ProfScope ps("elbo", st);
    dim3 grid(a.S), block(256);
    if (sig && grads) launch_k(ps, (elbo_kernel<true, true>), grid, block, 0, st, a);
    else if (sig) launch_k(ps, (elbo_kernel<true, false>), grid, block, 0, st, a);
    else if (grads) launch_k(ps, (elbo_kernel<false, true>), grid, block, 0, st, a);
    else launch_k(ps, (elbo_kernel<false, false>), grid, block, 0, st, a);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

// block s: rows [r0, r1); thread t -> column t % L, row group t / L (threads >= G*L idle)
__global__ __launch_bounds__(256) void reparam_bwd_kernel(float* dsamp, const float* mu, const float* z1,
                                                         float* partial, int rows, int L,
                                                         int rows_per_split, float inv_bt) {
    extern __shared__ float sh[];   // 256 floats
    const int s = blockIdx.x;
    const int r0 = s * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    const int G = 256 / L;          // L <= 256 checked on the host
    const int col = threadIdx.x % L, grp = threadIdx.x / L;
    float acc = 0.f;
    if (grp < G) {
        for (int r = r0 + grp; r < r1; r += G) {
            const long long o = (long long)r * L + col;
            const float d = dsamp[o];
            acc += d * z1[o];
            dsamp[o] = d + mu[o] * inv_bt;
        }
    }
    sh[threadIdx.x] = grp < G ? acc : 0.f;
    __syncthreads();
    if (threadIdx.x < L) {
        float t = 0.f;
        for (int g2 = 0; g2 < G; ++g2) t += sh[g2 * L + threadIdx.x];
        partial[(long long)s * L + threadIdx.x] = t;
    }
}

int launch_reparam_bwd(float* dsamp, const float* mu, const float* z1, float* partial,
                       int rows, int L, int S, int rows_per_split, float inv_bt, hipStream_t st) {
    if (L > 256) { set_error("latent_dim %d > 256 not supported by reparam_bwd", L); return VAEK_ERR_INVALID; }
    ProfScope ps("reparam_bwd", st);
    launch_k(ps, reparam_bwd_kernel, dim3(S), dim3(256), 256 * sizeof(float), st, dsamp, mu, z1, partial,
                       rows, L, rows_per_split, inv_bt);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

// Bias corrections 1 - beta^t, accurate to ~2e-7 relative for every t >= 1 in float32.
__device__ __forceinline__ void adam_apply(float& p, float g, float& m, float& v, float lr, float bc1, float bc2) {
    // (1 - beta) is formed in double first, as Python does for flax's float hyper-parameters:
    // 1.f - 0.999f is off by 4.7e-5 relative
    m = kAdamB1 * m + (float)(1.0 - 0.9) * g;
    v = kAdamB2 * v + (float)(1.0 - 0.999) * g * g;
    const float mh = m / bc1, vh = v / bc2;
    p = p - lr * mh / (sqrtf(vh) + kAdamEps);
}
__device__ __forceinline__ void adam_bias_corrections(int t, float& bc1, float& bc2) {
    // log(0.9), log(0.999) in float; -expm1(t log b) keeps full relative precision at small t
    bc1 = -expm1f((float)t * -0.10536051565782628f);
    bc2 = -expm1f((float)t * -0.0010005003335835335f);
}

// Block b covers outputs [n - 64(b+1), n - 64b): the scalar sums (and epsilon, which needs one of them)
// always sit in block 0.  1024 threads = 64 outputs x 16 row groups: every thread walks its share of
// the S dW|db slabs / Se elementwise partial rows with independent loads, then a fixed-order sum.
constexpr int FINQ = 16;
// sum of p[s * stride] over s = q, q + FINQ, ... < n, in that order -- but with 8 loads in flight: a model with 1024
// elementwise splits made this a chain of 64 dependent ~1 us loads in the one workgroup that owns the loss (60 us)
__device__ __forceinline__ float strided_sum(const float* p, long long stride, int q, int n) {
    float acc = 0.f;
    for (int s0 = q; s0 < n; s0 += FINQ * 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int s = s0 + u * FINQ; t[u] = s < n ? p[(long long)s * stride] : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
    }
    return acc;
}
__global__ __launch_bounds__(1024) void finalize_kernel(const FinalizeArgs a) {
    __shared__ float part[FINQ][64];
    __shared__ float sums[64];
    __shared__ float klred[1024];
    const int t = threadIdx.x, o = t & 63, q = t >> 6;
    const long long n = a.P + kExtra;
    const long long i = n - 64ll * ((long long)blockIdx.x + 1) + o;
    // Adam state of this output is fetched up front, under the partial loads
    const bool adam = a.params_rw != nullptr && q == 0 && i >= 0 && i < a.P;
    float p_old = 0.f, m_old = 0.f, v_old = 0.f;
    int tstep = 0;
    if (adam) { p_old = a.params_rw[i]; m_old = a.m[i]; v_old = a.v[i]; tstep = a.step_dev[0]; }
    float acc = 0.f;
    if (i >= 0 && i < a.P) {
        if (i == a.off_eps) {
            acc = strided_sum(a.epart + 2, 4, q, a.Se);
        } else if (i >= a.off_epsp && i < a.off_epsp + a.L) {
            acc = strided_sum(a.rpart + (i - a.off_epsp), a.L, q, a.Se);
        } else {
            int S = a.S;                               // slabs written for this output's layer
            for (int k = a.nseg - 1; k >= 0; --k) if (i < a.seg_end[k]) S = a.seg_S[k];
            acc = strided_sum(a.slabs + i, a.slab_stride, q, S);
        }
    } else if (i == a.P || i == a.P + 1) {
        acc = strided_sum(a.epart + (int)(i - a.P), 4, q, a.Se);                           // sum mse terms, sum mu^2
    }
    part[q][o] = acc;
    float kl = 0.f;                                   // closed-form KL constant sum_l (1 + lv - e^lv), block 0 only
    if (blockIdx.x == 0) for (int l = t; l < a.L; l += 1024) { const float lv = a.params[a.off_epsp + l]; kl += 1.f + lv - expf(lv); }
    klred[t] = kl;
    const float eps_par = a.off_eps >= 0 ? a.params[a.off_eps] : 0.f;
    const float lv_own = (q == 0 && i >= a.off_epsp && i < a.off_epsp + a.L) ? a.params[i] : 0.f;
    __syncthreads();
    if (q == 0) {
        float sacc = 0.f;
#pragma unroll
        for (int u = 0; u < FINQ; ++u) sacc += part[u][o];
        sums[o] = sacc;
    }
    if (blockIdx.x == 0) {
        for (int w = 512; w > 0; w >>= 1) { if (t < w) klred[t] += klred[t + w]; __syncthreads(); }
    } else {
        __syncthreads();
    }
    const bool live = q == 0 && i >= 0 && i >= a.lo;
    float g = live ? sums[o] : 0.f;
    const long long base = n - 64ll * ((long long)blockIdx.x + 1);
    if (!live) {
    } else if (i == a.off_eps) {
        // eps = param * eps_cli, networks.py:71; + 0.5 per (row, d) element is the constant part
        g = a.eps_cli * (g + 0.5f * a.rows * (float)a.D) * a.inv_bt;
    } else if (i >= a.off_epsp && i < a.off_epsp + a.L) {
        g = 0.5f * expf(0.5f * lv_own) * g - 0.5f * (1.f - expf(lv_own)) * a.rows_over_bt;
    } else if (i >= a.P) {
        if (i < a.P + 3) {
            const float eps = a.off_eps >= 0 ? eps_par * a.eps_cli : a.eps_cli;
            const float dkl = (0.5f * sums[a.P + 1 - base] - 0.5f * a.rows * klred[0]) * a.inv_bt;
            const float mse = (sums[a.P - base] + 0.5f * a.rows * (float)a.D * (kLog2Pi + eps)) * a.inv_bt;
            g = (i == a.P) ? dkl + mse : (i == a.P + 1 ? dkl : mse);
        } else {
            g = 0.f;
        }
    }
    __syncthreads();            // every read of params above precedes every Adam write below (block-local:
                                // the loss lanes live in block 0 and read only epsilon_p / epsilon, whose
                                // writers run after this barrier in block 0 or never touch those leaves)
    if (!live) return;
    a.grads[i] = g;
    if (i == a.P && a.loss_hist && a.step_dev) a.loss_hist[(long long)(a.step_dev[0] - 1) % a.loss_hist_cap] = g;
    if (adam) {
        float bc1, bc2;
        adam_bias_corrections(tstep, bc1, bc2);
        adam_apply(p_old, g, m_old, v_old, a.lr, bc1, bc2);
        a.params_rw[i] = p_old; a.m[i] = m_old; a.v[i] = v_old;
    }
}

int launch_finalize(const FinalizeArgs& a, hipStream_t st) {
    const long long n = a.P + kExtra;
    ProfScope ps(a.params_rw ? "finalize_adam" : "finalize", st);
    // block b covers outputs [n - 64(b+1), n - 64b): with a floor `lo` only the blocks reaching above it are launched
    const long long nblk = (n - std::max(0ll, a.lo) + 63) / 64;
    launch_k(ps, finalize_kernel, dim3((unsigned)nblk), dim3(1024), 0, st, a);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

// Large models (C3: 1.06 M parameters, up to 256 slabs per layer): the weights and biases [0, hi) are a plain streaming reduction --
// 16-byte loads, slab after slab in ascending order (the order finalize_kernel and sum_slabs_kernel use), Adam in the
// same pass -- and finalize_kernel keeps only the tail it exists for (epsilon_p, epsilon, the loss sums), which
// 64-output workgroups reading 256-byte runs of each slab did at 0.6 TB/s.
__global__ __launch_bounds__(256) void bulk_finalize_kernel(const FinalizeArgs a, const long long hi) {
    const long long i0 = 4 * ((long long)blockIdx.x * blockDim.x + threadIdx.x);
    if (i0 >= hi) return;
    auto slabs_of = [&](long long i) { int S = a.S;
        for (int k = a.nseg - 1; k >= 0; --k) if (i < a.seg_end[k]) S = a.seg_S[k];
        return S; };
    float bc1 = 1.f, bc2 = 1.f;
    if (a.params_rw) adam_bias_corrections(a.step_dev[0], bc1, bc2);
    const int S0 = slabs_of(i0);
    const bool vec = i0 + 3 < hi && slabs_of(i0 + 3) == S0 && a.slab_stride % 4 == 0 &&
                     (((uintptr_t)a.slabs | (uintptr_t)a.grads | (uintptr_t)a.params_rw | (uintptr_t)a.m | (uintptr_t)a.v) & 15) == 0;
    if (vec) {
        // Adam state first: it does not depend on the sum, and issued here its latency hides under the slab loads
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), m = p, v = p;
        if (a.params_rw) {
            p = *reinterpret_cast<const float4*>(a.params_rw + i0); m = *reinterpret_cast<const float4*>(a.m + i0);
            v = *reinterpret_cast<const float4*>(a.v + i0);
        }
        // 8 independent 16-byte loads in flight, summed in slab order; every load unconditional (clamped slab index, the
        // surplus selected away: a select on a loaded VALUE is a v_cndmask, a load under a condition is a wait)
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < S0; s0 += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                t[u] = *reinterpret_cast<const float4*>(a.slabs + (long long)min(s0 + u, S0 - 1) * a.slab_stride + i0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = s0 + u < S0;
                g.x += in ? t[u].x : 0.f; g.y += in ? t[u].y : 0.f; g.z += in ? t[u].z : 0.f; g.w += in ? t[u].w : 0.f;
            }
        }
        *reinterpret_cast<float4*>(a.grads + i0) = g;
        if (a.params_rw) {
            adam_apply(p.x, g.x, m.x, v.x, a.lr, bc1, bc2); adam_apply(p.y, g.y, m.y, v.y, a.lr, bc1, bc2);
            adam_apply(p.z, g.z, m.z, v.z, a.lr, bc1, bc2); adam_apply(p.w, g.w, m.w, v.w, a.lr, bc1, bc2);
            *reinterpret_cast<float4*>(a.params_rw + i0) = p; *reinterpret_cast<float4*>(a.m + i0) = m;
            *reinterpret_cast<float4*>(a.v + i0) = v;
        }
        return;
    }
    // A float4 that straddles a layer boundary (different slab counts), the end of the range or an unaligned buffer: element by
    // element, but STILL with 8 loads in flight.  The plain `for (s) g += slabs[s * stride + i]` that used to stand here was one
    // dependent round trip per slab, and ONE such thread at the seam of a 256-slab layer (C3's 6 -> 512 input layers) held the
    // whole launch for 180 of its 199 us -- the streaming part itself runs in ~35 us.
    for (long long i = i0; i < std::min(hi, i0 + 4); ++i) {
        const int S = slabs_of(i);
        float g = 0.f;
        for (int s0 = 0; s0 < S; s0 += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = a.slabs[(long long)min(s0 + u, S - 1) * a.slab_stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) g += s0 + u < S ? t[u] : 0.f;
        }
        a.grads[i] = g;
        if (a.params_rw) {
            float p = a.params_rw[i], m = a.m[i], v = a.v[i];
            adam_apply(p, g, m, v, a.lr, bc1, bc2);
            a.params_rw[i] = p; a.m[i] = m; a.v[i] = v;
        }
    }
}

int launch_bulk_finalize(const FinalizeArgs& a, int64_t hi, hipStream_t st) {
    if (hi <= 0) return VAEK_OK;
    ProfScope ps(a.params_rw ? "bulk_finalize_adam" : "bulk_finalize", st);
    launch_k(ps, bulk_finalize_kernel, dim3((unsigned)((hi + 1023) / 1024)), dim3(256), 0, st, a, (long long)hi);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

__global__ __launch_bounds__(256) void adam_kernel(float* params, const float* grads, float* m, float* v,
                                                  long long n, float lr, int step, const int32_t* step_dev,
                                                  float grad_scale) {
    float bc1, bc2;
    adam_bias_corrections(step_dev ? step_dev[0] : step, bc1, bc2);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float p = params[i], mm = m[i], vv = v[i];
        adam_apply(p, grads[i] * grad_scale, mm, vv, lr, bc1, bc2);
        params[i] = p; m[i] = mm; v[i] = vv;
    }
}

int launch_adam(float* params, const float* grads, float* m, float* v, int64_t n, float lr, int step,
                const int32_t* step_dev, float grad_scale, hipStream_t st) {
    if (n <= 0) return VAEK_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    ProfScope ps("adam", st);
    launch_k(ps, adam_kernel, dim3(blocks), dim3(256), 0, st, params, grads, m, v, (long long)n, lr, step,
                       step_dev, grad_scale);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

__global__ __launch_bounds__(256) void add_noise_kernel(const float* y_lin, const float* y_sig, const float* z2,
                                                       const float* eps_param, float eps_cli, float* x_hat,
                                                       long long n) {
    const float eps = eps_param ? eps_param[0] * eps_cli : eps_cli;
    const float sigma = expf(0.5f * eps);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = y_lin[i] + sigma * z2[i];
        if (y_sig) v += sigmoidf_(y_sig[i]);
        x_hat[i] = v;
    }
}

int launch_add_noise(const float* y_lin, const float* y_sig, const float* z2, const float* eps_param,
                     float eps_cli, float* x_hat, int64_t n, hipStream_t st) {
    if (n <= 0) return VAEK_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    ProfScope ps("add_noise", st);
    launch_k(ps, add_noise_kernel, dim3(blocks), dim3(256), 0, st, y_lin, y_sig, z2, eps_param, eps_cli,
                       x_hat, (long long)n);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

// single block: sums S partials of {mse, musq, deps}; lv / eps from explicit pointers or values
__global__ __launch_bounds__(64) void out4_kernel(const float* partial, int S, const float* lv,
                                                 const float* eps_param, float eps_cli, int L, int D, float rows,
                                                 float inv_bt, float* out4, int want_deps) {
    // lane l sums splits l, l + 64, ... ascending, then a fixed butterfly: one thread walking 512 splits was 45 us
    float smse = 0.f, smusq = 0.f, sdeps = 0.f;
    for (int s = threadIdx.x; s < S; s += 64) {
        smse += partial[s * 4 + 0];
        smusq += partial[s * 4 + 1];
        sdeps += partial[s * 4 + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { smse += __shfl_xor(smse, o, 64); smusq += __shfl_xor(smusq, o, 64); sdeps += __shfl_xor(sdeps, o, 64); }
    if (threadIdx.x != 0) return;
    float klc = 0.f;
    for (int l = 0; l < L; ++l) klc += 1.f + lv[l] - expf(lv[l]);
    const float eps = eps_param ? eps_param[0] * eps_cli : eps_cli;
    const float dkl = (0.5f * smusq - 0.5f * rows * klc) * inv_bt;
    const float mse = (smse + 0.5f * rows * (float)D * (kLog2Pi + eps)) * inv_bt;
    out4[0] = dkl + mse; out4[1] = dkl; out4[2] = mse;
    out4[3] = want_deps ? (sdeps + 0.5f * rows * (float)D) * inv_bt : eps;
}

int launch_elbo_out4(const float* partial, int S, const float* lv, int L, int D, const float* eps_param,
                     float eps, float rows, float inv_bt, float* out4, hipStream_t st) {
    ProfScope ps("out4", st);
    launch_k(ps, out4_kernel, dim3(1), dim3(64), 0, st, partial, S, lv,
                       eps_param, eps, L, D, rows, inv_bt, out4, 1);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int launch_eval_out4(const float* partial, int S, const float* params, int64_t off_epsp,
                     int64_t off_eps, int L, int D, float eps_cli, float rows, float inv_bt, float* out4,
                     hipStream_t st) {
    ProfScope ps("out4", st);
    launch_k(ps, out4_kernel, dim3(1), dim3(64), 0, st, partial, S, params + off_epsp,
                       off_eps >= 0 ? params + off_eps : (const float*)nullptr, eps_cli, L, D, rows, inv_bt,
                       out4, 0);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* slabs, long long stride, int S, float* out,
                                                       long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g = 0.f;
    for (int s0 = 0; s0 < S; s0 += 8) {         // 8 unconditional loads in flight (clamped index), summed in slab order
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = slabs[(long long)min(s0 + u, S - 1) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) g += s0 + u < S ? t[u] : 0.f;
    }
    out[i] = g;
}

// Many slabs, few outputs (the streaming convolution kernels' 1024 block partials of 544 floats; a bias gradient's 512 partials of one
// float): sum_slabs_kernel would be 2 workgroups walking 1024 dependent rounds.  First every group of 32 consecutive slabs is summed,
// ascending, into the group's FIRST slab (a thread reads only its own element of its own group's slabs before it writes), then the
// group sums are summed ascending: a fixed order again.
__global__ __launch_bounds__(256) void sum_slab_groups_kernel(float* slabs, long long stride, int S, int GS, long long n) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x, i = id % n;
    const int g = (int)(id / n), s_lo = g * GS, s_hi = min(S, s_lo + GS);
    if (s_lo >= S) return;
    float acc = 0.f;
    for (int s0 = s_lo; s0 < s_hi; s0 += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = slabs[(long long)min(s0 + u, s_hi - 1) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += s0 + u < s_hi ? t[u] : 0.f;
    }
    slabs[(long long)s_lo * stride + i] = acc;
}
int launch_sum_slabs_inplace(float* slabs, int64_t stride, int S, float* out, int64_t n, hipStream_t st) {
    constexpr int GS = 32;
    if (n <= 0) return VAEK_OK;
    if (S < 4 * GS || n > 65536) return launch_sum_slabs(slabs, stride, S, out, n, st);
    const int G = (S + GS - 1) / GS;
    {
        ProfScope ps("sum_slab_groups", st);
        launch_k(ps, sum_slab_groups_kernel, dim3((unsigned)((n * G + 255) / 256)), dim3(256), 0, st, slabs, (long long)stride, S, GS, (long long)n);
        VAEK_HIP_CHECK(hipGetLastError());
    }
    return launch_sum_slabs(slabs, stride * GS, G, out, n, st);
}

int launch_sum_slabs(const float* slabs, int64_t stride, int S, float* out, int64_t n, hipStream_t st) {
    if (n <= 0) return VAEK_OK;
    ProfScope ps("sum_slabs", st);
    launch_k(ps, sum_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slabs,
                       (long long)stride, S, out, (long long)n);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

}  // namespace vaek
