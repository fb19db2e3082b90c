// C ABI of libvaek.so (include/vaek.h): context, flat parameter layout, workspace carving and the
// layer-by-layer orchestration of VAE.train_step / VAE.loss / VAE.apply
// (/root/reference/networks.py:87-101, :103-113, :61-84).
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <utility>

#include "vaek_internal.h"

namespace vaek {

static thread_local char g_err[512] = "";
thread_local Profiler* g_prof = nullptr;
struct ProfBind {   // entry points bind the context's profiler for the duration of the call
    explicit ProfBind(vaek_ctx* c) { g_prof = c ? &c->prof : nullptr; }
    ~ProfBind() { g_prof = nullptr; }
};
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr int kElboBlockSplits = 512;           // row splits of the ELBO block entry point (vaek_elbo_fwd_bwd)

static void add_net(vaek_ctx* c, Net& net, const char* name, int fan_in, const int* hidden, int n_hidden, int last) {
    int k = fan_in;
    for (int i = 0; i <= n_hidden; ++i) {
        const int n = i < n_hidden ? hidden[i] : last;
        Layer l;
        l.n_in = k; l.n_out = n; l.w_off = c->P; l.relu = i < n_hidden;
        char nm[64];
        snprintf(nm, sizeof(nm), "%s/FC%d/kernel", name, i);
        c->leaves.push_back({nm, c->P, k, n});
        c->P += (int64_t)k * n;
        snprintf(nm, sizeof(nm), "%s/FC%d/bias", name, i);
        c->leaves.push_back({nm, c->P, 1, n});
        c->P += n;
        net.layers.push_back(l);
        c->max_width = std::max(c->max_width, std::max(k, n));
        k = n;
    }
}

static size_t carve_acts(vaek_ctx* c, Net& net, size_t off) {
    for (auto& l : net.layers) {
        net.act_off.push_back(off);
        off = align_up(off + (size_t)c->B * l.n_out * sizeof(float), 256);
    }
    return off;
}

static int check_ws(const vaek_ctx* c, const void* ws) {
    if (!ws || (reinterpret_cast<uintptr_t>(ws) & 255)) {
        set_error("workspace must be a non-null, 256-byte aligned device pointer of vaek_workspace_bytes() bytes");
        return VAEK_ERR_WORKSPACE;
    }
    (void)c;
    return VAEK_OK;
}

// dW|db slabs are flat-gradient images at a 64-float-aligned pitch (16-byte loads in the streaming reductions)
static inline int64_t slab_stride(const vaek_ctx* c) { return (int64_t)align_up((size_t)c->P, 64); }

template <typename T>
static T* at(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }

// dtype = VAEK_BF16 routes the WIDE layers (both dims >= 64) through the bf16 matrix-core kernels; skinny
// input/output layers (D, L of a handful of features) stay on the exact f32 kernels -- they are
// bandwidth-bound and carry the reparameterisation / ELBO inputs.
static bool use_bf16(const vaek_ctx* c, int n_in, int n_out) { return c->cfg.dtype == VAEK_BF16 && n_in >= 64 && n_out >= 64; }

// bf16-STORAGE mode of a stack (Net::b16, vaek_internal.h): with at least two hidden layers whose widths are all multiples
// of 64, the hidden activations / gradients are kept as bf16 and the hidden -> hidden layers run gemm_bf16s.hip.  Other
// stacks under dtype = VAEK_BF16 keep f32 storage and the round-1 kernels (gemm_bf16.hip) for their wide layers.
static bool net_is_b16(const vaek_ctx* c, const Net& net) {
    if (c->cfg.dtype != VAEK_BF16 || net.layers.size() < 3) return false;
    for (size_t i = 0; i + 1 < net.layers.size(); ++i)
        if (net.layers[i].n_out % 64) return false;
    return true;
}
static inline __bf16* wb16(const vaek_ctx* c, void* ws, const Net& net, size_t i, bool transposed) {
    const Layer& l = net.layers[i];
    return reinterpret_cast<__bf16*>(static_cast<char*>(ws) + c->ws_wb16) + net.wb_off[i] + (transposed ? (int64_t)l.n_in * l.n_out : 0);
}
// bf16 copies (W and W^T) of every hidden -> hidden kernel, once per entry point that runs a b16 stack
static int convert_weights(vaek_ctx* c, const float* params, void* ws, hipStream_t st) {
    int K[24], N[24], n = 0; int64_t woff[24], ooff[24];
    for (const Net* net : {&c->enc, &c->dec, &c->sig}) {
        if (!net->b16) continue;
        for (size_t i = 1; i + 1 < net->layers.size(); ++i) {
            if (n == 24) { set_error("too many wide layers"); return VAEK_ERR_INVALID; }
            K[n] = net->layers[i].n_in; N[n] = net->layers[i].n_out; woff[n] = net->layers[i].w_off; ooff[n] = net->wb_off[i]; ++n;
        }
    }
    int rc = launch_cvt_weights(params, reinterpret_cast<__bf16*>(static_cast<char*>(ws) + c->ws_wb16), K, N, woff, ooff, n, st);
    if (rc) return rc;
    // 32-row padded bf16 copies for the skinny first / last layers (gemm_skinny16.hip)
    int H[16], d[16], tr[16], m = 0; int64_t sw[16], so[16];
    for (const Net* net : {&c->enc, &c->dec, &c->sig}) {
        if (!net->b16) continue;
        const Layer& f = net->layers.front(); const Layer& l = net->layers.back();
        if (f.sk && m < 16) { H[m] = f.n_out; d[m] = f.n_in; tr[m] = 0; sw[m] = f.w_off; so[m] = f.sk_off; ++m; }
        if (l.sk && m < 16) { H[m] = l.n_in; d[m] = l.n_out; tr[m] = 1; sw[m] = l.w_off; so[m] = l.sk_off; ++m; }
    }
    return launch_sk_prep(params, reinterpret_cast<__bf16*>(static_cast<char*>(ws) + c->ws_sk16), H, d, tr, sw, so, m, st);
}
static inline __bf16* sk16(const vaek_ctx* c, void* ws, const Layer& l) {
    return reinterpret_cast<__bf16*>(static_cast<char*>(ws) + c->ws_sk16) + l.sk_off;
}

// ---- forward through one Dense/relu stack; `reparam` fuses networks.py:73-74 into the last layer
struct ElboFuse {           // decoder's last layer with the ELBO epilogue (gemm_f32.hip EPI_ELBO): inputs, and the tile map out
    const float* x; const float* z2; const float* eps_param; float eps_cli, inv_bt; float* part; int bm, nbx;
};

static int net_forward(vaek_ctx* c, const Net& net, const float* params, const float* in, void* ws, int rows,
                       bool reparam, const float* z1, hipStream_t st, ElboFuse* ef = nullptr) {
    const float* h = in;
    for (size_t i = 0; i < net.layers.size(); ++i) {
        const Layer& l = net.layers[i];
        const float* w = params + l.w_off;
        const float* b = w + (int64_t)l.n_in * l.n_out;
        float* y = at<float>(ws, net.act_off[i]);
        int rc;
        if (net.b16) {
            const bool last = i + 1 == net.layers.size();
            const __bf16* h16p = reinterpret_cast<const __bf16*>(h);
            if (i == 0 && l.sk) rc = launch_sk_first_fwd(h, w, b, reinterpret_cast<__bf16*>(y), rows, l.n_in, l.n_out, l.relu, st);
            else if (i == 0) rc = launch_dense_fwd_out16(h, w, b, reinterpret_cast<__bf16*>(y), rows, l.n_in, l.n_out, l.relu, st);
            else if (!last) rc = launch_hs_fwd(h16p, wb16(c, ws, net, i, true), b, reinterpret_cast<__bf16*>(y), rows, l.n_in, l.n_out, l.relu, st);
            else if (l.sk && ef) rc = launch_sk_last_fwd_elbo(h16p, sk16(c, ws, l), b, y, ef->x, ef->z2, ef->eps_param, ef->eps_cli, ef->inv_bt,
                                                              ef->part, rows, l.n_in, l.n_out, &ef->bm, &ef->nbx, st);
            else if (l.sk && reparam) rc = launch_sk_last_fwd_reparam(h16p, sk16(c, ws, l), b, y, at<float>(ws, c->ws_samples), z1,
                                                                      params + c->off_epsp, rows, l.n_in, l.n_out, st);
            else if (l.sk) rc = launch_sk_last_fwd(h16p, sk16(c, ws, l), b, y, rows, l.n_in, l.n_out, st);
            else if (ef) rc = launch_dense_fwd_elbo_in16(h16p, w, b, y, ef->x, ef->z2, ef->eps_param, ef->eps_cli, ef->inv_bt, ef->part, rows,
                                                         l.n_in, l.n_out, &ef->bm, &ef->nbx, st);
            else if (reparam) rc = launch_dense_fwd_reparam_in16(h16p, w, b, y, at<float>(ws, c->ws_samples), z1, params + c->off_epsp, rows,
                                                                 l.n_in, l.n_out, st);
            else rc = launch_dense_fwd_in16(h16p, w, b, y, rows, l.n_in, l.n_out, st);
            if (rc) return rc;
            h = y;
            continue;
        }
        const bool h16 = use_bf16(c, l.n_in, l.n_out);
        if (ef && i + 1 == net.layers.size())
            rc = launch_dense_fwd_elbo(h, w, b, y, ef->x, ef->z2, ef->eps_param, ef->eps_cli, ef->inv_bt, ef->part, rows, l.n_in,
                                       l.n_out, &ef->bm, &ef->nbx, st);
        else if (reparam && i + 1 == net.layers.size())
            rc = (h16 ? launch_dense_fwd_reparam_bf16 : launch_dense_fwd_reparam)(h, w, b, y, at<float>(ws, c->ws_samples), z1,
                                                                                  params + c->off_epsp, rows, l.n_in, l.n_out, st);
        else
            rc = (h16 ? launch_dense_fwd_bf16 : launch_dense_fwd)(h, w, b, y, rows, l.n_in, l.n_out, l.relu, st);
        if (rc) return rc;
        h = y;
    }
    return VAEK_OK;
}

// ---- backward through one stack.  d_out: gradient w.r.t. the last Dense output (lives in the
// stack's last activation buffer).  dx_first: where dL/d(input) goes (nullptr: not needed).
struct BucketSink {          // bucketed mode: where finished layers go and which event announces them
    float* grads; void* const* events; int next;
};

static int net_backward(vaek_ctx* c, const Net& net, const float* params, const float* in, float* d_out, void* ws,
                        float* dx_first, bool accumulate_first, hipStream_t st, BucketSink* sink = nullptr) {
    float* d = d_out;
    float* gb[2] = {at<float>(ws, c->ws_gbuf0), at<float>(ws, c->ws_gbuf1)};
    int tog = 0;
    float* slabs = at<float>(ws, c->ws_slabs);
    for (int i = (int)net.layers.size() - 1; i >= 0; --i) {
        const Layer& l = net.layers[i];
        const float* w = params + l.w_off;
        const float* h_in = i == 0 ? in : at<float>(ws, net.act_off[i - 1]);
        const bool h16 = !net.b16 && use_bf16(c, l.n_in, l.n_out);
        int rc;
        if (net.b16) {
            // hidden tensors are bf16: d is f32 only for the last layer (dL/d output), h_in is f32 only for layer 0
            const bool last = i + 1 == (int)net.layers.size();
            const __bf16* h_in16 = reinterpret_cast<const __bf16*>(h_in);
            const __bf16* d16 = reinterpret_cast<const __bf16*>(d);
            float* skpart = at<float>(ws, c->ws_skpart);
            if (last && l.sk)        // dW|db AND dX (into the ping-pong buffer the dX branch below would have used) from one pass over h
                rc = launch_sk_last_bwd(h_in16, d, w, reinterpret_cast<__bf16*>(gb[tog]), skpart, slabs + l.w_off, slab_stride(c), l.S, c->B,
                                        l.n_in, l.n_out, st);
            else if (i == 0 && l.sk) rc = launch_sk_first_bwd(h_in, d16, skpart, slabs + l.w_off, slab_stride(c), l.S, c->B, l.n_out, l.n_in, st);
            else if (last) rc = launch_dense_bwd_dw_x16(h_in16, d, slabs + l.w_off, slab_stride(c), l.S, l.rows_per_split, c->B, l.n_in, l.n_out, st);
            else if (i == 0) rc = launch_dense_bwd_dw_dy16(h_in, d16, slabs + l.w_off, slab_stride(c), l.S, l.rows_per_split, c->B, l.n_in, l.n_out, st);
            else rc = launch_hs_dw(h_in16, d16, slabs + l.w_off, slab_stride(c), l.S, l.rows_per_split, c->B, l.n_in, l.n_out, st);
        } else {
            rc = (h16 ? launch_dense_bwd_dw_bf16 : launch_dense_bwd_dw)(h_in, d, slabs + l.w_off, slab_stride(c), l.S, l.rows_per_split,
                                                                        c->B, l.n_in, l.n_out, st);
        }
        if (rc) return rc;
        if (sink) {     // this layer's [kernel | bias] slice is final once its slabs are summed: announce it
            const int64_t cnt = (int64_t)(l.n_in + 1) * l.n_out;
            if ((rc = launch_sum_slabs(slabs + l.w_off, slab_stride(c), l.S, sink->grads + l.w_off, cnt, st))) return rc;
            VAEK_HIP_CHECK(hipEventRecord((hipEvent_t)sink->events[sink->next++], st));
        }
        if (i > 0) {
            float* dx = gb[tog];
            tog ^= 1;
            if (net.b16) {
                const __bf16* h_in16 = reinterpret_cast<const __bf16*>(h_in);
                if (i + 1 == (int)net.layers.size())
                    rc = l.sk ? VAEK_OK      // already written by launch_sk_last_bwd above
                              : launch_dense_bwd_dx_out16(d, w, h_in16, reinterpret_cast<__bf16*>(dx), c->B, l.n_in, l.n_out, false, st);
                else
                    rc = launch_hs_dx(reinterpret_cast<const __bf16*>(d), wb16(c, ws, net, i, false), h_in16, reinterpret_cast<__bf16*>(dx),
                                      c->B, l.n_in, l.n_out, st);
            } else {
                rc = (h16 ? launch_dense_bwd_dx_bf16 : launch_dense_bwd_dx)(d, w, h_in, dx, c->B, l.n_in, l.n_out, true, false, st);
            }
            if (rc) return rc;
            d = dx;
        } else if (dx_first) {
            if (net.b16 && l.sk)
                rc = launch_sk_first_dx(reinterpret_cast<const __bf16*>(d), sk16(c, ws, l), dx_first, c->B, l.n_out, l.n_in, accumulate_first, st);
            else if (net.b16)
                rc = launch_dense_bwd_dx_in16(reinterpret_cast<const __bf16*>(d), w, dx_first, c->B, l.n_in, l.n_out, accumulate_first, st);
            else
                rc = (h16 ? launch_dense_bwd_dx_bf16 : launch_dense_bwd_dx)(d, w, nullptr, dx_first, c->B, l.n_in, l.n_out, false,
                                                                            accumulate_first, st);
            if (rc) return rc;
        }
    }
    return VAEK_OK;
}

static int generic_grads(vaek_ctx* c, const float* params, int32_t* step_dev, const float* x, const float* z1,
                         const float* z2, void* ws, hipStream_t st, BucketSink* sink = nullptr) {
    const bool sig = c->cfg.sigmoid_decoder != 0;
    const float inv_bt = (float)(1.0 / (double)c->Bt);
    int rc;
    if ((rc = convert_weights(c, params, ws, st))) return rc;
    if ((rc = net_forward(c, c->enc, params, x, ws, c->B, true, z1, st))) return rc;
    const float* samples = at<float>(ws, c->ws_samples);
    float* mu = at<float>(ws, c->enc.act_off.back());
    float* y_lin = at<float>(ws, c->dec.act_off.back());
    const float* eps_param = c->off_eps >= 0 ? params + c->off_eps : nullptr;
    const Layer& last = c->dec.layers.back();
    if (c->lwd) {
        // wide linear decoder: x and z2 read ONCE for the decoder's forward, the ELBO pass, dL/d samples and the decoder's kernel
        // gradient (linear_wide.hip); dL/d x_hat never exists in memory
        const float* wd = params + last.w_off;
        float* slabs = at<float>(ws, c->ws_slabs);
        float* part = at<float>(ws, c->ws_eblk);
        float* gpart = at<float>(ws, c->ws_lwd);
        if ((rc = launch_lwd(samples, wd, wd + (int64_t)last.n_in * last.n_out, x, z2, eps_param, c->cfg.eps_cli, inv_bt, gpart, slabs + last.w_off,
                             slab_stride(c), part, c->B, c->D, c->L, c->lwd_rb, st)))
            return rc;
        if (sink) {
            const int64_t cnt = (int64_t)(last.n_in + 1) * last.n_out;
            if ((rc = launch_sum_slabs(slabs + last.w_off, slab_stride(c), last.S, sink->grads + last.w_off, cnt, st))) return rc;
            VAEK_HIP_CHECK(hipEventRecord((hipEvent_t)sink->events[sink->next++], st));
        }
        float* dsamp = at<float>(ws, c->ws_dsamp);
        if ((rc = launch_lwd_second(gpart, c->D / 256, dsamp, mu, z1, at<float>(ws, c->ws_rpart), c->B, c->L, c->Se, c->rows_per_esplit, inv_bt, part,
                                    last.S * (c->D / 256), at<float>(ws, c->ws_epart), step_dev, st)))
            return rc;
        return net_backward(c, c->enc, params, x, dsamp, ws, nullptr, false, st, sink);
    }
    if (!sig && (c->dec.b16 || !use_bf16(c, last.n_in, last.n_out))) {
        // one decoder, exact f32 output layer: the ELBO's elementwise pass runs in that layer's epilogue -- its output never
        // goes to HBM, dL/dx_hat lands where the backward pass expects it
        ElboFuse ef{x, z2, eps_param, c->cfg.eps_cli, inv_bt, at<float>(ws, c->ws_eblk), 0, 0};
        if ((rc = net_forward(c, c->dec, params, samples, ws, c->B, false, nullptr, st, &ef))) return rc;
        if ((rc = launch_elbo_reduce(ef.part, ef.bm, ef.nbx, mu, at<float>(ws, c->ws_epart), c->B, c->L, c->Se, c->rows_per_esplit,
                                     step_dev, st)))
            return rc;
        float* dsamp = at<float>(ws, c->ws_dsamp);
        if ((rc = net_backward(c, c->dec, params, samples, y_lin, ws, dsamp, false, st, sink))) return rc;
        if ((rc = launch_reparam_bwd(dsamp, mu, z1, at<float>(ws, c->ws_rpart), c->B, c->L, c->Se, c->rows_per_esplit, inv_bt, st)))
            return rc;
        return net_backward(c, c->enc, params, x, dsamp, ws, nullptr, false, st, sink);
    }
    if ((rc = net_forward(c, c->dec, params, samples, ws, c->B, false, nullptr, st))) return rc;
    if (sig && (rc = net_forward(c, c->sig, params, samples, ws, c->B, false, nullptr, st))) return rc;
    float* y_sig = sig ? at<float>(ws, c->sig.act_off.back()) : nullptr;
    ElboArgs e{};
    e.x = x; e.y_lin = y_lin; e.y_sig = y_sig; e.z2 = z2; e.mu = mu;
    e.eps_param = eps_param;
    e.eps_cli = c->cfg.eps_cli;
    e.d_lin = y_lin; e.d_sig = y_sig;
    e.partial = at<float>(ws, c->ws_epart);
    e.rows = c->B; e.D = c->D; e.L = c->L; e.S = c->Se; e.rows_per_split = c->rows_per_esplit;
    e.inv_bt = inv_bt; e.step_dev = step_dev;
    if ((rc = launch_elbo(e, st))) return rc;
    float* dsamp = at<float>(ws, c->ws_dsamp);
    if ((rc = net_backward(c, c->dec, params, samples, y_lin, ws, dsamp, false, st, sink))) return rc;
    if (sig && (rc = net_backward(c, c->sig, params, samples, y_sig, ws, dsamp, true, st, sink))) return rc;
    if ((rc = launch_reparam_bwd(dsamp, mu, z1, at<float>(ws, c->ws_rpart), c->B, c->L, c->Se, c->rows_per_esplit,
                                 inv_bt, st)))
        return rc;
    return net_backward(c, c->enc, params, x, dsamp, ws, nullptr, false, st, sink);
}

constexpr int64_t kBulkFinalizeMin = 65536;      // parameters below which the 64-output finalize alone is faster

static int generic_finalize(vaek_ctx* c, const float* params, float* grads, float* params_rw, float* m, float* v,
                            const int32_t* step_dev, float lr, void* ws, hipStream_t st, int64_t lo = 0) {
    FinalizeArgs f{};
    f.lo = lo;
    f.slabs = at<float>(ws, c->ws_slabs); f.slab_stride = slab_stride(c); f.S = c->S;
    f.nseg = 0;
    for (const Net* net : {&c->enc, &c->dec, &c->sig})
        for (const auto& l : net->layers) {
            f.seg_end[f.nseg] = (int)(l.w_off + (int64_t)(l.n_in + 1) * l.n_out);
            f.seg_S[f.nseg++] = l.S;
        }
    f.epart = at<float>(ws, c->ws_epart); f.rpart = at<float>(ws, c->ws_rpart); f.Se = c->Se;
    f.P = c->P; f.off_epsp = c->off_epsp; f.off_eps = c->off_eps; f.L = c->L; f.D = c->D;
    f.params = params; f.eps_cli = c->cfg.eps_cli;
    f.rows_over_bt = (float)((double)c->B / (double)c->Bt); f.inv_bt = (float)(1.0 / (double)c->Bt);
    f.rows = (float)c->B;
    f.grads = grads; f.params_rw = params_rw; f.m = m; f.v = v; f.step_dev = step_dev; f.lr = lr;
    // Adam is fused into the finalize kernel only while epsilon_p, epsilon and the scalar sums all sit
    // in its block 0 (64 outputs): the loss lanes read epsilon_p, so its Adam writers must be behind the
    // same block-local barrier.  Wider latents take a separate Adam launch.
    f.loss_hist = c->cfg.world == 1 ? c->loss_hist : nullptr; f.loss_hist_cap = c->loss_hist_cap;
    if (!f.step_dev) f.loss_hist = nullptr;
    const bool fuse = params_rw != nullptr && c->L + 5 <= 64;
    int rc;
    int64_t adam_from = 0;
    if (lo == 0 && c->off_epsp >= kBulkFinalizeMin) {
        // big models: weights and biases as a streaming pass (with Adam), the 64-output finalize only for the tail
        if ((rc = launch_bulk_finalize(f, c->off_epsp, st))) return rc;
        f.lo = adam_from = c->off_epsp;
    }
    if (!fuse) f.params_rw = nullptr;
    if ((rc = launch_finalize(f, st)) || params_rw == nullptr || fuse) return rc;
    return launch_adam(params_rw + adam_from, grads + adam_from, m + adam_from, v + adam_from, c->P - adam_from, lr, 0, step_dev, 1.f, st);
}

}  // namespace vaek

using namespace vaek;

extern "C" {

int vaek_version(void) { return VAEK_VERSION; }
const char* vaek_last_error(void) { return g_err; }

int vaek_ctx_create(const vaek_config* cfg, vaek_ctx** out) {
    if (!cfg || !out) { set_error("null argument"); return VAEK_ERR_INVALID; }
    if (cfg->struct_size != (int32_t)sizeof(vaek_config)) {
        set_error("vaek_config.struct_size %d != %zu (header/library mismatch)", cfg->struct_size, sizeof(vaek_config));
        return VAEK_ERR_INVALID;
    }
    if (cfg->batch <= 0 || cfg->data_dim <= 0 || cfg->latent_dim <= 0 || cfg->n_enc_hidden < 0 ||
        cfg->n_enc_hidden > VAEK_MAX_HIDDEN || cfg->n_dec_hidden < 0 || cfg->n_dec_hidden > VAEK_MAX_HIDDEN ||
        cfg->world < 1 || cfg->rank < 0 || cfg->rank >= cfg->world) {
        set_error("invalid geometry: batch=%d D=%d L=%d n_enc=%d n_dec=%d world=%d rank=%d", cfg->batch,
                  cfg->data_dim, cfg->latent_dim, cfg->n_enc_hidden, cfg->n_dec_hidden, cfg->world, cfg->rank);
        return VAEK_ERR_INVALID;
    }
    for (int i = 0; i < cfg->n_enc_hidden; ++i)
        if (cfg->enc_hidden[i] <= 0) { set_error("encoder hidden width %d <= 0", cfg->enc_hidden[i]); return VAEK_ERR_INVALID; }
    for (int i = 0; i < cfg->n_dec_hidden; ++i)
        if (cfg->dec_hidden[i] <= 0) { set_error("decoder hidden width %d <= 0", cfg->dec_hidden[i]); return VAEK_ERR_INVALID; }
    if (cfg->latent_dim > 250) { set_error("latent_dim %d > 250 unsupported", cfg->latent_dim); return VAEK_ERR_INVALID; }
    if (cfg->dtype != VAEK_F32 && cfg->dtype != VAEK_BF16) { set_error("unknown dtype %d", cfg->dtype); return VAEK_ERR_INVALID; }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible: libvaek.so has no CPU fallback");
        return VAEK_ERR_NO_DEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { set_error("device %d out of range (%d visible)", cfg->device, ndev); return VAEK_ERR_INVALID; }
    hipDeviceProp_t prop;
    VAEK_HIP_CHECK(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);
        return VAEK_ERR_NO_DEVICE;
    }

    vaek_ctx* c = new (std::nothrow) vaek_ctx();
    if (!c) { set_error("out of host memory"); return VAEK_ERR_INVALID; }
    c->cfg = *cfg;
    c->B = cfg->batch; c->D = cfg->data_dim; c->L = cfg->latent_dim;
    c->Bt = cfg->global_batch > 0 ? cfg->global_batch : (int64_t)cfg->batch * cfg->world;
    c->P = 0; c->max_width = std::max(c->D, c->L);
    c->n_cu = prop.multiProcessorCount;
    add_net(c, c->enc, "Encoder", c->D, cfg->enc_hidden, cfg->n_enc_hidden, c->L);
    add_net(c, c->dec, "Decoder", c->L, cfg->dec_hidden, cfg->n_dec_hidden, c->D);
    if (cfg->sigmoid_decoder) add_net(c, c->sig, "SigDecoder", c->L, cfg->dec_hidden, cfg->n_dec_hidden, c->D);
    c->off_epsp = c->P;
    c->leaves.push_back({"epsilon_p", c->P, 1, c->L});
    c->P += c->L;
    c->off_eps = -1;
    if (cfg->tunable_eps) {
        c->off_eps = c->P;
        c->leaves.push_back({"epsilon", c->P, 1, 1});
        c->P += 1;
    }
    // batch splits: dW|db GEMM slabs sized so that (tiles x S) fills the chip; elementwise
    // partials one per <= 1024 row-blocks
    c->S = 1; c->rows_per_split = c->B;
    auto splits = [&](Net& n) {
        for (auto& l : n.layers) {
            // wide layers run 128 x 128 tiles (gemm_f32.hip) and want ~3 workgroups per CU; the rest 64-wide tiles
            const bool wide = l.n_in + 1 >= 128 && l.n_out >= 128;
            // ... counted in the tile shape gemm_f32.hip's launch() picks: 128 x 128, 128 x 32 (n_out <= 32), 32 x 128 (n_in < 32)
            const int tiles = wide ? ((l.n_in + 1 + 127) / 128) * ((l.n_out + 127) / 128)
                            : l.n_out <= 32 ? (l.n_in + 1 + 127) / 128
                            : l.n_in + 1 <= 32 ? (l.n_out + 127) / 128
                            : ((l.n_in + 1 + 63) / 64) * ((l.n_out + 63) / 64);
            // at most 64 slabs: the fixed-order slab sum (bulk_finalize_kernel) is a chain of S / 8 memory round trips per output,
            // and a 256-slab skinny layer (C3's 6 -> 512) held the whole finalize launch for ~60 us
            const int s_target = std::min(64, std::max(4, (wide ? 768 : 1024) / tiles));
            l.rows_per_split = std::max(64, (int)align_up((size_t)(c->B + s_target - 1) / s_target, 64));
            l.S = (c->B + l.rows_per_split - 1) / l.rows_per_split;
            if (l.S > c->S) { c->S = l.S; c->rows_per_split = l.rows_per_split; }
        }
    };
    c->enc.b16 = net_is_b16(c, c->enc); c->dec.b16 = net_is_b16(c, c->dec); c->sig.b16 = net_is_b16(c, c->sig);
    splits(c->enc); splits(c->dec); splits(c->sig);
    // hidden -> hidden layers of a bf16-storage stack (gemm_bf16s.hip, 128 x 128 tiles, 64-row k-tiles): ~512 workgroups
    int64_t wb_elems = 0;
    for (Net* net : {&c->enc, &c->dec, &c->sig}) {
        net->wb_off.assign(net->layers.size(), 0);
        if (!net->b16) continue;
        for (size_t i = 1; i + 1 < net->layers.size(); ++i) {
            Layer& l = net->layers[i];
            const int tiles = ((l.n_in + 127) / 128) * ((l.n_out + 127) / 128);
            const int s_target = std::min(256, std::max(1, 512 / tiles));
            l.rows_per_split = std::max(64, (int)align_up((size_t)(c->B + s_target - 1) / s_target, 64));
            l.S = (c->B + l.rows_per_split - 1) / l.rows_per_split;
            net->wb_off[i] = wb_elems;
            wb_elems += 2 * (int64_t)l.n_in * l.n_out;
        }
    }
    // ... and its skinny ends (gemm_skinny16.hip): S slabs, each the fixed-order sum of 64 workgroups' partial images
    int64_t sk_elems = 0; size_t sk_part = 0;
    for (Net* net : {&c->enc, &c->dec, &c->sig}) {
        if (!net->b16) continue;
        for (Layer* l : {&net->layers.front(), &net->layers.back()}) {
            const bool first = l == &net->layers.front();
            const int d = first ? l->n_in : l->n_out, H = first ? l->n_out : l->n_in;
            if (!sk_supported(d, H)) continue;
            l->sk = true;
            l->S = std::min(8, std::max(1, c->B / 4096));
            l->rows_per_split = (c->B + l->S - 1) / l->S;
            l->sk_off = sk_elems; sk_elems += 32ll * H;
            sk_part = std::max(sk_part, sk_partial_bytes(d, H, l->S));
        }
    }
    // wide linear decoder (BASELINE config 4): one fused kernel for its forward, the ELBO pass and both backward products; its
    // [kernel | bias] gradient comes in one slab per ROW BLOCK of that kernel's grid
    c->lwd = !cfg->force_generic && !cfg->sigmoid_decoder && cfg->n_dec_hidden == 0 && !c->dec.b16 && lwd_supported(c->B, c->D, c->L) &&
             !(cfg->dtype == VAEK_BF16 && use_bf16(c, c->L, c->D));
    if (c->lwd) {
        Layer& l = c->dec.layers.back();
        c->lwd_rb = lwd_row_block(c->B, c->D, c->n_cu);
        l.rows_per_split = c->lwd_rb; l.S = (c->B + c->lwd_rb - 1) / c->lwd_rb;
    }
    c->S = 1; c->rows_per_split = c->B;
    for (const Net* net : {&c->enc, &c->dec, &c->sig})
        for (const auto& l : net->layers) if (l.S > c->S) { c->S = l.S; c->rows_per_split = l.rows_per_split; }
    if (c->enc.layers.size() + c->dec.layers.size() + c->sig.layers.size() > 32) {
        set_error("too many layers"); delete c; return VAEK_ERR_INVALID;
    }
    c->rows_per_esplit = std::max(128, (int)align_up((size_t)(c->B + 1023) / 1024, 128));      // whole 64- and 128-row GEMM tiles
    c->Se = (c->B + c->rows_per_esplit - 1) / c->rows_per_esplit;

    size_t off = 0;
    off = carve_acts(c, c->enc, off);
    c->ws_samples = off; off = align_up(off + (size_t)c->B * c->L * sizeof(float), 256);
    off = carve_acts(c, c->dec, off);
    off = carve_acts(c, c->sig, off);
    c->ws_dsamp = off; off = align_up(off + (size_t)c->B * c->L * sizeof(float), 256);
    c->ws_gbuf0 = off; off = align_up(off + (size_t)c->B * c->max_width * sizeof(float), 256);
    c->ws_gbuf1 = off; off = align_up(off + (size_t)c->B * c->max_width * sizeof(float), 256);
    c->ws_slabs = off; off = align_up(off + (size_t)c->S * slab_stride(c) * sizeof(float), 256);
    c->ws_epart = off; off = align_up(off + (size_t)c->Se * 4 * sizeof(float), 256);
    c->ws_epart_blk = off; off = align_up(off + (size_t)kElboBlockSplits * 4 * sizeof(float), 256);      // vaek_elbo_fwd_bwd's own, finer splits
    c->ws_rpart = off; off = align_up(off + (size_t)c->Se * c->L * sizeof(float), 256);
    // {mse, d eps} per output tile of the decoder's last GEMM: at most (B/64) x (D/32) tiles in any of its tile shapes
    c->ws_eblk = off; off = align_up(off + (size_t)((c->B + 63) / 64) * ((c->D + 31) / 32) * 2 * sizeof(float), 256);
    c->fused = !cfg->force_generic && fused_supported(c);
    c->ws_fused = off; off = align_up(off + fused_workspace_bytes(c), 256);
    c->ws_wb16 = off; off = align_up(off + (size_t)wb_elems * sizeof(__bf16), 256);
    c->ws_sk16 = off; off = align_up(off + (size_t)sk_elems * sizeof(__bf16), 256);
    c->ws_skpart = off; off = align_up(off + sk_part, 256);
    c->ws_lin = off; off = align_up(off + lin_steps_workspace_bytes(c), 256);
    c->ws_lwd = off; off = align_up(off + (c->lwd ? lwd_gpart_bytes(c->B, c->D, c->L) : 0), 256);
    c->ws_total = off;
    *out = c;
    return VAEK_OK;
}

int vaek_ctx_destroy(vaek_ctx* ctx) {
    if (!ctx) return VAEK_OK;
    vaek_comm_destroy(ctx);
    for (auto e : ctx->prof.ev) (void)hipEventDestroy(e);
    delete ctx;
    return VAEK_OK;
}

int vaek_param_count(const vaek_ctx* ctx, int64_t* P) {
    if (!ctx || !P) { set_error("null argument"); return VAEK_ERR_INVALID; }
    *P = ctx->P;
    return VAEK_OK;
}
int vaek_grad_len(const vaek_ctx* ctx, int64_t* n) {
    if (!ctx || !n) { set_error("null argument"); return VAEK_ERR_INVALID; }
    *n = ctx->P + kExtra;
    return VAEK_OK;
}
int vaek_leaf_count(const vaek_ctx* ctx, int32_t* n) {
    if (!ctx || !n) { set_error("null argument"); return VAEK_ERR_INVALID; }
    *n = (int32_t)ctx->leaves.size();
    return VAEK_OK;
}
int vaek_leaf_info(const vaek_ctx* ctx, int32_t leaf, char* name, int32_t name_cap, int64_t* offset, int32_t* rows,
                   int32_t* cols) {
    if (!ctx || leaf < 0 || leaf >= (int32_t)ctx->leaves.size()) { set_error("leaf index out of range"); return VAEK_ERR_INVALID; }
    const Leaf& l = ctx->leaves[leaf];
    if (name && name_cap > 0) { strncpy(name, l.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = l.offset;
    if (rows) *rows = l.rows;
    if (cols) *cols = l.cols;
    return VAEK_OK;
}
int vaek_workspace_bytes(const vaek_ctx* ctx, size_t* bytes) {
    if (!ctx || !bytes) { set_error("null argument"); return VAEK_ERR_INVALID; }
    *bytes = ctx->ws_total;
    return VAEK_OK;
}
int vaek_uses_fused_path(const vaek_ctx* ctx, int32_t* fused) {
    if (!ctx || !fused) { set_error("null argument"); return VAEK_ERR_INVALID; }
    *fused = ctx->fused ? 1 : 0;
    return VAEK_OK;
}

// ---- building blocks ------------------------------------------------------------------------
int vaek_dense_fwd(vaek_ctx* ctx, const float* x, const float* w, const float* b, float* y, int32_t rows,
                   int32_t n_in, int32_t n_out, int32_t act, void* stream) {
    if (!ctx || !x || !w || !y || rows <= 0 || n_in <= 0 || n_out <= 0 || (act != VAEK_ACT_NONE && act != VAEK_ACT_RELU)) {
        set_error("vaek_dense_fwd: invalid argument");
        return VAEK_ERR_INVALID;
    }
    return launch_dense_fwd(x, w, b, y, rows, n_in, n_out, act == VAEK_ACT_RELU, (hipStream_t)stream);
}

int vaek_dense_fwd_reparam(vaek_ctx* ctx, const float* x, const float* w, const float* b, float* mu, float* samples, const float* z1,
                           const float* logvar_e, int32_t rows, int32_t n_in, int32_t n_out, void* stream) {
    if (!ctx || !x || !w || !mu || !samples || !z1 || !logvar_e || rows <= 0 || n_in <= 0 || n_out <= 0) {
        set_error("vaek_dense_fwd_reparam: invalid argument");
        return VAEK_ERR_INVALID;
    }
    return launch_dense_fwd_reparam(x, w, b, mu, samples, z1, logvar_e, rows, n_in, n_out, (hipStream_t)stream);
}

// d logvar_e from the per-split sums of d_samples * z1 (fixed order) and the closed-form KL part
__global__ void reparam_finish_kernel(const float* partial, int S, int L, const float* lv, float rows_over_bt, float* out) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    float acc = 0.f;
    for (int s2 = 0; s2 < S; ++s2) acc += partial[(long long)s2 * L + l];
    const float v = lv[l];
    out[l] = 0.5f * expf(0.5f * v) * acc - 0.5f * (1.f - expf(v)) * rows_over_bt;
}

int vaek_reparam_bwd(vaek_ctx* ctx, float* d_samples, const float* mu, const float* z1, const float* logvar_e, float* d_logvar_e,
                     int32_t rows, int32_t latent_dim, int64_t batch_total, void* workspace, void* stream) {
    if (!ctx || !d_samples || !mu || !z1 || !logvar_e || !d_logvar_e || rows <= 0 || latent_dim <= 0 || latent_dim > 256) {
        set_error("vaek_reparam_bwd: invalid argument");
        return VAEK_ERR_INVALID;
    }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    const double bt = batch_total > 0 ? (double)batch_total : (double)rows;
    const int S = std::max(1, std::min(ctx->Se, (int)(((size_t)ctx->Se * ctx->L) / latent_dim)));     // what the context's partial area holds
    const int rps = (rows + S - 1) / S;
    const int Su = (rows + rps - 1) / rps;
    float* part = at<float>(workspace, ctx->ws_rpart);
    hipStream_t st = (hipStream_t)stream;
    if ((rc = launch_reparam_bwd(d_samples, mu, z1, part, rows, latent_dim, Su, rps, (float)(1.0 / bt), st))) return rc;
    hipLaunchKernelGGL(reparam_finish_kernel, dim3((latent_dim + 63) / 64), dim3(64), 0, st, (const float*)part, Su, (int)latent_dim, logvar_e,
                       (float)((double)rows / bt), d_logvar_e);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int vaek_dense_bwd_dx(vaek_ctx* ctx, const float* dy, const float* w, const float* x_post, float* dx, int32_t rows,
                      int32_t n_in, int32_t n_out, int32_t act, int32_t accumulate, void* stream) {
    if (!ctx || !dy || !w || !dx || rows <= 0 || n_in <= 0 || n_out <= 0 || (act == VAEK_ACT_RELU && !x_post) ||
        (act != VAEK_ACT_NONE && act != VAEK_ACT_RELU)) {
        set_error("vaek_dense_bwd_dx: invalid argument");
        return VAEK_ERR_INVALID;
    }
    return launch_dense_bwd_dx(dy, w, x_post, dx, rows, n_in, n_out, act == VAEK_ACT_RELU, accumulate != 0,
                               (hipStream_t)stream);
}

int vaek_dense_bwd_dw(vaek_ctx* ctx, const float* x, const float* dy, float* dwb, int32_t rows, int32_t n_in,
                      int32_t n_out, void* workspace, void* stream) {
    if (!ctx || !x || !dy || !dwb || rows <= 0 || n_in <= 0 || n_out <= 0) {
        set_error("vaek_dense_bwd_dw: invalid argument");
        return VAEK_ERR_INVALID;
    }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    const int64_t n = (int64_t)(n_in + 1) * n_out;
    const int rps = std::max(64, (int)align_up((size_t)(rows + ctx->S - 1) / ctx->S, 64));
    const int S = (rows + rps - 1) / rps;
    if ((size_t)S * n * sizeof(float) > (size_t)ctx->S * ctx->P * sizeof(float)) {
        set_error("vaek_dense_bwd_dw: layer (%d+1)x%d does not fit this context's slab workspace", n_in, n_out);
        return VAEK_ERR_WORKSPACE;
    }
    float* slabs = at<float>(workspace, ctx->ws_slabs);
    if ((rc = launch_dense_bwd_dw(x, dy, slabs, n, S, rps, rows, n_in, n_out, (hipStream_t)stream))) return rc;
    return launch_sum_slabs(slabs, n, S, dwb, n, (hipStream_t)stream);
}

static int elbo_fwd_bwd_impl(vaek_ctx* ctx, const float* x, const float* x_hat_lin, const float* x_hat_sig, const float* z2,
                             const float* mu, const float* logvar_e, const float* eps_param, float eps, float* d_lin, float* d_sig, float* out4,
                             int32_t rows, int32_t data_dim, int32_t latent_dim, int64_t batch_total, void* workspace,
                             void* stream) {
    if (!ctx || !x || !x_hat_lin || !z2 || !mu || !logvar_e || !out4 || rows <= 0 || data_dim <= 0 || latent_dim <= 0 ||
        (d_lin && x_hat_sig && !d_sig)) {
        set_error("vaek_elbo_fwd_bwd: invalid argument");
        return VAEK_ERR_INVALID;
    }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    // the block entry point splits the rows finely (its own partial area): the train step's splits are whole GEMM tiles of >= 128 rows,
    // which at a few thousand rows of a wide model is a few dozen workgroups (the conv VAE's 4 096 x 4 096 ELBO: 32 workgroups, 0.32 ms)
    if (rows > ctx->B) { set_error("vaek_elbo_fwd_bwd: rows %d exceed this context's batch %d", rows, ctx->B); return VAEK_ERR_WORKSPACE; }
    const int rpe = std::max(1, (rows + kElboBlockSplits - 1) / kElboBlockSplits);
    const int Se = (rows + rpe - 1) / rpe;
    const int64_t bt = batch_total > 0 ? batch_total : rows;
    ElboArgs e{};
    e.x = x; e.y_lin = x_hat_lin; e.y_sig = x_hat_sig; e.z2 = z2; e.mu = mu;
    e.eps_param = eps_param; e.eps_cli = eps; e.d_lin = d_lin; e.d_sig = d_sig;
    e.partial = at<float>(workspace, ctx->ws_epart_blk);
    e.rows = rows; e.D = data_dim; e.L = latent_dim; e.S = Se; e.rows_per_split = rpe;
    e.inv_bt = (float)(1.0 / (double)bt); e.step_dev = nullptr;
    if ((rc = launch_elbo(e, (hipStream_t)stream))) return rc;
    return launch_elbo_out4(e.partial, Se, logvar_e, latent_dim, data_dim, eps_param, eps, (float)rows, e.inv_bt, out4,
                            (hipStream_t)stream);
}

int vaek_elbo_fwd_bwd(vaek_ctx* ctx, const float* x, const float* x_hat_lin, const float* x_hat_sig, const float* z2,
                      const float* mu, const float* logvar_e, float eps, float* d_lin, float* d_sig, float* out4,
                      int32_t rows, int32_t data_dim, int32_t latent_dim, int64_t batch_total, void* workspace,
                      void* stream) {
    return elbo_fwd_bwd_impl(ctx, x, x_hat_lin, x_hat_sig, z2, mu, logvar_e, nullptr, eps, d_lin, d_sig, out4, rows, data_dim, latent_dim,
                             batch_total, workspace, stream);
}

int vaek_elbo_fwd_bwd_dev(vaek_ctx* ctx, const float* x, const float* x_hat_lin, const float* x_hat_sig, const float* z2,
                          const float* mu, const float* logvar_e, const float* eps_param_dev, float eps_scale, float* d_lin, float* d_sig,
                          float* out4, int32_t rows, int32_t data_dim, int32_t latent_dim, int64_t batch_total, void* workspace,
                          void* stream) {
    if (!eps_param_dev) { set_error("vaek_elbo_fwd_bwd_dev: invalid argument"); return VAEK_ERR_INVALID; }
    return elbo_fwd_bwd_impl(ctx, x, x_hat_lin, x_hat_sig, z2, mu, logvar_e, eps_param_dev, eps_scale, d_lin, d_sig, out4, rows, data_dim,
                             latent_dim, batch_total, workspace, stream);
}

int vaek_adam_step(vaek_ctx* ctx, float* params, const float* grads, float* m, float* v, int64_t n, float lr,
                   int32_t step, const int32_t* step_dev, float grad_scale, void* stream) {
    if (!ctx || !params || !grads || !m || !v || n < 0 || (!step_dev && step < 1)) {
        set_error("vaek_adam_step: invalid argument");
        return VAEK_ERR_INVALID;
    }
    return launch_adam(params, grads, m, v, n, lr, step, step_dev, grad_scale, (hipStream_t)stream);
}

// ---- the hot path ----------------------------------------------------------------------------
int vaek_train_step_grads_only(vaek_ctx* ctx, const float* params, float* grads, int32_t* step_dev, const float* x,
                               const float* z1, const float* z2, void* workspace, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !grads || !step_dev || !x || !z1 || !z2) { set_error("vaek_train_step_grads_only: null argument"); return VAEK_ERR_INVALID; }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (ctx->fused)
        return fused_train_step(ctx, const_cast<float*>(params), grads, nullptr, nullptr, step_dev, x, z1, z2, 0.f, false,
                                false, workspace, st);
    if ((rc = generic_grads(ctx, params, step_dev, x, z1, z2, workspace, st))) return rc;
    return generic_finalize(ctx, params, grads, nullptr, nullptr, nullptr, nullptr, 0.f, workspace, st);
}

int vaek_train_step_apply(vaek_ctx* ctx, float* params, const float* grads, float* m, float* v,
                          const int32_t* step_dev, float lr, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !grads || !m || !v || !step_dev) { set_error("vaek_train_step_apply: null argument"); return VAEK_ERR_INVALID; }
    return launch_adam(params, grads, m, v, ctx->P, lr, 0, step_dev, 1.f, (hipStream_t)stream);
}

int vaek_train_step(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev, const float* x,
                    const float* z1, const float* z2, float lr, void* workspace, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !grads || !m || !v || !step_dev || !x || !z1 || !z2) { set_error("vaek_train_step: null argument"); return VAEK_ERR_INVALID; }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (ctx->cfg.world > 1) {
        if (!ctx->comm.ready) {
            set_error("vaek_train_step: world=%d but no communicator; call vaek_comm_init or use "
                      "vaek_train_step_grads_only + all-reduce + vaek_train_step_apply", ctx->cfg.world);
            return VAEK_ERR_COMM;
        }
        // fused path: the exchange happens inside the finalize kernel (the step stays two launches)
        if (ctx->fused) return fused_train_step(ctx, params, grads, m, v, step_dev, x, z1, z2, lr, true, true, workspace, st);
        if ((rc = vaek_train_step_grads_only(ctx, params, grads, step_dev, x, z1, z2, workspace, stream))) return rc;
        if ((rc = vaek_comm_allreduce(ctx, grads, ctx->P + kExtra, stream))) return rc;
        return vaek_train_step_apply(ctx, params, grads, m, v, step_dev, lr, stream);
    }
    if (ctx->fused) return fused_train_step(ctx, params, grads, m, v, step_dev, x, z1, z2, lr, true, false, workspace, st);
    if ((rc = generic_grads(ctx, params, step_dev, x, z1, z2, workspace, st))) return rc;
    return generic_finalize(ctx, params, grads, params, m, v, step_dev, lr, workspace, st);
}

// Train on the batch (x, z1, z2) AND draw the next one (x_next, z1_next, z2_next; none may alias the current batch)
// from the self-advancing generator counter: see include/vaek.h.
int vaek_train_step_gen(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev, const float* x,
                        const float* z1, const float* z2, float lr, void* workspace, int32_t kind, const float* A, int32_t dd,
                        int32_t did, int32_t pad, float var_added, float* x_next, float* z1_next, float* z2_next, int64_t row0,
                        uint64_t seed, int32_t* counter, int32_t which, uint32_t tag, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !grads || !m || !v || !step_dev || !x || !z1 || !z2 || !x_next || !z1_next || !z2_next || !counter) {
        set_error("vaek_train_step_gen: null argument");
        return VAEK_ERR_INVALID;
    }
    if (x_next == x || z1_next == z1 || z2_next == z2) {
        set_error("vaek_train_step_gen: the next batch must not alias the current one");
        return VAEK_ERR_INVALID;
    }
    BatchArgs gen;
    int rc = make_batch_args(ctx, kind, A, dd, did, pad, var_added, x_next, z1_next, z2_next, ctx->B, row0, seed, nullptr, 0, counter,
                             which, tag, &gen);
    if (rc) return rc;
    if ((rc = check_ws(ctx, workspace))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool comm_ok = ctx->cfg.world == 1 || ctx->comm.ready;
    if (ctx->fused && comm_ok)
        return fused_train_step(ctx, params, grads, m, v, step_dev, x, z1, z2, lr, true, ctx->cfg.world > 1, workspace, st, &gen);
    // layer-by-layer path: the same two operations as separate launches on the one stream
    if ((rc = make_batch_launch(ctx, gen, st))) return rc;
    return vaek_train_step(ctx, params, grads, m, v, step_dev, x, z1, z2, lr, workspace, stream);
}

int vaek_supports_train_steps(const vaek_ctx* ctx, int32_t* yes) {
    if (!ctx || !yes) { set_error("null argument"); return VAEK_ERR_INVALID; }
    *yes = lin_steps_supported(ctx) ? 1 : 0;
    return VAEK_OK;
}

int vaek_train_steps(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev, const float* const* xs,
                     const float* const* z1s, const float* const* z2s, int32_t n_steps, float lr, void* workspace, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !grads || !m || !v || !step_dev || !xs || !z1s || !z2s || n_steps < 0) { set_error("vaek_train_steps: invalid argument"); return VAEK_ERR_INVALID; }
    if (!lin_steps_supported(ctx)) {
        set_error("vaek_train_steps: this context is not a single-GPU float32 linear VAE with L + 2 D + 1 <= 64 (use vaek_train_step)");
        return VAEK_ERR_INVALID;
    }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    for (int i = 0; i < n_steps; ++i)
        if (!xs[i] || !z1s[i] || !z2s[i] || ((reinterpret_cast<uintptr_t>(xs[i]) | reinterpret_cast<uintptr_t>(z1s[i]) | reinterpret_cast<uintptr_t>(z2s[i])) & 15)) {
            set_error("vaek_train_steps: batch %d has a null or not 16-byte aligned pointer", i);
            return VAEK_ERR_INVALID;
        }
    if (n_steps == 0) return VAEK_OK;
    return lin_train_steps(ctx, params, grads, m, v, step_dev, xs, z1s, z2s, n_steps, lr, workspace, (hipStream_t)stream);
}

int vaek_supports_train_steps_gen(const vaek_ctx* ctx, int32_t kind, int32_t* yes) {
    if (!ctx || !yes) { set_error("null argument"); return VAEK_ERR_INVALID; }
    *yes = lin_steps_gen_supported(ctx, kind) ? 1 : 0;
    return VAEK_OK;
}

int vaek_train_steps_gen(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev, int32_t kind, const float* A,
                         int32_t dd, int32_t did, int32_t pad, float var_added, int64_t row0, uint64_t seed, uint32_t tag, int32_t n_steps,
                         float lr, void* workspace, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !grads || !m || !v || !step_dev || n_steps < 0) { set_error("vaek_train_steps_gen: invalid argument"); return VAEK_ERR_INVALID; }
    if (!lin_steps_gen_supported(ctx, kind)) {
        set_error("vaek_train_steps_gen: needs a context vaek_train_steps' persistent form covers and dataset kind 0 or 2 (use vaek_train_step_gen)");
        return VAEK_ERR_INVALID;
    }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    BatchArgs gen;
    // (the generator's own output pointers stay unused: a dummy non-null x / z pair passes its argument check)
    float* dummy = reinterpret_cast<float*>(workspace);
    if ((rc = make_batch_args(ctx, kind, A, dd, did, pad, var_added, dummy, dummy, dummy, ctx->B, row0, seed, step_dev, 0, nullptr, 0, tag, &gen))) return rc;
    gen.x = gen.z1 = gen.z2 = nullptr;
    if (n_steps == 0) return VAEK_OK;
    return lin_train_steps_gen(ctx, params, grads, m, v, step_dev, gen, n_steps, lr, workspace, (hipStream_t)stream);
}

int vaek_train_steps_moment_len(const vaek_ctx* ctx, int64_t* len) {
    if (!ctx || !len) { set_error("null argument"); return VAEK_ERR_INVALID; }
    *len = (int64_t)lin_moment_len(ctx);
    return VAEK_OK;
}

int vaek_train_steps_moments(vaek_ctx* ctx, const float* x, const float* z1, const float* z2, double* M, void* workspace, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !x || !z1 || !z2 || !M) { set_error("vaek_train_steps_moments: null argument"); return VAEK_ERR_INVALID; }
    if (!lin_moments_supported(ctx)) { set_error("vaek_train_steps_moments: not a float32 linear VAE with L + 2 D + 1 <= 64"); return VAEK_ERR_INVALID; }
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(z1) | reinterpret_cast<uintptr_t>(z2)) & 15) {
        set_error("vaek_train_steps_moments: batch pointers must be 16-byte aligned");
        return VAEK_ERR_INVALID;
    }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    return lin_moments(ctx, x, z1, z2, M, workspace, (hipStream_t)stream);
}

int vaek_train_steps_update(vaek_ctx* ctx, float* params, float* grads, float* m, float* v, int32_t* step_dev, const double* M, float lr,
                            void* workspace, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !grads || !m || !v || !step_dev || !M) { set_error("vaek_train_steps_update: null argument"); return VAEK_ERR_INVALID; }
    if (!lin_moments_supported(ctx)) { set_error("vaek_train_steps_update: not a float32 linear VAE with L + 2 D + 1 <= 64"); return VAEK_ERR_INVALID; }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    return lin_update(ctx, params, grads, m, v, step_dev, M, lr, (hipStream_t)stream);
}

int vaek_train_steps_status(vaek_ctx* ctx, void* workspace, int32_t* gave_up) {
    if (!ctx || !workspace || !gave_up) { set_error("null argument"); return VAEK_ERR_INVALID; }
    int g = 0;
    const int rc = lin_steps_status(ctx, workspace, &g);
    *gave_up = g;
    return rc;
}

// buckets in the order the backward pass completes them: Decoder (last layer first), SigDecoder, Encoder, tail
static void bucket_list(const vaek_ctx* c, std::vector<std::pair<int64_t, int64_t>>& out) {
    out.clear();
    if (c->fused) { out.push_back({0, c->P + kExtra}); return; }
    for (const Net* net : {&c->dec, &c->sig, &c->enc})
        for (int i = (int)net->layers.size() - 1; i >= 0; --i) {
            const auto& l = net->layers[i];
            out.push_back({l.w_off, (int64_t)(l.n_in + 1) * l.n_out});
        }
    out.push_back({c->off_epsp, c->P + kExtra - c->off_epsp});
}

int vaek_bucket_count(const vaek_ctx* ctx, int32_t* n) {
    if (!ctx || !n) { set_error("null argument"); return VAEK_ERR_INVALID; }
    std::vector<std::pair<int64_t, int64_t>> b;
    bucket_list(ctx, b);
    *n = (int32_t)b.size();
    return VAEK_OK;
}

int vaek_bucket_info(const vaek_ctx* ctx, int32_t i, int64_t* offset, int64_t* count) {
    if (!ctx || !offset || !count) { set_error("null argument"); return VAEK_ERR_INVALID; }
    std::vector<std::pair<int64_t, int64_t>> b;
    bucket_list(ctx, b);
    if (i < 0 || i >= (int32_t)b.size()) { set_error("bucket index out of range"); return VAEK_ERR_INVALID; }
    *offset = b[i].first; *count = b[i].second;
    return VAEK_OK;
}

int vaek_train_step_grads_bucketed(vaek_ctx* ctx, const float* params, float* grads, int32_t* step_dev, const float* x,
                                   const float* z1, const float* z2, void* const* ready_events, void* workspace, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !grads || !step_dev || !x || !z1 || !z2 || !ready_events) { set_error("vaek_train_step_grads_bucketed: null argument"); return VAEK_ERR_INVALID; }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (ctx->fused) {
        if ((rc = fused_train_step(ctx, const_cast<float*>(params), grads, nullptr, nullptr, step_dev, x, z1, z2, 0.f, false, false,
                                   workspace, st)))
            return rc;
        VAEK_HIP_CHECK(hipEventRecord((hipEvent_t)ready_events[0], st));
        return VAEK_OK;
    }
    BucketSink sink{grads, ready_events, 0};
    if ((rc = generic_grads(ctx, params, step_dev, x, z1, z2, workspace, st, &sink))) return rc;
    // tail: epsilon_p, epsilon and the loss slots (everything the layer buckets did not cover)
    if ((rc = generic_finalize(ctx, params, grads, nullptr, nullptr, nullptr, nullptr, 0.f, workspace, st, ctx->off_epsp))) return rc;
    VAEK_HIP_CHECK(hipEventRecord((hipEvent_t)ready_events[sink.next], st));
    return VAEK_OK;
}

int vaek_loss_eval(vaek_ctx* ctx, const float* params, const float* x, const float* z1, const float* z2, float* out4,
                   void* workspace, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !x || !z1 || !z2 || !out4) { set_error("vaek_loss_eval: null argument"); return VAEK_ERR_INVALID; }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    vaek_ctx* c = ctx;
    const bool sig = c->cfg.sigmoid_decoder != 0;
    if ((rc = convert_weights(c, params, workspace, st))) return rc;
    if ((rc = net_forward(c, c->enc, params, x, workspace, c->B, true, z1, st))) return rc;
    const float* samples = at<float>(workspace, c->ws_samples);
    if ((rc = net_forward(c, c->dec, params, samples, workspace, c->B, false, nullptr, st))) return rc;
    if (sig && (rc = net_forward(c, c->sig, params, samples, workspace, c->B, false, nullptr, st))) return rc;
    ElboArgs e{};
    e.x = x; e.y_lin = at<float>(workspace, c->dec.act_off.back());
    e.y_sig = sig ? at<float>(workspace, c->sig.act_off.back()) : nullptr;
    e.z2 = z2; e.mu = at<float>(workspace, c->enc.act_off.back());
    e.eps_param = c->off_eps >= 0 ? params + c->off_eps : nullptr; e.eps_cli = c->cfg.eps_cli;
    e.partial = at<float>(workspace, c->ws_epart);
    e.rows = c->B; e.D = c->D; e.L = c->L; e.S = c->Se; e.rows_per_split = c->rows_per_esplit;
    e.inv_bt = (float)(1.0 / (double)c->B);     // VAE.loss is a plain mean over the eval batch
    if ((rc = launch_elbo(e, st))) return rc;
    return launch_eval_out4(e.partial, c->Se, params, c->off_epsp, c->off_eps, c->L, c->D, c->cfg.eps_cli, (float)c->B,
                            e.inv_bt, out4, st);
}

int vaek_forward(vaek_ctx* ctx, const float* params, const float* x, const float* z1, const float* z2,
                 int32_t sampling, float eps, float* x_hat, float* mu_out, int32_t rows, void* workspace, void* stream) {
    ProfBind pb(ctx);
    if (!ctx || !params || !z1 || !z2 || !x_hat || (!sampling && !x) || rows <= 0) { set_error("vaek_forward: invalid argument"); return VAEK_ERR_INVALID; }
    if (rows > ctx->B) { set_error("vaek_forward: rows %d exceed this context's batch %d", rows, ctx->B); return VAEK_ERR_WORKSPACE; }
    int rc = check_ws(ctx, workspace);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    vaek_ctx* c = ctx;
    const bool sig = c->cfg.sigmoid_decoder != 0;
    const float* samples = z1;                      // sampling: mu = 0, logvar_e = 0 -> samples = z1
    if ((rc = convert_weights(c, params, workspace, st))) return rc;
    if (!sampling) {
        if ((rc = net_forward(c, c->enc, params, x, workspace, rows, true, z1, st))) return rc;
        samples = at<float>(workspace, c->ws_samples);
        if (mu_out)
            VAEK_HIP_CHECK(hipMemcpyAsync(mu_out, at<float>(workspace, c->enc.act_off.back()),
                                          (size_t)rows * c->L * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (mu_out) {
        VAEK_HIP_CHECK(hipMemsetAsync(mu_out, 0, (size_t)rows * c->L * sizeof(float), st));
    }
    if ((rc = net_forward(c, c->dec, params, samples, workspace, rows, false, nullptr, st))) return rc;
    if (sig && (rc = net_forward(c, c->sig, params, samples, workspace, rows, false, nullptr, st))) return rc;
    const float* eps_param = (!sampling && c->off_eps >= 0) ? params + c->off_eps : nullptr;
    const float eps_val = sampling ? eps : c->cfg.eps_cli;
    return launch_add_noise(at<float>(workspace, c->dec.act_off.back()),
                            sig ? at<float>(workspace, c->sig.act_off.back()) : nullptr, z2, eps_param, eps_val, x_hat,
                            (int64_t)rows * c->D, st);
}

// Diagnostic hook (not part of include/vaek.h): device buffer for the in-kernel s_memtime stamps of
// a -DVAEK_STAMPS build (tools/stamps.sh).  No effect in the shipped build.
int vaek_debug_set_stamps(vaek_ctx* ctx, unsigned long long* buf) {
    if (!ctx) return VAEK_ERR_INVALID;
    ctx->dbg_stamps = buf;
    return VAEK_OK;
}

int vaek_profile_begin(vaek_ctx* ctx, int32_t max_records) {
    if (!ctx || max_records <= 0) { set_error("vaek_profile_begin: invalid argument"); return VAEK_ERR_INVALID; }
    Profiler& p = ctx->prof;
    while ((int)p.ev.size() < 2 * max_records) {
        hipEvent_t e;
        VAEK_HIP_CHECK(hipEventCreate(&e));
        p.ev.push_back(e);
    }
    p.label.assign(max_records, nullptr);
    p.cap = max_records; p.n = 0; p.on = true;
    return VAEK_OK;
}

int vaek_profile_report(vaek_ctx* ctx, char* buf, size_t cap) {
    if (!ctx || !buf || cap < 8) { set_error("vaek_profile_report: invalid argument"); return VAEK_ERR_INVALID; }
    Profiler& p = ctx->prof;
    p.on = false;
    std::vector<std::string> names;
    std::vector<double> total;
    std::vector<int> count;
    for (int i = 0; i < p.n; ++i) {
        VAEK_HIP_CHECK(hipEventSynchronize(p.ev[2 * i + 1]));
        float ms = 0.f;
        VAEK_HIP_CHECK(hipEventElapsedTime(&ms, p.ev[2 * i], p.ev[2 * i + 1]));
        size_t k = 0;
        for (; k < names.size(); ++k) if (names[k] == p.label[i]) break;
        if (k == names.size()) { names.push_back(p.label[i]); total.push_back(0); count.push_back(0); }
        total[k] += ms; count[k] += 1;
    }
    std::string js = "{";
    for (size_t k = 0; k < names.size(); ++k) {
        char tmp[160];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"count\": %d, \"total_ms\": %.6f}", k ? ", " : "", names[k].c_str(), count[k], total[k]);
        js += tmp;
    }
    js += "}";
    if (js.size() + 1 > cap) { set_error("vaek_profile_report: buffer too small"); return VAEK_ERR_INVALID; }
    memcpy(buf, js.c_str(), js.size() + 1);
    p.n = 0;
    return VAEK_OK;
}

}  // extern "C"
