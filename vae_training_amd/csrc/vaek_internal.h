// Internal declarations shared by the translation units of libvaek.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/vaek.h"

namespace vaek {

constexpr int kExtra = 4;          // grads[P..P+3] = loss, mean Dkl, mean mse, 0
constexpr float kLog2Pi = 1.8378770664093453f;
constexpr float kAdamB1 = 0.9f, kAdamB2 = 0.999f, kAdamEps = 1e-8f;

void set_error(const char* fmt, ...);
#define VAEK_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            vaek::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return VAEK_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

struct Leaf {
    std::string name;
    int64_t offset;
    int rows, cols;   // bias / epsilon_p / epsilon: rows == 1
};

struct Layer {
    int n_in, n_out;
    int64_t w_off;    // kernel offset in the flat buffers; bias follows at w_off + n_in*n_out
    bool relu;        // relu after this layer (every layer but the last, networks.py:35-39)
    int S, rows_per_split;   // batch split of this layer's dW|db GEMM (skinny layers get more, smaller splits)
    bool sk = false;         // first / last layer of a bf16-storage stack on the skinny kernels (gemm_skinny16.hip)
    int64_t sk_off = 0;      // element offset of its padded bf16 kernel copy in ws_sk16
};

struct Net {
    std::vector<Layer> layers;
    // workspace byte offsets of each layer's OUTPUT activation [B, n_out] (float32; bf16 for the hidden layers of a b16 net)
    std::vector<size_t> act_off;
    // bf16-storage mode (dtype = VAEK_BF16, >= 2 hidden layers, every hidden width a multiple of 64): hidden activations
    // and gradients are bf16; layer 0 and the last layer run the exact f32 kernel with one bf16-stored side, the layers in
    // between gemm_bf16s.hip.  wb_off[i]: element offset of layer i's bf16 kernel copies (W, then W^T) in ws_wb16.
    bool b16 = false;
    std::vector<int64_t> wb_off;
};

// Event-pair pool for vaek_profile_*: one pair per kernel launch while enabled.
struct Profiler {
    bool on = false;
    std::vector<hipEvent_t> ev;          // 2 * max_records
    std::vector<const char*> label;      // per record
    int n = 0, cap = 0;
};
extern thread_local Profiler* g_prof;    // set by the C entry points for the duration of a call
// One launch = one event pair, handed to hipExtLaunchKernelGGL so the timestamps are the kernel's own
// begin/end (what rocprofv3 reports), not host-side record times (which would count submission gaps
// into the ~5-10 us kernels of the metric workload).
struct ProfScope {
    int idx = -1;
    ProfScope(const char* label, hipStream_t) {
        Profiler* p = g_prof;
        if (p && p->on && p->n < p->cap) { idx = p->n++; p->label[idx] = label; }
    }
    hipEvent_t start() const { return g_prof->ev[2 * idx]; }
    hipEvent_t stop() const { return g_prof->ev[2 * idx + 1]; }
};
}  // namespace vaek
#include <hip/hip_ext.h>
namespace vaek {
template <typename K, typename... Args>
inline void launch_k(const ProfScope& ps, K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, Args... args) {
    if (ps.idx >= 0) hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, ps.start(), ps.stop(), 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, shmem, st, args...);
}

struct Comm {
    bool ready = false;
    void* local = nullptr;      // uncached granule buffer of this rank (hipExtMallocWithFlags)
    size_t bytes = 0;
    int ng = 0;                 // granules per (bank, source rank)
    std::vector<void*> peers;   // world entries; peers[rank] == local, others IPC-mapped
    uint32_t epoch = 0;         // stand-alone all-reduce epochs (region 1)
    size_t lin_off = 0, lin_bytes = 0;   // the moment-exchange region of vaek_train_steps (linear_moments.hip), behind the status line
};

}  // namespace vaek

struct vaek_ctx {
    vaek_config cfg;
    int B, D, L;
    int64_t Bt;                      // global batch (divisor of the mean)
    int64_t P;                       // trainable floats
    std::vector<vaek::Leaf> leaves;
    vaek::Net enc, dec, sig;         // sig empty unless cfg.sigmoid_decoder
    int64_t off_epsp, off_eps;       // flat offsets; off_eps = -1 without -tdv
    // batch split of the deterministic reductions: S slabs for the dW|db GEMMs, Se for elementwise
    int S, rows_per_split, Se, rows_per_esplit;
    bool fused;                      // fused small-model path available and selected
    // workspace layout (bytes)
    size_t ws_samples, ws_dsamp, ws_gbuf0, ws_gbuf1, ws_slabs, ws_epart, ws_epart_blk, ws_rpart, ws_eblk, ws_fused, ws_wb16, ws_sk16, ws_skpart, ws_lin, ws_lwd, ws_total;
    bool lwd = false; int lwd_rb = 0;           // wide linear decoder: fused forward / ELBO / backward (linear_wide.hip), rows per row block
    int max_width;
    int n_cu;
    vaek::Comm comm;
    vaek::Profiler prof;
    unsigned long long* dbg_stamps = nullptr;   // diagnostic builds (-DVAEK_STAMPS) only
    float* loss_hist = nullptr;                 // optional device ring: loss of Adam step t -> [(t-1) % cap]
    int64_t loss_hist_cap = 0;
    bool lin_ws_reinit = false;                 // ... and must be zeroed again before the next launch (a wait gave up; a diagnostic launch without updater)
    void* lin_ws_inited = nullptr;              // workspace whose vaek_train_steps arrival counters have been zeroed (linear_moments.hip)
};

namespace vaek {

// ---- gemm_f32.hip -------------------------------------------------------------------------
int launch_dense_fwd(const float* x, const float* w, const float* b, float* y, int rows, int n_in,
                     int n_out, bool relu, hipStream_t st);
// y = mu (stored), samples = mu + exp(lv/2) * z1
int launch_dense_fwd_reparam(const float* x, const float* w, const float* b, float* mu, float* samples,
                             const float* z1, const float* lv, int rows, int n_in, int n_out, hipStream_t st);
int launch_dense_bwd_dx(const float* dy, const float* w, const float* x_post, float* dx, int rows,
                        int n_in, int n_out, bool relu, bool accumulate, hipStream_t st);
// writes S slabs: slab[s][row*n_out + col], row in [0, n_in] (row n_in = bias), stride in floats
int launch_dense_bwd_dw(const float* x, const float* dy, float* slab0, int64_t slab_stride, int S,
                        int rows_per_split, int rows, int n_in, int n_out, hipStream_t st);

// ---- gemm_bf16.hip: the same four launchers with bf16 matrix-core arithmetic (f32 storage) -----------
int launch_dense_fwd_bf16(const float* x, const float* w, const float* b, float* y, int rows, int n_in, int n_out,
                          bool relu, hipStream_t st);
int launch_dense_fwd_reparam_bf16(const float* x, const float* w, const float* b, float* mu, float* samples,
                                  const float* z1, const float* lv, int rows, int n_in, int n_out, hipStream_t st);
int launch_dense_bwd_dx_bf16(const float* dy, const float* w, const float* x_post, float* dx, int rows, int n_in,
                             int n_out, bool relu, bool accumulate, hipStream_t st);
int launch_dense_bwd_dw_bf16(const float* x, const float* dy, float* slab0, int64_t slab_stride, int S, int rows_per_split,
                             int rows, int n_in, int n_out, hipStream_t st);

// ---- bf16-STORAGE mode (hidden activations / gradients kept as bf16 in HBM) ------------------------------------------
// gemm_f32.hip: the skinny first / last layer of a stack on the exact f32 kernel, hidden-side operand stored as bf16
int launch_dense_fwd_out16(const float* x, const float* w, const float* b, __bf16* y, int rows, int n_in, int n_out, bool relu,
                           hipStream_t st);
int launch_dense_fwd_in16(const __bf16* x, const float* w, const float* b, float* y, int rows, int n_in, int n_out, hipStream_t st);
int launch_dense_fwd_reparam_in16(const __bf16* x, const float* w, const float* b, float* mu, float* samples, const float* z1,
                                  const float* lv, int rows, int n_in, int n_out, hipStream_t st);
int launch_dense_fwd_elbo_in16(const __bf16* h, const float* w, const float* b, float* d_out, const float* x, const float* z2,
                               const float* eps_param, float eps_cli, float inv_bt, float* part, int rows, int n_in, int n_out,
                               int* bm, int* nbx, hipStream_t st);
int launch_dense_bwd_dx_out16(const float* dy, const float* w, const __bf16* x_post, __bf16* dx, int rows, int n_in, int n_out,
                              bool accumulate, hipStream_t st);
int launch_dense_bwd_dw_x16(const __bf16* x, const float* dy, float* slab0, int64_t slab_stride, int S, int rows_per_split,
                            int rows, int n_in, int n_out, hipStream_t st);
int launch_dense_bwd_dw_dy16(const float* x, const __bf16* dy, float* slab0, int64_t slab_stride, int S, int rows_per_split,
                             int rows, int n_in, int n_out, hipStream_t st);
int launch_dense_bwd_dx_in16(const __bf16* dy, const float* w, float* dx, int rows, int n_in, int n_out, bool accumulate,
                             hipStream_t st);
// gemm_bf16s.hip: the wide hidden -> hidden layers, bf16 in / bf16 out on v_mfma_f32_32x32x16_bf16
int launch_hs_fwd(const __bf16* x, const __bf16* wT, const float* b, __bf16* y, int rows, int n_in, int n_out, bool relu,
                  hipStream_t st);
int launch_hs_dx(const __bf16* dy, const __bf16* w16, const __bf16* x_post, __bf16* dx, int rows, int n_in, int n_out,
                 hipStream_t st);
int launch_hs_dw(const __bf16* x, const __bf16* dy, float* slab0, int64_t slab_stride, int S, int rows_per_split, int rows,
                 int n_in, int n_out, hipStream_t st);
int launch_cvt_weights(const float* params, __bf16* out, const int* K, const int* N, const int64_t* w_off, const int64_t* out_off,
                       int n, hipStream_t st);

// gemm_skinny16.hip: first (d -> H) / last (H -> d) layer of a bf16-storage stack, d <= 16: one HBM pass per big tensor
bool sk_supported(int d, int H);
size_t sk_partial_bytes(int d, int H, int S);
int launch_sk_first_fwd(const float* x, const float* w, const float* b, __bf16* y, int rows, int d, int H, bool relu, hipStream_t st);
int launch_sk_last_fwd(const __bf16* h, const __bf16* wp, const float* b, float* y, int rows, int H, int d, hipStream_t st);
int launch_sk_last_fwd_reparam(const __bf16* h, const __bf16* wp, const float* b, float* mu, float* samples, const float* z1,
                               const float* lv, int rows, int H, int d, hipStream_t st);
int launch_sk_last_fwd_elbo(const __bf16* h, const __bf16* wp, const float* b, float* d_out, const float* x, const float* z2,
                            const float* eps_param, float eps_cli, float inv_bt, float* part, int rows, int H, int d, int* bm,
                            int* nbx, hipStream_t st);
int launch_sk_first_dx(const __bf16* dy, const __bf16* wp, float* dx, int rows, int H, int d, bool accumulate, hipStream_t st);
int launch_sk_last_bwd(const __bf16* h, const float* dy, const float* w, __bf16* dh, float* partial, float* slab0, int64_t slab_stride,
                       int S, int rows, int H, int d, hipStream_t st);
int launch_sk_first_bwd(const float* x, const __bf16* dy, float* partial, float* slab0, int64_t slab_stride, int S, int rows, int H,
                        int d, hipStream_t st);
int launch_sk_prep(const float* params, __bf16* out, const int* H, const int* d, const int* transposed, const int64_t* w_off,
                   const int64_t* out_off, int n, hipStream_t st);

// ---- elbo.hip -----------------------------------------------------------------------------
struct ElboArgs {
    const float* x; const float* y_lin; const float* y_sig; const float* z2; const float* mu;
    const float* eps_param;   // device (1,) or nullptr
    float eps_cli;            // eps = eps_param ? *eps_param * eps_cli : eps_cli
    float* d_lin; float* d_sig;   // nullptr -> no gradients (eval)
    float* partial;           // [S][4]: +0 = sum mse terms (variable part), +1 = sum mu^2, +2 = d eps sum (variable part)
    int rows, D, L, S, rows_per_split;
    float inv_bt;
    int32_t* step_dev;        // incremented by block 0 (may be nullptr)
};
int launch_elbo(const ElboArgs& a, hipStream_t st);
int launch_elbo_reduce(const float* part, int bm, int nbx, const float* mu, float* partial, int rows, int L, int S,
                       int rows_per_split, int32_t* step_dev, hipStream_t st);
int launch_dense_fwd_elbo(const float* h, const float* w, const float* b, float* d_out, const float* x, const float* z2,
                          const float* eps_param, float eps_cli, float inv_bt, float* part, int rows, int n_in, int n_out,
                          int* bm, int* nbx, hipStream_t st);
// dmu = dsamp + mu * inv_bt (in place on dsamp); partial[s][l] = sum_rows dsamp * z1
int launch_reparam_bwd(float* dsamp, const float* mu, const float* z1, float* partial,
                       int rows, int L, int S, int rows_per_split, float inv_bt, hipStream_t st);
struct FinalizeArgs {
    const float* slabs; int64_t slab_stride; int S;   // dW|db partials in flat-gradient layout (S = max over layers)
    int nseg; int seg_end[32]; int seg_S[32];          // per layer: flat index one past its [kernel|bias], its slab count
    const float* epart; const float* rpart; int Se;   // elbo partials [Se][4], reparam partials [Se][L]
    int64_t P, off_epsp, off_eps; int L, D;
    const float* params;      // for logvar_e / epsilon
    float eps_cli; float rows_over_bt; float inv_bt; float rows;
    float* grads;             // P + 4
    // optional fused Adam (world == 1): params_rw != nullptr
    float* params_rw; float* m; float* v; const int32_t* step_dev; float lr;
    float* loss_hist; long long loss_hist_cap;    // optional: loss of Adam step t -> loss_hist[(t-1) % cap]
    long long lo;             // outputs below lo are left alone (bucketed mode finalises the tail only)
};
int launch_finalize(const FinalizeArgs& a, hipStream_t st);
// streaming slab sum (+ Adam) of the outputs [0, hi); pair with launch_finalize(lo = hi) for the tail
int launch_bulk_finalize(const FinalizeArgs& a, int64_t hi, hipStream_t st);
int launch_adam(float* params, const float* grads, float* m, float* v, int64_t n, float lr, int step,
                const int32_t* step_dev, float grad_scale, hipStream_t st);
// x_hat = y_lin (+ sigmoid(y_sig)) + z2 * exp(eps/2)
int launch_add_noise(const float* y_lin, const float* y_sig, const float* z2, const float* eps_param,
                     float eps_cli, float* x_hat, int64_t n, hipStream_t st);
// stand-alone ELBO finalisation for vaek_elbo_fwd_bwd: out4 = {loss, dkl, mse, d eps}
int launch_elbo_out4(const float* partial, int S, const float* lv, int L, int D, const float* eps_param,
                     float eps, float rows, float inv_bt, float* out4, hipStream_t st);   // eps = eps_param ? eps_param[0] * eps : eps
// eval: out4 = {loss, dkl, mse, eps} from slabs
int launch_eval_out4(const float* partial, int S, const float* params, int64_t off_epsp,
                     int64_t off_eps, int L, int D, float eps_cli, float rows, float inv_bt, float* out4,
                     hipStream_t st);
int launch_sum_slabs(const float* slabs, int64_t stride, int S, float* out, int64_t n, hipStream_t st);
int launch_sum_slabs_inplace(float* slabs, int64_t stride, int S, float* out, int64_t n, hipStream_t st);   // many slabs, few outputs; clobbers the slabs
#ifdef __HIPCC__
int launch_cvt_bf16(const float* src, __bf16* dst, int64_t n, __bf16* zero8, hipStream_t st);
int launch_cvt_bf16_t(const float* src, __bf16* dst, int K, int N, __bf16* zero8, hipStream_t st);
// "Has this been set up on the CURRENT device?" -- hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device, and a per-thread
// flag skipped it for the second device a thread drives (launches needing > 64 KB of LDS then fail with an invalid value).
struct PerDeviceOnce {
    unsigned long long mask[2] = {0ull, 0ull};          // devices 0 .. 127
    static int slot() { int dev = 0; return hipGetDevice(&dev) == hipSuccess ? (dev & 127) : 0; }
    bool need() const { const int d = slot(); return !(mask[d >> 6] >> (d & 63) & 1ull); }
    void mark() { const int d = slot(); mask[d >> 6] |= 1ull << (d & 63); }
};
int launch_hs_conv(int mode, const __bf16* x, const __bf16* wb, const __bf16* zeros, const float* bias, const float* mask, const __bf16* mask16,
                   float* out, __bf16* out16, int batch, int H, int W, int Cin, int Cout, bool relu, hipStream_t st);      // mode 0: forward, 1: transposed
int launch_hs_conv_dw(const __bf16* x, const __bf16* dy, const __bf16* zeros, float* slab0, int64_t slab_stride, int S, int rows_per_split,
                      int batch, int H, int W, int Cin, int Cout, hipStream_t st);
#endif

// ---- comm.hip -------------------------------------------------------------------------------
struct CommDev;
CommDev comm_dev(const vaek_ctx* c, int region);

#ifdef __HIPCC__
// flax.optim.Adam.apply_gradient (networks.py:100) for one parameter; (1 - beta) formed in double, as the oracle does
__device__ __forceinline__ void adam_apply_f(float& p, float g, float& m, float& v, float lr, float bc1, float bc2) {
    m = kAdamB1 * m + (float)(1.0 - 0.9) * g;
    v = kAdamB2 * v + (float)(1.0 - 0.999) * g * g;
    p = p - lr * (m / bc1) / (sqrtf(v / bc2) + kAdamEps);
}
#endif

// ---- fused_small.hip / fused_mfma.hip: arguments of the fused linear-VAE kernels ----------------
struct BatchArgs {
    int kind;                   // 0 linear_gaussian, 1 sigmoid, 2 sphere
    const float* A;             // linear: [dd][did] row-major; sigmoid: [dd]; sphere: unused
    int dd, did, pad; float noise_std;
    float* x; float* z1; float* z2;
    int rows; long long row0; int D, L;
    unsigned long long seed; const int32_t* step_dev; unsigned step_host, tag;
    int32_t* counter; int which; // make_batch_next: step = counter[which]; the launch stores counter[which ^ 1] = step + 1
};

struct FusedArgs {
    const float* x; const float* z1; const float* z2;
    float* partials; int pstride;      // [grid][pstride]
    int B, D, L, ntiles;
    float inv_bt, eps_cli;
    int off_be, off_wd, off_bd, off_ws, off_bs, off_epsp, off_eps, P;
    int32_t* step_dev;
    unsigned long long* stamps;        // diagnostic builds only (-DVAEK_STAMPS): [block][wave][8] s_memtime
    // single-launch step (fused_mfma.hip): the batch fits ONE workgroup, so there is nothing to reduce across workgroups
    // and workgroup 0 finalizes itself -- closed-form terms, loss, Adam, step counter; workgroups 1.. (if any) draw the
    // next batch (vaek_train_step_gen): the reference's loop body at its own batch size is one launch
    int single;
    float* grads; float* params_rw; float* m; float* v; float lr, rows_over_bt, rows;
    float* loss_hist; long long loss_hist_cap;
    int has_gen; BatchArgs gen;
};
bool fused_mfma_supported(const vaek_ctx* c);
int fused_mfma_launch(const vaek_ctx* c, const float* params, const void* fused_args, int grid, hipStream_t st);

// ---- linear_moments.hip: N pipelined steps of a linear VAE through the batch's second-moment matrix --------------------
bool lin_steps_supported(const vaek_ctx* c);
size_t lin_comm_bytes(const vaek_ctx* c);
size_t lin_steps_workspace_bytes(const vaek_ctx* c);
int lin_train_steps(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev, const float* const* xs,
                    const float* const* z1s, const float* const* z2s, int n_steps, float lr, void* ws, hipStream_t st);
int lin_steps_status(vaek_ctx* c, void* ws, int* gave_up);
bool lin_steps_gen_supported(const vaek_ctx* c, int kind);
// ---- linear_wide.hip: fused forward / ELBO / backward of a wide linear decoder (BASELINE config 4) ---------------------------
bool lwd_supported(int B, int D, int L);
int lwd_row_block(int B, int D, int n_cu);
size_t lwd_gpart_bytes(int B, int D, int L);
int launch_lwd(const float* samples, const float* Wd, const float* bd, const float* x, const float* z2, const float* eps_param, float eps_cli,
               float inv_bt, float* gpart, float* slab0, int64_t slab_stride, float* part, int B, int D, int L, int RB, hipStream_t st);
int launch_lwd_second(const float* gpart, int ncb, float* dsamp, const float* mu, const float* z1, float* partial, int rows, int L, int S,
                      int rows_per_split, float inv_bt, const float* part, int nblk, float* epartial, int32_t* step_dev, hipStream_t st);
bool lin_moments_supported(const vaek_ctx* c);
size_t lin_moment_len(const vaek_ctx* c);
int lin_moments(vaek_ctx* c, const float* x, const float* z1, const float* z2, double* M_out, void* ws, hipStream_t st);
int lin_update(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev, const double* M_in, float lr, hipStream_t st);
struct BatchArgs;
int lin_train_steps_gen(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev, const BatchArgs& gen, int n_steps,
                        float lr, void* ws, hipStream_t st);

// ---- rng.hip ------------------------------------------------------------------------------
// validates the arguments of vaek_make_batch* and fills `out`
int make_batch_args(vaek_ctx* ctx, int32_t kind, const float* A, int32_t dd, int32_t did, int32_t pad, float var_added,
                    float* x, float* z1, float* z2, int32_t rows, int64_t row0, uint64_t seed, const int32_t* step_dev,
                    uint32_t step_host, int32_t* counter, int32_t which, uint32_t tag, BatchArgs* out);
int make_batch_launch(vaek_ctx* ctx, const BatchArgs& a, hipStream_t st);

// ---- fused_small.hip ----------------------------------------------------------------------
bool mlp1_supported(const vaek_ctx* c);          // fused_mlp1.hip: whole-network kernel for one-hidden-layer MLPs (<= 256 units)
int mlp1_grid(const vaek_ctx* c);
int mlp1_launch(vaek_ctx* c, const float* params, const float* x, const float* z1, const float* z2, float* partials, int pstride,
                int32_t* step_dev, hipStream_t st);
bool fused_supported(const vaek_ctx* c);
size_t fused_workspace_bytes(const vaek_ctx* c);
// gen != nullptr: the finalize launch carries extra blocks that draw the NEXT step's batch (vaek_train_step_gen)
int fused_train_step(vaek_ctx* c, float* params, float* grads, float* m, float* v, int32_t* step_dev,
                     const float* x, const float* z1, const float* z2, float lr, bool apply_adam, bool exchange,
                     void* ws, hipStream_t st, const BatchArgs* gen = nullptr);

}  // namespace vaek
