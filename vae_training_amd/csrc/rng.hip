// K7 of SURVEY.md 7.1: on-device generation of what feeds the hot path every iteration -- the dataset
// batch (datasets.py:75-84 sphere, :183-195 linear_gaussian, :240-249 sigmoid) and the latent draw
// z ~ N(0,1)^(B x (L+D)) (model.py:225-228, split at vae.py:127-128) -- as one Philox4x32-10 kernel.
// Counter-based: row i of step t draws block q from counter (i, q, t, tag) under key = seed, so a
// batch is reproducible, shardable (a rank generates only its rows, with GLOBAL row indices) and
// replayable from a hipGraph (t is the device-resident Adam step counter).  jax.random's threefry
// streams are not reproduced (not possible without JAX); the distributions are (tests/test_rng.py).
#include "rng_dev.h"

namespace vaek {

__global__ __launch_bounds__(256) void make_batch_kernel(const BatchArgs a) {
    const unsigned step = make_batch_step(a);
    make_batch_items(a, step, (long long)blockIdx.x * blockDim.x + threadIdx.x);
    make_batch_advance(a, step, blockIdx.x == 0 && threadIdx.x == 0);
}

__global__ __launch_bounds__(256) void rng_fill_kernel(float* out_n, unsigned* out_u, long long n, unsigned long long seed,
                                                      unsigned step, unsigned tag) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one Philox block = 4 outputs
    if (4 * b >= n) return;
    const uint4 r = philox4x32_10(make_uint4((unsigned)b, (unsigned)(b >> 32), step, tag), make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    float nv[4];
    normals4(r, nv);
    const unsigned uv[4] = {r.x, r.y, r.z, r.w};
    for (int k = 0; k < 4 && 4 * b + k < n; ++k) {
        if (out_n) out_n[4 * b + k] = nv[k];
        if (out_u) out_u[4 * b + k] = uv[k];
    }
}

}  // namespace vaek

using namespace vaek;

extern "C" {

}  // extern "C"

namespace vaek {

int make_batch_args(vaek_ctx* ctx, int32_t kind, const float* A, int32_t dd, int32_t did, int32_t pad, float var_added,
                    float* x, float* z1, float* z2, int32_t rows, int64_t row0, uint64_t seed, const int32_t* step_dev,
                    uint32_t step_host, int32_t* counter, int32_t which, uint32_t tag, BatchArgs* out) {
    if (!ctx || (counter && (which | 1) != 1) || (!z1) != (!z2) || (!x && !z1) || rows <= 0 || kind < 0 || kind > 2 || dd <= 0 || dd > 16 || did > 16 || pad < 0 ||
        (kind != 2 && x && !A) || tag >= 0x40000000u) {
        set_error("vaek_make_batch: invalid argument");
        return VAEK_ERR_INVALID;
    }
    const int D = dd + pad + (kind == 1 ? 1 : 0);
    if (z1 && D != ctx->D) { set_error("vaek_make_batch: dataset dimension %d != context data_dim %d", D, ctx->D); return VAEK_ERR_INVALID; }
    BatchArgs a{};
    a.kind = kind; a.A = A; a.dd = dd; a.did = did; a.pad = pad; a.noise_std = var_added > 0.f ? sqrtf(var_added) : 0.f;
    a.x = x; a.z1 = z1; a.z2 = z2; a.rows = rows; a.row0 = row0; a.D = D; a.L = ctx->L;
    a.seed = seed; a.step_dev = step_dev; a.step_host = step_host; a.tag = tag; a.counter = counter; a.which = which;
    *out = a;
    return VAEK_OK;
}

int make_batch_launch(vaek_ctx* ctx, const BatchArgs& a, hipStream_t st) {
    Profiler* outer = g_prof;
    g_prof = &ctx->prof;
    {
        ProfScope ps("make_batch", st);
        launch_k(ps, make_batch_kernel, dim3((unsigned)((make_batch_item_count(a) + 255) / 256)), dim3(256), 0, st, a);
    }
    g_prof = outer;
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

}  // namespace vaek

extern "C" {

static int make_batch_impl(vaek_ctx* ctx, int32_t kind, const float* A, int32_t dd, int32_t did, int32_t pad, float var_added,
                           float* x, float* z1, float* z2, int32_t rows, int64_t row0, uint64_t seed, const int32_t* step_dev,
                           uint32_t step_host, int32_t* counter, int32_t which, uint32_t tag, void* stream) {
    BatchArgs a;
    int rc = make_batch_args(ctx, kind, A, dd, did, pad, var_added, x, z1, z2, rows, row0, seed, step_dev, step_host, counter, which, tag, &a);
    if (rc) return rc;
    return make_batch_launch(ctx, a, (hipStream_t)stream);
}

int vaek_make_batch(vaek_ctx* ctx, int32_t kind, const float* A, int32_t dd, int32_t did, int32_t pad, float var_added,
                    float* x, float* z1, float* z2, int32_t rows, int64_t row0, uint64_t seed, const int32_t* step_dev,
                    uint32_t step_host, uint32_t tag, void* stream) {
    return make_batch_impl(ctx, kind, A, dd, did, pad, var_added, x, z1, z2, rows, row0, seed, step_dev, step_host, nullptr, 0, tag, stream);
}

int vaek_make_batch_next(vaek_ctx* ctx, int32_t kind, const float* A, int32_t dd, int32_t did, int32_t pad, float var_added,
                         float* x, float* z1, float* z2, int32_t rows, int64_t row0, uint64_t seed, int32_t* counter,
                         int32_t which, uint32_t tag, void* stream) {
    if (!counter) { set_error("vaek_make_batch_next: counter is NULL"); return VAEK_ERR_INVALID; }
    return make_batch_impl(ctx, kind, A, dd, did, pad, var_added, x, z1, z2, rows, row0, seed, nullptr, 0, counter, which, tag, stream);
}

int vaek_rng_fill(vaek_ctx* ctx, float* normals, uint32_t* bits, int64_t n, uint64_t seed, uint32_t step, uint32_t tag,
                  void* stream) {
    if (!ctx || n < 0 || (!normals && !bits)) { set_error("vaek_rng_fill: invalid argument"); return VAEK_ERR_INVALID; }
    if (n == 0) return VAEK_OK;
    const long long nb = (n + 3) / 4;
    ProfScope ps("rng_fill", (hipStream_t)stream);
    launch_k(ps, rng_fill_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, normals, bits,
             (long long)n, (unsigned long long)seed, step, tag);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int vaek_set_loss_history(vaek_ctx* ctx, float* buf, int64_t cap) {
    if (!ctx || (buf && cap <= 0)) { set_error("vaek_set_loss_history: invalid argument"); return VAEK_ERR_INVALID; }
    ctx->loss_hist = buf;
    ctx->loss_hist_cap = buf ? cap : 0;
    return VAEK_OK;
}

}  // extern "C"
