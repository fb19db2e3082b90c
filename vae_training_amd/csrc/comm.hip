// Host side of the one-shot peer-to-peer all-reduce (include/vaek.h, vaek_comm_*): allocation of the
// uncached granule buffer, HIP IPC export/import, a stand-alone all-reduce kernel and the self-test
// the Python side runs before trusting the transport (vae_training_amd/parallel.py).
#include <string.h>

#include "comm_dev.h"
#include "vaek_internal.h"

namespace vaek {

static size_t granules_per_region(const vaek_ctx* c) { return 2ull * c->cfg.world * c->comm.ng; }

// The epoch of a stand-alone all-reduce lives on the DEVICE (a word of the status line, behind the give-up flag): every workgroup
// reads it, a one-thread kernel behind the exchange advances it -- so the call can be captured into a hipGraph and replayed (a
// host-side counter would be frozen into the captured kernel arguments).  Every rank makes the same calls: the counters agree.
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(CommDev c, float* buf, long long n, const unsigned* epoch_dev) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    buf[i] = comm_exchange_sum(c, epoch_dev[0] + 1u, (int)i, buf[i]);
}
__global__ void p2p_epoch_bump_kernel(unsigned* epoch_dev) { epoch_dev[0] += 1u; }

CommDev comm_dev(const vaek_ctx* c, int region) {
    CommDev d{};
    const size_t off = (size_t)region * granules_per_region(c);
    for (int r = 0; r < c->cfg.world; ++r) d.peer[r] = reinterpret_cast<unsigned long long*>(c->comm.peers[r]) + off;
    d.status = reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned long long*>(c->comm.local) + 2 * granules_per_region(c));
    d.world = c->cfg.world; d.rank = c->cfg.rank; d.ng = c->comm.ng;
    return d;
}

}  // namespace vaek

using namespace vaek;

extern "C" {

int vaek_comm_buffer_bytes(const vaek_ctx* ctx, size_t* bytes) {
    if (!ctx || !bytes) { set_error("null argument"); return VAEK_ERR_INVALID; }
    const size_t ng = (size_t)((ctx->P + kExtra + 63) / 64 * 64);
    // two regions (train step: epoch = Adam step; stand-alone all-reduce: its own epochs) + status line
    *bytes = 2 * (2ull * ctx->cfg.world * ng) * sizeof(unsigned long long) + 256 + lin_comm_bytes(ctx);
    return VAEK_OK;
}

int vaek_comm_create(vaek_ctx* ctx, uint8_t handle_out[64]) {
    if (!ctx || !handle_out) { set_error("null argument"); return VAEK_ERR_INVALID; }
    if (ctx->cfg.world < 2 || ctx->cfg.world > kMaxWorld) { set_error("p2p communicator supports 2..%d ranks, world=%d", kMaxWorld, ctx->cfg.world); return VAEK_ERR_COMM; }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (ctx->comm.local) { set_error("communicator already created"); return VAEK_ERR_COMM; }
    size_t bytes = 0;
    vaek_comm_buffer_bytes(ctx, &bytes);
    VAEK_HIP_CHECK(hipSetDevice(ctx->cfg.device));
    void* p = nullptr;
    // uncached (MTYPE_UC): peers' xGMI stores land in HBM, and this device's polling loads must never be
    // served from a stale L2 line
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            set_error("cannot allocate an uncached/fine-grained exchange buffer of %zu bytes", bytes);
            return VAEK_ERR_COMM;
        }
    }
    VAEK_HIP_CHECK(hipMemset(p, 0, bytes));
    VAEK_HIP_CHECK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        set_error("hipIpcGetMemHandle failed (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
        return VAEK_ERR_COMM;
    }
    memcpy(handle_out, &h, 64);
    ctx->comm.local = p;
    ctx->comm.ng = (int)((ctx->P + kExtra + 63) / 64 * 64);
    ctx->comm.bytes = bytes;
    ctx->comm.lin_bytes = lin_comm_bytes(ctx);
    ctx->comm.lin_off = bytes - ctx->comm.lin_bytes;          // behind the two granule regions and the 256-byte status line
    return VAEK_OK;
}

int vaek_comm_init(vaek_ctx* ctx, const uint8_t* all_handles) {
    if (!ctx || !all_handles) { set_error("null argument"); return VAEK_ERR_INVALID; }
    if (!ctx->comm.local) { set_error("vaek_comm_create first"); return VAEK_ERR_COMM; }
    VAEK_HIP_CHECK(hipSetDevice(ctx->cfg.device));
    ctx->comm.peers.assign(ctx->cfg.world, nullptr);
    for (int r = 0; r < ctx->cfg.world; ++r) {
        if (r == ctx->cfg.rank) { ctx->comm.peers[r] = ctx->comm.local; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, all_handles + 64 * r, 64);
        void* q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("hipIpcOpenMemHandle(rank %d) failed: %s", r, hipGetErrorString(e));
            vaek_comm_destroy(ctx);
            return VAEK_ERR_COMM;
        }
        ctx->comm.peers[r] = q;
    }
    ctx->comm.epoch = 0;
    ctx->comm.ready = true;
    return VAEK_OK;
}

int vaek_comm_destroy(vaek_ctx* ctx) {
    if (!ctx) return VAEK_OK;
    for (int r = 0; r < (int)ctx->comm.peers.size(); ++r)
        if (r != ctx->cfg.rank && ctx->comm.peers[r]) (void)hipIpcCloseMemHandle(ctx->comm.peers[r]);
    ctx->comm.peers.clear();
    if (ctx->comm.local) (void)hipFree(ctx->comm.local);
    ctx->comm.local = nullptr;
    ctx->comm.ready = false;
    return VAEK_OK;
}

int vaek_comm_allreduce(vaek_ctx* ctx, float* buf, int64_t n, void* stream) {
    if (!ctx || !buf || n < 0) { set_error("vaek_comm_allreduce: invalid argument"); return VAEK_ERR_INVALID; }
    if (!ctx->comm.ready) { set_error("communicator not initialised"); return VAEK_ERR_COMM; }
    if (n > ctx->comm.ng) { set_error("vaek_comm_allreduce: n=%lld exceeds the exchange buffer (%d)", (long long)n, ctx->comm.ng); return VAEK_ERR_COMM; }
    if (n == 0) return VAEK_OK;
    const CommDev d = comm_dev(ctx, 1);
    unsigned* epoch_dev = d.status + 1;               // zeroed with the buffer at vaek_comm_create
    {
        ProfScope ps("p2p_allreduce", (hipStream_t)stream);
        launch_k(ps, p2p_allreduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, buf, (long long)n,
                 (const unsigned*)epoch_dev);
    }
    hipLaunchKernelGGL(p2p_epoch_bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, epoch_dev);
    VAEK_HIP_CHECK(hipGetLastError());
    return VAEK_OK;
}

int vaek_comm_status(vaek_ctx* ctx, int32_t* timed_out) {
    if (!ctx || !timed_out) { set_error("null argument"); return VAEK_ERR_INVALID; }
    if (!ctx->comm.local) { *timed_out = 0; return VAEK_OK; }
    unsigned int s = 0;
    const CommDev d = comm_dev(ctx, 0);
    VAEK_HIP_CHECK(hipMemcpy(&s, d.status, sizeof(s), hipMemcpyDeviceToHost));
    *timed_out = (int32_t)s;
    return VAEK_OK;
}

}  // extern "C"
