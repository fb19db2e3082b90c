// One-shot peer-to-peer all-reduce over xGMI (placeholder until the P2P kernels land).
#include "vaek_internal.h"

using namespace vaek;
extern "C" {
int vaek_comm_buffer_bytes(const vaek_ctx* ctx, size_t* bytes) {
    if (!ctx || !bytes) { set_error("null argument"); return VAEK_ERR_INVALID; }
    *bytes = 0;
    return VAEK_OK;
}
int vaek_comm_export(vaek_ctx*, void*, uint8_t*) { set_error("communicator not built"); return VAEK_ERR_COMM; }
int vaek_comm_init(vaek_ctx*, void*, const uint8_t*) { set_error("communicator not built"); return VAEK_ERR_COMM; }
int vaek_comm_destroy(vaek_ctx*) { return VAEK_OK; }
int vaek_comm_allreduce(vaek_ctx*, float*, int64_t, void*) { set_error("communicator not initialised"); return VAEK_ERR_COMM; }
}
