// Fused small-model train step (placeholder until the fused kernels land).
#include "vaek_internal.h"

namespace vaek {
bool fused_supported(const vaek_ctx*) { return false; }
size_t fused_workspace_bytes(const vaek_ctx*) { return 0; }
int fused_train_step(vaek_ctx*, float*, float*, float*, float*, int32_t*, const float*, const float*, const float*,
                     float, bool, void*, hipStream_t) {
    set_error("fused path not available");
    return VAEK_ERR_INVALID;
}
}  // namespace vaek
