// attention_mfma.hip - bf16 / head-size-64 flash attention on MFMA (placeholder until the kernels land:
// attn_mfma_supported() returns false so every call takes the generic kernels).
#include "attention.h"

namespace leccr {
bool attn_mfma_supported(const AttnDims&, int) { return false; }
int attn_mfma_fwd(const void*, const void*, const void*, void*, float*, const uint8_t*, const AttnDims&, hipStream_t) {
    set_error("attn_mfma_fwd: not built");
    return LECCR_EINVAL;
}
int attn_mfma_bwd(const void*, const void*, const void*, const void*, const void*, const float*, const uint8_t*, void*,
                  void*, void*, float*, const AttnDims&, hipStream_t) {
    set_error("attn_mfma_bwd: not built");
    return LECCR_EINVAL;
}
}  // namespace leccr
