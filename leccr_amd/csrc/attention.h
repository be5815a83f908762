// attention.h - internal interface between the attention dispatcher and its kernels.
#pragma once
#include "common.h"

namespace leccr {

struct AttnDims {
    int64_t b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo;
    DropCfg drop;       // attention-probability dropout (thr == 0: off)
    bool dropout;
};

template <typename T>
int attn_generic_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                     const AttnDims& D, hipStream_t st);
template <typename T>
int attn_generic_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                     const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, const AttnDims& D,
                     hipStream_t st);

// bf16, head size 64, MFMA flash kernels (attention_mfma.hip)
bool attn_mfma_supported(const AttnDims& D, int dtype);
int attn_mfma_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                  const AttnDims& D, hipStream_t st);
int attn_mfma_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                  const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, const AttnDims& D, hipStream_t st);

extern bool g_force_generic_attn;
int attn_set_variant(int v);      // 0 = default (environment), 16 / 32 = queries per wave of the MFMA forward kernel; returns the one in effect

}  // namespace leccr
