// attention.h - internal interface between the attention dispatcher and its kernels.
#pragma once
#include "common.h"

namespace leccr {

struct AttnDims {
    int64_t b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo;
    DropCfg drop;       // attention-probability dropout (thr == 0: off)
    bool dropout;
    bool causal = false;   // key j visible to query i only for j <= i (CLIP's text transformer, clip/model.py:328-334); generic kernels only
};

template <typename T>
int attn_generic_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                     const AttnDims& D, hipStream_t st);
template <typename T>
int attn_generic_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                     const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, const AttnDims& D,
                     hipStream_t st);

// bf16 attention with a thin side (<= 8 keys or <= 8 queries per batch element), any head size (attention_thin.hip)
int attn_thin_kind(const AttnDims& D, int dtype);          // 0: not taken, 1: thin-K, 2: thin-Q
int attn_thin_fwd(int kind, const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                  const AttnDims& D, hipStream_t st);
int attn_thin_bwd(int kind, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                  const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, const AttnDims& D, hipStream_t st);

// optional fp8 copies of the outputs (leccr_attn_fwd_q8 / leccr_attn_bwd_q8): forward o8 (e4m3), backward dq8 / dk8 / dv8 (e5m2), row
// stride ld8 bytes, under `site`.  *fused is set when the kernel that ran wrote them itself; otherwise the caller runs quantise passes.
struct AttnQ8 {
    void *o8 = nullptr, *dq8 = nullptr, *dk8 = nullptr, *dv8 = nullptr;
    int64_t ld8 = 0;
    float* site = nullptr;
};

// bf16, head size 64, MFMA flash kernels (attention_mfma.hip)
bool attn_mfma_supported(const AttnDims& D, int dtype);
int attn_mfma_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                  const AttnDims& D, hipStream_t st, const AttnQ8* q8 = nullptr, bool* fused = nullptr);
int attn_mfma_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                  const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, const AttnDims& D, hipStream_t st,
                  const AttnQ8* q8 = nullptr, bool* fused = nullptr);

// f32, head size 64: flash kernels on v_mfma_f32_32x32x2_f32 (attention_f32.hip) - the reference-precision path
bool attn_f32_supported(const AttnDims& D, int dtype, const void* const* ptrs, int nptrs);
int attn_f32_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask, const AttnDims& D, hipStream_t st);
int attn_f32_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, const uint8_t* kmask,
                 void* dq, void* dk, void* dv, float* delta, const AttnDims& D, hipStream_t st);

// .. and the same problems with every operand as a bf16 (hi, lo) pair on v_mfma_f32_32x32x16_bf16 (attention_x3.hip): the default for f32
bool attn_x3_wanted();
// operand copies written / read by the fp32 pair-form kernels themselves (attention_x3.hip: struct S3): every pointer at column 0 of its tensor
// inside segment 0 of a [rows, 3 seg] bf16 copy (leccr_split3's layout); null = the f32 tensor
struct AttnS3 {
    void* o3 = nullptr;             // forward: the X-form (hi | hi | lo) copy of o
    const void* o3_in = nullptr;    // backward: that copy in place of o
    void* dq3 = nullptr;            // backward: G-form (hi | lo | hi) copies
    void* dk3 = nullptr;
    void* dv3 = nullptr;
    int64_t ld3 = 0, seg = 0;       // layout of the output copies (o3 / dq3 / dk3 / dv3)
    int64_t ldo3 = 0, sego3 = 0;    // layout of o3_in
    float* bq = nullptr;            // optional [heads * hd] each: += column sums of dq / dk / dv (the in-projection's bias gradient)
    float* bk = nullptr;
    float* bv = nullptr;
};
int attn_x3_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask, const AttnDims& D, hipStream_t st,
                const AttnS3* s3 = nullptr);
int attn_x3_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, const uint8_t* kmask,
                void* dq, void* dk, void* dv, float* delta, const AttnDims& D, hipStream_t st, const AttnS3* s3 = nullptr);

extern bool g_force_generic_attn;
int attn_set_variant(int v);      // 0 = default (environment), 16 / 32 = queries per wave of the MFMA forward kernel; returns the one in effect

}  // namespace leccr
