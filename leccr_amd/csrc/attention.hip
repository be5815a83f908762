// attention.hip - C entry points for attention; picks the MFMA flash kernels (bf16, head size 64) or the generic
// kernels (f32 parity path, other head sizes).
#include "attention.h"

namespace leccr {
bool g_force_generic_attn = false;
static int g_last_attn_kernel = -1;      // 0 generic, 1 MFMA flash (head 64), 2 thin-K, 3 thin-Q, 4 f32 MFMA flash (head 64), 5 f32 as bf16 hi / lo pairs (head 64)
// LECCR_ATTN_THIN=2: the thin kernels also take the head-64 problems with a thin side (default: those stay on the MFMA kernels)
static bool thin_first() {
    static const bool v = [] { const char* e = getenv("LECCR_ATTN_THIN"); return e && e[0] == '2'; }();
    return v;
}

static int check_dims(const char* what, const AttnDims& D, int dtype) {
    LECCR_REQUIRE(dtype == LECCR_F32 || dtype == LECCR_BF16, LECCR_EINVAL, "%s: bad dtype", what);
    LECCR_REQUIRE(D.b >= 0 && D.heads > 0 && D.Lq >= 0 && D.Lk > 0 && D.hd > 0, LECCR_EINVAL, "%s: bad sizes", what);
    LECCR_REQUIRE(D.ldq >= D.heads * D.hd && D.ldk >= D.heads * D.hd && D.ldv >= D.heads * D.hd &&
                      D.ldo >= D.heads * D.hd,
                  LECCR_EINVAL, "%s: row stride smaller than heads*hd", what);
    return LECCR_OK;
}
}  // namespace leccr

using namespace leccr;

static int attn_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                         int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk,
                         int64_t ldv, int64_t ldo, int dtype, float drop_p, uint64_t seed, void* stream, const AttnQ8* q8, bool* fused,
                         bool causal = false) {
    LECCR_REQUIRE(q && k && v && o && lse, LECCR_EINVAL, "attn_fwd: null pointer");
    LECCR_REQUIRE(drop_p >= 0.f && drop_p < 1.f, LECCR_EINVAL, "attn_fwd: drop_p must be in [0,1)");
    LECCR_REQUIRE(b * heads * Lq * Lk < (1ll << 32) || drop_p == 0.f, LECCR_EINVAL, "attn_fwd: dropout index space exceeds 2^32");
    AttnDims D{b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, make_drop(drop_p, seed), drop_p > 0.f};
    D.causal = causal;
    if (int rc = check_dims("attn_fwd", D, dtype)) return rc;
    if (b == 0 || Lq == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PROF_ATTN_FWD, 4.0 * (double)b * heads * Lq * Lk * hd, st);
    if (causal) {                        // the causal mask exists in the generic kernels only (77-token CLIP text tower: a small problem)
        g_last_attn_kernel = 0;
        return dtype == LECCR_F32 ? attn_generic_fwd<float>(q, k, v, o, lse, kmask, D, st) : attn_generic_fwd<bf16_t>(q, k, v, o, lse, kmask, D, st);
    }
    if (!g_force_generic_attn && attn_mfma_supported(D, dtype) && !(thin_first() && attn_thin_kind(D, dtype))) {
        g_last_attn_kernel = 1;
        return attn_mfma_fwd(q, k, v, o, lse, kmask, D, st, q8, fused);
    }
    if (!g_force_generic_attn) {         // head sizes the MFMA kernels do not cover, one side thin (the caption interaction at d = 768)
        if (const int kind = attn_thin_kind(D, dtype)) {
            g_last_attn_kernel = 1 + kind;
            return attn_thin_fwd(kind, q, k, v, o, lse, kmask, D, st);
        }
    }
    {
        const void* ptrs[4] = {q, k, v, o};
        if (!g_force_generic_attn && attn_f32_supported(D, dtype, ptrs, 4)) {      // fp32 (reference precision) on the f32 matrix-core instruction
            if (attn_x3_wanted()) {                                                // .. as bf16 (hi, lo) operand pairs: 5.3 x fewer matrix-pipe cycles
                g_last_attn_kernel = 5;
                return attn_x3_fwd(q, k, v, o, lse, kmask, D, st);
            }
            g_last_attn_kernel = 4;
            return attn_f32_fwd(q, k, v, o, lse, kmask, D, st);
        }
    }
    g_last_attn_kernel = 0;
    return dtype == LECCR_F32 ? attn_generic_fwd<float>(q, k, v, o, lse, kmask, D, st)
                              : attn_generic_fwd<bf16_t>(q, k, v, o, lse, kmask, D, st);
}

extern "C" int leccr_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                              int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk,
                              int64_t ldv, int64_t ldo, int dtype, float drop_p, uint64_t seed, void* stream) {
    return attn_fwd_impl(q, k, v, o, lse, kmask, b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, dtype, drop_p, seed, stream, nullptr, nullptr);
}

// .. and the e4m3 copy of o for the product that consumes it (out_proj in fp8 mode): written by the attention kernel where the variant
// that runs has the epilogue (the long unmasked sequences of the ViT), by a leccr_fp8_quantize pass over o otherwise - same bytes contract
extern "C" int leccr_attn_fwd_q8(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                                 int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk,
                                 int64_t ldv, int64_t ldo, int dtype, float drop_p, uint64_t seed, void* o8, int64_t ldo8, float* site,
                                 void* stream) {
    LECCR_REQUIRE(o8 && site && ldo8 >= heads * hd && dtype == LECCR_BF16, LECCR_EINVAL, "attn_fwd_q8: needs o8, its site, ldo8 >= heads * hd, bf16");
    AttnQ8 q8;
    q8.o8 = o8; q8.ld8 = ldo8; q8.site = site;
    bool fused = false;
    int rc = attn_fwd_impl(q, k, v, o, lse, kmask, b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, dtype, drop_p, seed, stream, &q8, &fused);
    if (rc == LECCR_OK && !fused && b * Lq > 0) rc = leccr_fp8_quantize(o, o8, b * Lq, heads * hd, ldo, ldo8, dtype, 0, site, stream);
    return rc;
}

// Causal self-attention (key j visible to query i only for j <= i; no dropout): CLIP's text transformer (clip/model.py:328-334, 354), the
// caption encoder of caption_encoder_name = 'clip' (models/model_retrieval_caption.py:57-58).  Generic kernels, Lq == Lk.
extern "C" int leccr_attn_fwd_causal(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                                     int64_t b, int64_t heads, int64_t L, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                     int dtype, void* stream) {
    return attn_fwd_impl(q, k, v, o, lse, kmask, b, heads, L, L, hd, ldq, ldk, ldv, ldo, dtype, 0.f, 0, stream, nullptr, nullptr, true);
}

extern "C" int leccr_set_attn_variant(int v) { return attn_set_variant(v); }
extern "C" int leccr_last_attn_kernel(void) { return g_last_attn_kernel; }

static int attn_bwd_impl(const void* q, const void* k, const void* v, const void* o, const void* dout,
                         const float* lse, const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta,
                         int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk,
                         int64_t ldv, int64_t ldo, int dtype, float drop_p, uint64_t seed, void* stream, const AttnQ8* q8, bool* fused,
                         bool causal = false) {
    LECCR_REQUIRE(q && k && v && o && dout && lse && dq && dk && dv && delta, LECCR_EINVAL, "attn_bwd: null pointer");
    LECCR_REQUIRE(drop_p >= 0.f && drop_p < 1.f, LECCR_EINVAL, "attn_bwd: drop_p must be in [0,1)");
    AttnDims D{b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, make_drop(drop_p, seed), drop_p > 0.f};
    D.causal = causal;
    if (int rc = check_dims("attn_bwd", D, dtype)) return rc;
    if (b == 0 || Lq == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PROF_ATTN_BWD, 10.0 * (double)b * heads * Lq * Lk * hd, st);
    if (causal) {
        g_last_attn_kernel = 0;
        return dtype == LECCR_F32 ? attn_generic_bwd<float>(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, D, st)
                                  : attn_generic_bwd<bf16_t>(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, D, st);
    }
    if (!g_force_generic_attn && attn_mfma_supported(D, dtype) && !(thin_first() && attn_thin_kind(D, dtype))) {
        g_last_attn_kernel = 1;
        return attn_mfma_bwd(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, D, st, q8, fused);
    }
    if (!g_force_generic_attn) {
        if (const int kind = attn_thin_kind(D, dtype)) {
            g_last_attn_kernel = 1 + kind;
            return attn_thin_bwd(kind, q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, D, st);
        }
    }
    {
        const void* ptrs[8] = {q, k, v, o, dout, dq, dk, dv};
        if (!g_force_generic_attn && attn_f32_supported(D, dtype, ptrs, 8)) {
            if (attn_x3_wanted()) {
                g_last_attn_kernel = 5;
                return attn_x3_bwd(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, D, st);
            }
            g_last_attn_kernel = 4;
            return attn_f32_bwd(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, D, st);
        }
    }
    g_last_attn_kernel = 0;
    return dtype == LECCR_F32 ? attn_generic_bwd<float>(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, D, st)
                              : attn_generic_bwd<bf16_t>(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, D, st);
}

extern "C" int leccr_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                              const float* lse, const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta,
                              int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk,
                              int64_t ldv, int64_t ldo, int dtype, float drop_p, uint64_t seed, void* stream) {
    return attn_bwd_impl(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, dtype, drop_p, seed,
                         stream, nullptr, nullptr);
}

// .. and the e5m2 copies of dq / dk / dv (row stride ld8: the three column blocks of one packed [rows, 3 d] gradient, or separate
// tensors) under ONE site, for the in_proj dX and weight-gradient products in fp8 mode
extern "C" int leccr_attn_bwd_q8(const void* q, const void* k, const void* v, const void* o, const void* dout,
                                 const float* lse, const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta,
                                 int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk,
                                 int64_t ldv, int64_t ldo, int dtype, float drop_p, uint64_t seed, void* dq8, void* dk8, void* dv8,
                                 int64_t ld8, float* site, void* stream) {
    LECCR_REQUIRE(dq8 && dk8 && dv8 && site && ld8 >= heads * hd && dtype == LECCR_BF16, LECCR_EINVAL,
                  "attn_bwd_q8: needs dq8 / dk8 / dv8, their site, ld8 >= heads * hd, bf16");
    AttnQ8 q8;
    q8.dq8 = dq8; q8.dk8 = dk8; q8.dv8 = dv8; q8.ld8 = ld8; q8.site = site;
    bool fused = false;
    int rc = attn_bwd_impl(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, dtype, drop_p, seed,
                           stream, &q8, &fused);
    if (rc == LECCR_OK && !fused) {
        if (b * Lq > 0) rc = leccr_fp8_quantize(dq, dq8, b * Lq, heads * hd, ldq, ld8, dtype, 1, site, stream);
        if (rc == LECCR_OK && b * Lk > 0) rc = leccr_fp8_quantize(dk, dk8, b * Lk, heads * hd, ldk, ld8, dtype, 1, site, stream);
        if (rc == LECCR_OK && b * Lk > 0) rc = leccr_fp8_quantize(dv, dv8, b * Lk, heads * hd, ldv, ld8, dtype, 1, site, stream);
    }
    return rc;
}

extern "C" int leccr_attn_bwd_causal(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                                     const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, int64_t b, int64_t heads, int64_t L,
                                     int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int dtype, void* stream) {
    return attn_bwd_impl(q, k, v, o, dout, lse, kmask, dq, dk, dv, delta, b, heads, L, L, hd, ldq, ldk, ldv, ldo, dtype, 0.f, 0, stream,
                         nullptr, nullptr, true);
}
