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
    // (leccr_attn_bwd_q8: dq = dk = dv = NULL asks for the e5m2 copies alone - the head-64 MFMA kernels write them from their registers)
    const bool only8 = !dq && !dk && !dv && q8 && q8->dq8 && q8->dk8 && q8->dv8;
    LECCR_REQUIRE(q && k && v && o && dout && lse && ((dq && dk && dv) || only8) && delta, LECCR_EINVAL, "attn_bwd: null pointer");
    LECCR_REQUIRE(drop_p >= 0.f && drop_p < 1.f, LECCR_EINVAL, "attn_bwd: drop_p must be in [0,1)");
    AttnDims D{b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, make_drop(drop_p, seed), drop_p > 0.f};
    D.causal = causal;
    if (int rc = check_dims("attn_bwd", D, dtype)) return rc;
    if (only8)
        LECCR_REQUIRE(!causal && !g_force_generic_attn && attn_mfma_supported(D, dtype) && !(thin_first() && attn_thin_kind(D, dtype)) &&
                          (q8->ld8 & 3) == 0 && (((uintptr_t)q8->dq8 | (uintptr_t)q8->dk8 | (uintptr_t)q8->dv8) & 3) == 0,
                      LECCR_EINVAL, "attn_bwd_q8: the copies alone (null dq / dk / dv) exist on the head-64 bf16 MFMA kernels only (4-byte aligned copies)");
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

// ---- fp32 attention with the operand copies of its outputs (round 4).  In the fp32 path (operands as bf16 hi / lo pairs) o is read only as
// the X-form copy of out_proj's products and dq / dk / dv only as the G-form copy of in_proj's: the pair-form kernels (attention_x3.hip) write
// those copies - [rows, 3 seg] bf16 in leccr_split3's layout, bit for bit what leccr_split3 makes of the f32 tensor - from their registers, and
// the f32 tensors are not written.  leccr_attn_s3_takes says whether a problem runs on those kernels (f32, head size 64, no causal mask, row
// strides % 4 == 0, LECCR_ATTN_F32 != exact): callers fall back to leccr_attn_fwd / _bwd + leccr_split3 otherwise.
extern "C" int leccr_attn_s3_takes(int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv,
                                   int64_t ld3, int64_t seg3) {
    AttnDims D{b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldq, make_drop(0.f, 0), false};
    if (b < 0 || heads <= 0 || Lq < 0 || Lk <= 0 || ldq < heads * hd || ldk < heads * hd || ldv < heads * hd) return 0;
    if (seg3 < heads * hd || ld3 < 3 * seg3 || ((ld3 | seg3) & 7)) return 0;          // 16-byte rows and segments (the delta read takes 8 bf16 at a time)
    return !g_force_generic_attn && attn_f32_supported(D, LECCR_F32, nullptr, 0) && attn_x3_wanted() ? 1 : 0;
}

// o3: the X-form (hi | hi | lo) copy of o, pointer at o's column 0 inside segment 0, row stride ld3, segment width seg3; o itself is not written
extern "C" int leccr_attn_fwd_s3(const void* q, const void* k, const void* v, void* o3, float* lse, const uint8_t* kmask,
                                 int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv,
                                 int64_t ld3, int64_t seg3, float drop_p, uint64_t seed, void* stream) {
    LECCR_REQUIRE(q && k && v && o3 && lse, LECCR_EINVAL, "attn_fwd_s3: null pointer");
    LECCR_REQUIRE(drop_p >= 0.f && drop_p < 1.f, LECCR_EINVAL, "attn_fwd_s3: drop_p must be in [0,1)");
    LECCR_REQUIRE(b * heads * Lq * Lk < (1ll << 32) || drop_p == 0.f, LECCR_EINVAL, "attn_fwd_s3: dropout index space exceeds 2^32");
    AttnDims D{b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldq, make_drop(drop_p, seed), drop_p > 0.f};
    const void* ptrs[4] = {q, k, v, o3};
    LECCR_REQUIRE(leccr_attn_s3_takes(b, heads, Lq, Lk, hd, ldq, ldk, ldv, ld3, seg3) && attn_f32_supported(D, LECCR_F32, ptrs, 4), LECCR_EINVAL,
                  "attn_fwd_s3: not a problem of the fp32 pair-form kernels (leccr_attn_s3_takes), or a pointer not 16-byte aligned");
    if (b == 0 || Lq == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PROF_ATTN_FWD, 4.0 * (double)b * heads * Lq * Lk * hd, st);
    AttnS3 s3;
    s3.o3 = o3; s3.ld3 = ld3; s3.seg = seg3;
    g_last_attn_kernel = 5;
    return attn_x3_fwd(q, k, v, nullptr, lse, kmask, D, st, &s3);
}

// o3: the copy leccr_attn_fwd_s3 wrote (read back as hi + lo for delta = dO . O); dq3 / dk3 / dv3: the G-form (hi | lo | hi) copies of the three
// gradients, each pointer at its tensor's column 0 inside segment 0 of a copy with row stride ld3 and segment width seg3 (one packed
// [rows, 3 x 3 d] copy of dqkv: dq3 = base, dk3 = base + d, dv3 = base + 2 d, seg3 = 3 d); dout has row stride ldo; bias_sum (optional,
// [3 heads hd] f32): += the column sums of [dq | dk | dv] - the packed in-projection's bias gradient, taken from the tiles as they are staged
extern "C" int leccr_attn_bwd_s3(const void* q, const void* k, const void* v, const void* o3, int64_t ldo3, int64_t sego3, const void* dout,
                                 const float* lse, const uint8_t* kmask, void* dq3, void* dk3, void* dv3, float* delta,
                                 int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                 int64_t ld3, int64_t seg3, float* bias_sum, float drop_p, uint64_t seed, void* stream) {
    LECCR_REQUIRE(q && k && v && o3 && dout && lse && dq3 && dk3 && dv3 && delta, LECCR_EINVAL, "attn_bwd_s3: null pointer");
    LECCR_REQUIRE(drop_p >= 0.f && drop_p < 1.f, LECCR_EINVAL, "attn_bwd_s3: drop_p must be in [0,1)");
    LECCR_REQUIRE(sego3 >= heads * hd && ldo3 >= 3 * sego3 && ((ldo3 | sego3) & 7) == 0, LECCR_EINVAL,
                  "attn_bwd_s3: the o copy needs sego3 >= heads * hd, ldo3 >= 3 sego3, both multiples of 8");
    AttnDims D{b, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, make_drop(drop_p, seed), drop_p > 0.f};
    const void* ptrs[8] = {q, k, v, o3, dout, dq3, dk3, dv3};
    LECCR_REQUIRE(leccr_attn_s3_takes(b, heads, Lq, Lk, hd, ldq, ldk, ldv, ld3, seg3) && ldo >= heads * hd && attn_f32_supported(D, LECCR_F32, ptrs, 8),
                  LECCR_EINVAL, "attn_bwd_s3: not a problem of the fp32 pair-form kernels (leccr_attn_s3_takes), or a pointer not 16-byte aligned");
    if (b == 0 || Lq == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PROF_ATTN_BWD, 10.0 * (double)b * heads * Lq * Lk * hd, st);
    // (the o copy may have its own layout; the kernels take one (ld3, seg) pair for the outputs and read the o copy with ldo3 / sego3 -
    //  passed through the same struct by pointing o3_in at the copy and keeping its strides beside it)
    AttnS3 s3;
    s3.o3_in = o3; s3.dq3 = dq3; s3.dk3 = dk3; s3.dv3 = dv3; s3.ld3 = ld3; s3.seg = seg3; s3.ldo3 = ldo3; s3.sego3 = sego3;
    if (bias_sum) { s3.bq = bias_sum; s3.bk = bias_sum + heads * hd; s3.bv = bias_sum + 2 * heads * hd; }
    g_last_attn_kernel = 5;
    return attn_x3_bwd(q, k, v, nullptr, dout, lse, kmask, nullptr, nullptr, nullptr, delta, D, st, &s3);
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
