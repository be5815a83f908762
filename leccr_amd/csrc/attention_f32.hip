// attention_f32.hip - fp32 flash attention on the f32 matrix-core instruction (v_mfma_f32_32x32x2_f32), head size 64.
//
// The reference-precision path (clip/model.py:185-192 `nn.MultiheadAttention` in fp32, LECCR/models/xvlm.py:88; HF BertModel's
// attention; models/attention.py:27,47,90,107): exact f32 products, f32 accumulation, no operand rounding anywhere - the scalar
// kernels of attention_generic.hip computed the same numbers at one FMA per lane and clock; here a 32 x 32 x 64 product is 32 MFMAs.
//
// Formulation ("lane = row of the softmax"): a wave owns 32 queries (forward, dQ) or 32 keys (dK / dV); the score tile is computed
// TRANSPOSED for the query-owning kernels, S^T[key][query] = K Q^T, so that
//   * a lane holds ONE query (column lane & 31) and 16 of the tile's 32 keys (accumulator registers; the other 16 sit in lane ^ 32):
//     row max / row sum are in-register reductions plus one cross-half exchange, and the per-query rescale is a per-lane scalar;
//   * the probabilities ARE the B operand of the next product as they stand: accumulator register r of lane (query, half) holds key
//     8 (r >> 2) + 4 half + (r & 3), and MFMA step r of  O^T[hd][query] += V^T[hd][key] P^T[key][query]  consumes exactly that
//     register from every lane - no LDS round trip, no cross-lane movement of P or dS.
//   The dK / dV kernel keeps S[query][key] (lane = key) for the same reason: P and dS feed dV^T = dO^T P and dK^T = Q^T dS directly.
// The two k slots of an MFMA step may be ANY two reduction indices as long as A and B agree: lane half h walks the contiguous
// head-dimension half [32 h, 32 h + 32) of its row with ds_read_b128 (operands read by rows), and the key / query of slot (h, r)
// for operands read by columns.
//
// Backward = two kernels (dQ per query block; dK, dV per key block), p recomputed from the saved log-sum-exp: 3 + 4 products.  A
// single pass would need dQ summed over key blocks (f32 atomics: 19 x the dQ bytes at 32-key tiles) or every query's dQ resident
// (148 KB per head) while Q / dO are re-streamed per key tile; at the f32 MFMA rate (2048 cycles per 32 x 32 x 64 product) the two
// extra products cost less than either.
//
// LDS tiles: 32 rows x 64 floats, row stride 68 floats (the 16 lanes of a ds_read_b128 phase hit 16 distinct 4-bank groups); the
// forward's V tile, read by columns only, uses 72 (the two lane halves' rows, 4 apart, land 32 banks apart).
#include "attention.h"
#include <stdlib.h>

namespace leccr {
namespace f32a {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr int RS = 68, RSV = 72;
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// 32 consecutive floats of a global row -> registers (8 x 16-byte loads), optionally scaled
__device__ __forceinline__ void load32(const float* __restrict__ p, float (&f)[32], float mul) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[4 * i + e] = v[e] * mul;
    }
}

// one 32 x 64 tile: thread t holds chunks e = t and t + 256 (row e >> 4, columns 4 (e & 15) ..)
struct TileRegs { f32x4 v[2]; };

__device__ __forceinline__ TileRegs tile_load(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows) {
    TileRegs t;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = (int)threadIdx.x + 256 * i;
        int64_t r = row0 + (e >> 4);
        r = r < nrows ? r : nrows - 1;              // rows past the end are clamped; their scores are masked
        t.v[i] = *reinterpret_cast<const f32x4*>(base + r * ld + 4 * (e & 15));
    }
    return t;
}

template <int STRIDE>
__device__ __forceinline__ void tile_store(float* __restrict__ lds, const TileRegs& t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = (int)threadIdx.x + 256 * i;
        *reinterpret_cast<f32x4*>(lds + (e >> 4) * STRIDE + 4 * (e & 15)) = t.v[i];
    }
}

// acc += sum_i rowtile[lane & 31][32 half + i] * f[i]   (A = the tile read by rows, B = the per-lane fragment)
__device__ __forceinline__ void mma_rows(const float* __restrict__ rowp, const float (&f)[32], f32x16_t& acc0, f32x16_t& acc1) {
#pragma unroll
    for (int i4 = 0; i4 < 8; i4 += 2) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(rowp + 4 * i4), b = *reinterpret_cast<const f32x4*>(rowp + 4 * i4 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = MFMA(a[e], f[4 * i4 + e], acc0);
            acc1 = MFMA(b[e], f[4 * i4 + 4 + e], acc1);
        }
    }
}

__device__ __forceinline__ int slot_row(int half, int r) { return 8 * (r >> 2) + 4 * half + (r & 3); }

// bits of the 32 keys / queries of a tile that are VALID (in range and not padding-masked), as a wave-uniform word
__device__ __forceinline__ uint32_t valid_bits(const uint8_t* __restrict__ mask_row, int64_t i0, int64_t n, int l32) {
    const int64_t i = i0 + l32;
    const bool ok = i < n && (mask_row == nullptr || mask_row[i] == 0);
    return (uint32_t)__ballot(ok);        // lanes 0-31 and 32-63 vote alike; the low word is the tile
}

// ------------------------------------------------------------------------------------------------ forward
template <bool DROP>
__global__ __launch_bounds__(256) void fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                  float* __restrict__ o, float* __restrict__ lse, const uint8_t* __restrict__ kmask,
                                                  AttnDims D, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * RS];
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * RSV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
    const int64_t nqb = (D.Lq + 127) / 128;
    const int64_t bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
    const int64_t bb = bh / D.heads, h = bh - bb * D.heads;
    const int64_t q0 = qb * 128 + wave * 32, qi = q0 + l32;
    const bool active = q0 < D.Lq, q_ok = qi < D.Lq;
    const int64_t qc = q_ok ? qi : D.Lq - 1;
    const int64_t row = bh * D.Lq + qi;                                 // softmax row id: lse index, dropout counter
    const float* kbase = k + bb * D.Lk * D.ldk + h * 64;
    const float* vbase = v + bb * D.Lk * D.ldv + h * 64;
    const uint8_t* mrow = kmask ? kmask + bb * D.Lk : nullptr;

    float qf[32];
    load32(q + (bb * D.Lq + qc) * D.ldq + h * 64 + half * 32, qf, scale * kLog2e);      // scores in the log2 domain
    f32x16_t o0 = zero16(), o1 = zero16();
    float m = -INFINITY, l = 0.f;

    const int nt = (int)((D.Lk + 31) / 32);
    TileRegs tk = tile_load(kbase, D.ldk, 0, D.Lk), tv = tile_load(vbase, D.ldv, 0, D.Lk);
    tile_store<RS>(Ks[0], tk);
    tile_store<RSV>(Vs[0], tv);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const int64_t k0 = (int64_t)t * 32;
        if (t + 1 < nt) {
            tk = tile_load(kbase, D.ldk, k0 + 32, D.Lk);
            tv = tile_load(vbase, D.ldv, k0 + 32, D.Lk);
        }
        if (active) {
            const uint32_t vb = valid_bits(mrow, k0, D.Lk, l32) >> (4 * half);
            f32x16_t sa = zero16(), sb = zero16();
            mma_rows(Ks[buf] + l32 * RS + half * 32, qf, sa, sb);
            float s[16], mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = (vb >> (8 * (r >> 2) + (r & 3))) & 1u;
                s[r] = ok ? sa[r] + sb[r] : -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx);
            const float mu = mn == -INFINITY ? 0.f : mn;                 // a row with no visible key so far: p = 0, nothing rescaled
            const float alpha = __builtin_amdgcn_exp2f(m - mu);
            m = mn;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - mu);
                ps += s[r];
            }
            l = l * alpha + ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            if constexpr (DROP) {       // dropout on the (to be normalised) probabilities; the row sum keeps the undropped values
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] *= drop_mul(D.drop, (uint32_t)(row * D.Lk + k0 + slot_row(half, r)));
            }
            const float* vp = Vs[buf] + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = slot_row(half, r) * RSV;
                o0 = MFMA(vp[kr], s[r], o0);
                o1 = MFMA(vp[kr + 32], s[r], o1);
            }
        }
        if (t + 1 < nt) {
            tile_store<RS>(Ks[buf ^ 1], tk);
            tile_store<RSV>(Vs[buf ^ 1], tv);
        }
        __syncthreads();
    }
    if (!active) return;
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (q_ok) {
        float* op = o + (bb * D.Lq + qi) * D.ldo + h * 64 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<f32x4*>(op + 8 * g) = f32x4{o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(op + 32 + 8 * g) = f32x4{o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
        }
        if (half == 0) lse[row] = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * kLn2 : INFINITY;     // v_log_f32 = log2
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <bool DROP>
__global__ __launch_bounds__(256) void dq_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                 const float* __restrict__ o, const float* __restrict__ dout, const float* __restrict__ lse,
                                                 const uint8_t* __restrict__ kmask, float* __restrict__ dq, float* __restrict__ delta,
                                                 AttnDims D, float scale) {
    __shared__ __attribute__((aligned(16))) float Ks[2][32 * RS];
    __shared__ __attribute__((aligned(16))) float Vs[2][32 * RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
    const int64_t nqb = (D.Lq + 127) / 128;
    const int64_t bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
    const int64_t bb = bh / D.heads, h = bh - bb * D.heads;
    const int64_t q0 = qb * 128 + wave * 32, qi = q0 + l32;
    const bool active = q0 < D.Lq, q_ok = qi < D.Lq;
    const int64_t qc = q_ok ? qi : D.Lq - 1;
    const int64_t row = bh * D.Lq + qi;
    const float* kbase = k + bb * D.Lk * D.ldk + h * 64;
    const float* vbase = v + bb * D.Lk * D.ldv + h * 64;
    const uint8_t* mrow = kmask ? kmask + bb * D.Lk : nullptr;

    float qf[32], dof[32];
    load32(q + (bb * D.Lq + qc) * D.ldq + h * 64 + half * 32, qf, scale * kLog2e);
    load32(dout + (bb * D.Lq + qc) * D.ldo + h * 64 + half * 32, dof, 1.f);
    float dl = 0.f;
    {
        const float* op = o + (bb * D.Lq + qc) * D.ldo + h * 64 + half * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 ov = *reinterpret_cast<const f32x4*>(op + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) dl = fmaf(ov[e], dof[4 * i + e], dl);
        }
    }
    dl += __shfl_xor(dl, 32, 64);                                  // delta_i = dO_i . O_i
    if (q_ok && half == 0) delta[row] = dl;                        // .. kept for the dK / dV kernel
    const float lse2 = lse[bh * D.Lq + qc] * kLog2e;               // +inf for a row without visible keys: p = 0
    f32x16_t dq0 = zero16(), dq1 = zero16();

    const int nt = (int)((D.Lk + 31) / 32);
    TileRegs tk = tile_load(kbase, D.ldk, 0, D.Lk), tv = tile_load(vbase, D.ldv, 0, D.Lk);
    tile_store<RS>(Ks[0], tk);
    tile_store<RS>(Vs[0], tv);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const int64_t k0 = (int64_t)t * 32;
        if (t + 1 < nt) {
            tk = tile_load(kbase, D.ldk, k0 + 32, D.Lk);
            tv = tile_load(vbase, D.ldv, k0 + 32, D.Lk);
        }
        if (active) {
            const uint32_t vb = valid_bits(mrow, k0, D.Lk, l32) >> (4 * half);
            f32x16_t s = zero16(), dp = zero16();
            {       // S^T = K Q^T and dP^T = V dO^T, the two chains interleaved
                const float* kp = Ks[buf] + l32 * RS + half * 32;
                const float* vp = Vs[buf] + l32 * RS + half * 32;
#pragma unroll
                for (int i4 = 0; i4 < 8; ++i4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(kp + 4 * i4), b = *reinterpret_cast<const f32x4*>(vp + 4 * i4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s = MFMA(a[e], qf[4 * i4 + e], s);
                        dp = MFMA(b[e], dof[4 * i4 + e], dp);
                    }
                }
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = (vb >> (8 * (r >> 2) + (r & 3))) & 1u;
                const float p = ok ? __builtin_amdgcn_exp2f(s[r] - lse2) : 0.f;
                float dpr = dp[r];
                if constexpr (DROP) dpr *= drop_mul(D.drop, (uint32_t)(row * D.Lk + k0 + slot_row(half, r)));
                ds[r] = p * (dpr - dl);
            }
            const float* kc = Ks[buf] + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = slot_row(half, r) * RS;
                dq0 = MFMA(kc[kr], ds[r], dq0);
                dq1 = MFMA(kc[kr + 32], ds[r], dq1);
            }
        }
        if (t + 1 < nt) {
            tile_store<RS>(Ks[buf ^ 1], tk);
            tile_store<RS>(Vs[buf ^ 1], tv);
        }
        __syncthreads();
    }
    if (!active || !q_ok) return;
    float* dp_ = dq + (bb * D.Lq + qi) * D.ldq + h * 64 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(dp_ + 8 * g) = f32x4{dq0[4 * g] * scale, dq0[4 * g + 1] * scale, dq0[4 * g + 2] * scale, dq0[4 * g + 3] * scale};
        *reinterpret_cast<f32x4*>(dp_ + 32 + 8 * g) = f32x4{dq1[4 * g] * scale, dq1[4 * g + 1] * scale, dq1[4 * g + 2] * scale, dq1[4 * g + 3] * scale};
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <bool DROP>
__global__ __launch_bounds__(256) void dkv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                  const float* __restrict__ dout, const float* __restrict__ lse,
                                                  const float* __restrict__ delta, const uint8_t* __restrict__ kmask,
                                                  float* __restrict__ dk, float* __restrict__ dv, AttnDims D, float scale) {
    __shared__ __attribute__((aligned(16))) float Qs[2][32 * RS];
    __shared__ __attribute__((aligned(16))) float Os[2][32 * RS];       // dO tiles
    __shared__ float Ls[2][32], Ds[2][32];                              // log2-domain lse and delta of the tile's queries
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
    const int64_t nkb = (D.Lk + 127) / 128;
    const int64_t bh = blockIdx.x / nkb, kb = blockIdx.x - bh * nkb;
    const int64_t bb = bh / D.heads, h = bh - bb * D.heads;
    const int64_t k0w = kb * 128 + wave * 32, ki = k0w + l32;
    const bool active = k0w < D.Lk, k_in = ki < D.Lk;
    const int64_t kc = k_in ? ki : D.Lk - 1;
    const bool k_ok = k_in && !(kmask && kmask[bb * D.Lk + kc]);
    const float* qbase = q + bb * D.Lq * D.ldq + h * 64;
    const float* obase = dout + bb * D.Lq * D.ldo + h * 64;
    const float* lbase = lse + bh * D.Lq;
    const float* dbase = delta + bh * D.Lq;

    float kf[32], vf[32];
    load32(k + (bb * D.Lk + kc) * D.ldk + h * 64 + half * 32, kf, scale * kLog2e);
    load32(v + (bb * D.Lk + kc) * D.ldv + h * 64 + half * 32, vf, 1.f);
    f32x16_t dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();

    const int nt = (int)((D.Lq + 31) / 32);
    auto side_load = [&](int64_t i0, float& lv, float& dv_) {           // (threads 0-31) lse and delta of query i0 + l32
        const int64_t i = i0 + l32 < D.Lq ? i0 + l32 : D.Lq - 1;
        lv = lbase[i] * kLog2e;
        dv_ = dbase[i];
    };
    TileRegs tq = tile_load(qbase, D.ldq, 0, D.Lq), to = tile_load(obase, D.ldo, 0, D.Lq);
    float lv = 0.f, dvv = 0.f;
    if (threadIdx.x < 32) side_load(0, lv, dvv);
    tile_store<RS>(Qs[0], tq);
    tile_store<RS>(Os[0], to);
    if (threadIdx.x < 32) { Ls[0][l32] = lv; Ds[0][l32] = dvv; }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const int64_t i0 = (int64_t)t * 32;
        if (t + 1 < nt) {
            tq = tile_load(qbase, D.ldq, i0 + 32, D.Lq);
            to = tile_load(obase, D.ldo, i0 + 32, D.Lq);
            if (threadIdx.x < 32) side_load(i0 + 32, lv, dvv);
        }
        if (active) {
            f32x16_t s = zero16(), dp = zero16();
            {       // S = Q K^T and dP = dO V^T (lane = key, registers = queries)
                const float* qp = Qs[buf] + l32 * RS + half * 32;
                const float* op = Os[buf] + l32 * RS + half * 32;
#pragma unroll
                for (int i4 = 0; i4 < 8; ++i4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 4 * i4), b = *reinterpret_cast<const f32x4*>(op + 4 * i4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s = MFMA(a[e], kf[4 * i4 + e], s);
                        dp = MFMA(b[e], vf[4 * i4 + e], dp);
                    }
                }
            }
            float pd[16], ds[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[buf][8 * g + 4 * half]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&Ds[buf][8 * g + 4 * half]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const int64_t qi = i0 + 8 * g + 4 * half + e;
                    const bool ok = k_ok && qi < D.Lq;
                    const float p = ok ? __builtin_amdgcn_exp2f(s[r] - l4[e]) : 0.f;
                    float dm = 1.f;
                    if constexpr (DROP) dm = drop_mul(D.drop, (uint32_t)((bh * D.Lq + qi) * D.Lk + ki));
                    pd[r] = p * dm;                                 // dV uses the dropped probabilities
                    ds[r] = p * (dm * dp[r] - d4[e]);
                }
            }
            const float* oc = Os[buf] + l32;
            const float* qc_ = Qs[buf] + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = slot_row(half, r) * RS;
                dv0 = MFMA(oc[qr], pd[r], dv0);
                dk0 = MFMA(qc_[qr], ds[r], dk0);
                dv1 = MFMA(oc[qr + 32], pd[r], dv1);
                dk1 = MFMA(qc_[qr + 32], ds[r], dk1);
            }
        }
        if (t + 1 < nt) {
            tile_store<RS>(Qs[buf ^ 1], tq);
            tile_store<RS>(Os[buf ^ 1], to);
            if (threadIdx.x < 32) { Ls[buf ^ 1][l32] = lv; Ds[buf ^ 1][l32] = dvv; }
        }
        __syncthreads();
    }
    if (!active || !k_in) return;
    float* dkp = dk + (bb * D.Lk + ki) * D.ldk + h * 64 + 4 * half;
    float* dvp = dv + (bb * D.Lk + ki) * D.ldv + h * 64 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(dkp + 8 * g) = f32x4{dk0[4 * g] * scale, dk0[4 * g + 1] * scale, dk0[4 * g + 2] * scale, dk0[4 * g + 3] * scale};
        *reinterpret_cast<f32x4*>(dkp + 32 + 8 * g) = f32x4{dk1[4 * g] * scale, dk1[4 * g + 1] * scale, dk1[4 * g + 2] * scale, dk1[4 * g + 3] * scale};
        *reinterpret_cast<f32x4*>(dvp + 8 * g) = f32x4{dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]};
        *reinterpret_cast<f32x4*>(dvp + 32 + 8 * g) = f32x4{dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]};
    }
}

}  // namespace f32a

// f32, head size 64, 16-byte aligned rows, no causal mask (the 77-token CLIP text tower stays on the generic kernels).
// LECCR_ATTN_F32_MFMA=0: the scalar generic kernels instead (correctness yardstick).
bool attn_f32_supported(const AttnDims& D, int dtype, const void* const* ptrs, int nptrs) {
    static const bool on = [] { const char* e = getenv("LECCR_ATTN_F32_MFMA"); return !(e && e[0] == '0'); }();
    if (!on || dtype != LECCR_F32 || D.hd != 64 || D.causal) return false;
    if ((D.ldq | D.ldk | D.ldv | D.ldo) & 3) return false;
    for (int i = 0; i < nptrs; ++i)
        if (reinterpret_cast<uintptr_t>(ptrs[i]) & 15) return false;
    return D.b * D.heads * ((D.Lq + 127) / 128) < (1ll << 31) && D.b * D.heads * ((D.Lk + 127) / 128) < (1ll << 31);
}

int attn_f32_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask, const AttnDims& D, hipStream_t st) {
    const float scale = 1.f / sqrtf((float)D.hd);
    const dim3 grid((unsigned)(D.b * D.heads * ((D.Lq + 127) / 128)));
    if (D.dropout)
        hipLaunchKernelGGL(f32a::fwd_kernel<true>, grid, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (float*)o, lse, kmask, D, scale);
    else
        hipLaunchKernelGGL(f32a::fwd_kernel<false>, grid, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (float*)o, lse, kmask, D, scale);
    return check_launch("attn_f32_fwd");
}

int attn_f32_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, const uint8_t* kmask,
                 void* dq, void* dk, void* dv, float* delta, const AttnDims& D, hipStream_t st) {
    const float scale = 1.f / sqrtf((float)D.hd);
    const dim3 gq((unsigned)(D.b * D.heads * ((D.Lq + 127) / 128))), gk((unsigned)(D.b * D.heads * ((D.Lk + 127) / 128)));
#define ARGS_Q (const float*)q, (const float*)k, (const float*)v, (const float*)o, (const float*)dout, lse, kmask, (float*)dq, delta, D, scale
#define ARGS_K (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse, (const float*)delta, kmask, (float*)dk, (float*)dv, D, scale
    if (D.dropout) {
        hipLaunchKernelGGL(f32a::dq_kernel<true>, gq, dim3(256), 0, st, ARGS_Q);
        hipLaunchKernelGGL(f32a::dkv_kernel<true>, gk, dim3(256), 0, st, ARGS_K);
    } else {
        hipLaunchKernelGGL(f32a::dq_kernel<false>, gq, dim3(256), 0, st, ARGS_Q);
        hipLaunchKernelGGL(f32a::dkv_kernel<false>, gk, dim3(256), 0, st, ARGS_K);
    }
#undef ARGS_Q
#undef ARGS_K
    return check_launch("attn_f32_bwd");
}

}  // namespace leccr
