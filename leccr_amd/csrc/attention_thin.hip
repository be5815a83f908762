// attention_thin.hip - bf16 attention with a THIN side (at most 8 keys, or at most 8 queries, per batch element), any head size.
//
// LECCR's caption interaction (models/model_retrieval_caption.py:76-84 through models/attention.py:24-58) runs nn.MultiheadAttention(d, 8)
// between the image tokens and FOUR caption queries: the image tokens attend to 4 keys (crossattn), the 4 queries attend to 577 / 730 image
// tokens (crossattn2) and to 64 caption tokens (crossattn_query).  At d = 512 (ViT-B) the heads are 64 wide and the MFMA flash kernels take
// them; at d = 768 (ViT-L/14's embedding, BASELINE configs[4]) they are 96 wide and fell to the scalar generic kernels: 17.8 ms of a
// 420 ms step for ~6 GFLOP.  There is no matrix-core work here - 4 scores per row and head - so these are HBM streams over the long side:
//   thin-K: a wave owns one query row of ALL heads (lane l = columns EPL l .. EPL l + EPL - 1 of the d-wide row, EPL = d / 64, so a head is
//           64 / heads adjacent lanes), K / V of the batch element sit in LDS as f32, scores are reduced over a head's lanes with shuffles;
//           the backward is ONE pass over q / dO that writes dq and accumulates dK / dV in registers (delta = sum_k p~_k dP_k needs no O).
//   thin-Q: a workgroup owns one batch element, its waves stride over the key rows; forward in two phases over K then V with the scores of
//           all heads in LDS (exact softmax, no online rescaling); backward one pass over K / V that writes dK / dV rows and accumulates dQ.
// Same conventions as attention_generic.hip: lse = max + log(sum) of the scaled scores (+inf for a fully masked row, o = 0), dropout on the
// normalised probabilities with keep(b, h, q, k) = hash(seed, ((b heads + h) Lq + q) Lk + k), kmask 1 = padded key.
#include "attention.h"

namespace leccr {
namespace thin {

constexpr int MAXT = 8;       // thin side: up to 8 rows

template <int EPL>
__device__ __forceinline__ void load_part(const bf16_t* __restrict__ p, float (&x)[EPL]) {      // EPL consecutive bf16 -> f32 (8-byte loads)
#pragma unroll
    for (int c = 0; c < EPL / 4; ++c) {
        const uint2 u = *reinterpret_cast<const uint2*>(p + 4 * c);
        x[4 * c] = __uint_as_float(u.x << 16);
        x[4 * c + 1] = __uint_as_float(u.x & 0xffff0000u);
        x[4 * c + 2] = __uint_as_float(u.y << 16);
        x[4 * c + 3] = __uint_as_float(u.y & 0xffff0000u);
    }
}
template <int EPL>
__device__ __forceinline__ void store_part(bf16_t* __restrict__ p, const float (&x)[EPL]) {
#pragma unroll
    for (int c = 0; c < EPL / 4; ++c) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
        bf16x4_t w = {(bf16_t)x[4 * c], (bf16_t)x[4 * c + 1], (bf16_t)x[4 * c + 2], (bf16_t)x[4 * c + 3]};
        *reinterpret_cast<bf16x4_t*>(p + 4 * c) = w;
    }
}
// sum over the `lph` (a power of two) adjacent lanes of a head
__device__ __forceinline__ float head_sum(float v, int lph) {
    for (int o = 1; o < lph; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

struct TP {
    const bf16_t *q, *k, *v, *o, *dout;
    bf16_t *out, *dq, *dk, *dv;
    float* lse;
    float* ws;                 // thin-K backward with row splits: f32 [2][b][Lk][d] partial dK / dV (zeroed by the launcher)
    const uint8_t* kmask;
    int64_t b, heads, Lq, Lk, ldq, ldk, ldv, ldo;      // heads: those of THIS column group (a launch covers d' = 64 EPL columns = whole heads)
    int64_t heads_total;                               // .. of the problem; head0: first head of the group (lse rows, dropout indices)
    int head0;
    int lph, nsplit;
    float scale;
    DropCfg drop;
    bool dropout;
};

// ------------------------------------------------------------------------------------------------ thin-K forward
// grid (chunks of 64 query rows, b); 4 waves, wave w takes rows w, w + 4, ..
template <int EPL, int KB>
__global__ __launch_bounds__(256) void thin_k_fwd_kernel(TP p) {
    extern __shared__ float sm[];
    const int d = 64 * EPL, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t bb = blockIdx.y;
    float* Ks = sm;
    float* Vs = sm + KB * d;
    for (int i = tid; i < KB * d; i += 256) {          // rows Lk .. KB - 1: zeros (their probabilities are 0, 0 x garbage would not be)
        const int j = i / d, c = i - j * d;
        Ks[i] = j < p.Lk ? (float)p.k[(bb * p.Lk + j) * p.ldk + c] : 0.f;
        Vs[i] = j < p.Lk ? (float)p.v[(bb * p.Lk + j) * p.ldv + c] : 0.f;
    }
    uint32_t dead = 0;
    for (int j = 0; j < KB; ++j)
        if (j >= p.Lk || (p.kmask && p.kmask[bb * p.Lk + (j < p.Lk ? j : 0)])) dead |= 1u << j;
    __syncthreads();
    const int h = lane / p.lph;
    const int64_t r0 = (int64_t)blockIdx.x * 64, r1 = r0 + 64 < p.Lq ? r0 + 64 : p.Lq;
    for (int64_t i = r0 + wave; i < r1; i += 4) {
        float qv[EPL];
        load_part<EPL>(p.q + (bb * p.Lq + i) * p.ldq + lane * EPL, qv);
        float s[KB], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) a = fmaf(qv[e], Ks[j * d + lane * EPL + e], a);
            a = head_sum(a, p.lph) * p.scale;
            s[j] = (dead >> j) & 1 ? -INFINITY : a;
            mx = fmaxf(mx, s[j]);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            s[j] = mx == -INFINITY ? 0.f : __expf(s[j] - mx);
            l += s[j];
        }
        const float inv = l > 0.f ? 1.f / l : 0.f;
        const int64_t row = (bb * p.heads_total + p.head0 + h) * p.Lq + i;
        if ((lane & (p.lph - 1)) == 0) p.lse[row] = l > 0.f ? mx + __logf(l) : INFINITY;
        float ov[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) ov[e] = 0.f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            float pj = s[j] * inv;
            if (p.dropout) pj *= drop_mul(p.drop, (uint32_t)(row * p.Lk + j));
#pragma unroll
            for (int e = 0; e < EPL; ++e) ov[e] = fmaf(pj, Vs[j * d + lane * EPL + e], ov[e]);
        }
        store_part<EPL>(p.out + (bb * p.Lq + i) * p.ldo + lane * EPL, ov);
    }
}

// ------------------------------------------------------------------------------------------------ thin-K backward
// grid (nsplit row chunks, b); one pass over q / dO: dq rows written, dK / dV of the chunk summed (registers -> LDS -> memory)
template <int EPL, int KB>
__global__ __launch_bounds__(256) void thin_k_bwd_kernel(TP p) {
    extern __shared__ float sm[];
    const int d = 64 * EPL, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t bb = blockIdx.y;
    float* Ks = sm;
    float* Vs = sm + KB * d;
    float* dKs = sm + 2 * KB * d;
    float* dVs = sm + 3 * KB * d;
    for (int i = tid; i < KB * d; i += 256) {
        const int j = i / d, c = i - j * d;
        Ks[i] = j < p.Lk ? (float)p.k[(bb * p.Lk + j) * p.ldk + c] : 0.f;
        Vs[i] = j < p.Lk ? (float)p.v[(bb * p.Lk + j) * p.ldv + c] : 0.f;
        dKs[i] = 0.f;
        dVs[i] = 0.f;
    }
    uint32_t dead = 0;
    for (int j = 0; j < KB; ++j)
        if (j >= p.Lk || (p.kmask && p.kmask[bb * p.Lk + (j < p.Lk ? j : 0)])) dead |= 1u << j;
    __syncthreads();
    const int h = lane / p.lph;
    const int64_t per = (p.Lq + p.nsplit - 1) / p.nsplit;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < p.Lq ? r0 + per : p.Lq;
    float adk[KB][EPL], adv[KB][EPL];
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) { adk[j][e] = 0.f; adv[j][e] = 0.f; }
    for (int64_t i = r0 + wave; i < r1; i += 4) {
        float qv[EPL], gv[EPL];
        load_part<EPL>(p.q + (bb * p.Lq + i) * p.ldq + lane * EPL, qv);
        load_part<EPL>(p.dout + (bb * p.Lq + i) * p.ldo + lane * EPL, gv);
        float s[KB], dp[KB], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            float a = 0.f, g = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                a = fmaf(qv[e], Ks[j * d + lane * EPL + e], a);
                g = fmaf(gv[e], Vs[j * d + lane * EPL + e], g);
            }
            a = head_sum(a, p.lph) * p.scale;
            dp[j] = head_sum(g, p.lph);
            s[j] = (dead >> j) & 1 ? -INFINITY : a;
            mx = fmaxf(mx, s[j]);
        }
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            s[j] = mx == -INFINITY ? 0.f : __expf(s[j] - mx);
            l += s[j];
        }
        const float inv = l > 0.f ? 1.f / l : 0.f;
        const int64_t row = (bb * p.heads_total + p.head0 + h) * p.Lq + i;
        float dm[KB], delta = 0.f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            s[j] *= inv;                                                         // p_j
            dm[j] = p.dropout ? drop_mul(p.drop, (uint32_t)(row * p.Lk + j)) : 1.f;
            delta = fmaf(s[j] * dm[j], dp[j], delta);                            // dO.O = sum_j p~_j dP_j
        }
        float dqv[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) dqv[e] = 0.f;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const float ds = s[j] * (dm[j] * dp[j] - delta), pt = s[j] * dm[j];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                dqv[e] = fmaf(ds, Ks[j * d + lane * EPL + e], dqv[e]);
                adk[j][e] = fmaf(ds, qv[e], adk[j][e]);
                adv[j][e] = fmaf(pt, gv[e], adv[j][e]);
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) dqv[e] *= p.scale;
        store_part<EPL>(p.dq + (bb * p.Lq + i) * p.ldq + lane * EPL, dqv);
    }
#pragma unroll
    for (int j = 0; j < KB; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            atomicAdd(&dKs[j * d + lane * EPL + e], adk[j][e] * p.scale);
            atomicAdd(&dVs[j * d + lane * EPL + e], adv[j][e]);
        }
    __syncthreads();
    if (p.nsplit == 1) {
        for (int i = tid; i < (int)p.Lk * d; i += 256) {
            const int j = i / d, c = i - j * d;
            p.dk[(bb * p.Lk + j) * p.ldk + c] = (bf16_t)dKs[i];
            p.dv[(bb * p.Lk + j) * p.ldv + c] = (bf16_t)dVs[i];
        }
    } else {
        float* wk = p.ws + bb * p.Lk * d;
        float* wv = p.ws + (p.b + bb) * p.Lk * d;
        for (int i = tid; i < (int)p.Lk * d; i += 256) {
            atomicAdd(wk + i, dKs[i]);
            atomicAdd(wv + i, dVs[i]);
        }
    }
}

__global__ void thin_k_finish_kernel(TP p, int d) {          // f32 partial sums -> bf16 dk / dv
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, n = p.b * p.Lk * d;
    if (i >= n) return;
    const int64_t row = i / d, c = i - row * d;
    p.dk[row * p.ldk + c] = (bf16_t)p.ws[i];
    p.dv[row * p.ldv + c] = (bf16_t)p.ws[n + i];
}

// ------------------------------------------------------------------------------------------------ thin-Q forward
// grid b; NW waves stride over the key rows.  LDS: S[QB][heads][Lk] (scores -> exp), Os[QB][d], inv[QB][heads]
template <int EPL, int QB, int NW>
__global__ __launch_bounds__(NW * 64) void thin_q_fwd_kernel(TP p) {
    extern __shared__ float sm[];
    const int d = 64 * EPL, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t bb = blockIdx.x;
    const int Lk = (int)p.Lk, heads = (int)p.heads;
    float* S = sm;                                  // [(qi * heads + h) * Lk + j]
    float* Os = sm + QB * heads * Lk;
    float* invs = Os + QB * d;
    for (int i = tid; i < QB * d; i += NW * 64) Os[i] = 0.f;
    const int h = lane / p.lph;
    const bool head_lane = (lane & (p.lph - 1)) == 0;
    {
        float qv[QB][EPL];
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            if (qi < p.Lq) load_part<EPL>(p.q + (bb * p.Lq + qi) * p.ldq + lane * EPL, qv[qi]);
            else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) qv[qi][e] = 0.f;
            }
        }
        for (int j = wave; j < Lk; j += NW) {
            float kv[EPL];
            load_part<EPL>(p.k + (bb * p.Lk + j) * p.ldk + lane * EPL, kv);
            const bool masked = p.kmask && p.kmask[bb * p.Lk + j];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) a = fmaf(qv[qi][e], kv[e], a);
                a = head_sum(a, p.lph) * p.scale;
                if (head_lane && qi < p.Lq) S[(qi * heads + h) * Lk + j] = masked ? -INFINITY : a;
            }
        }
    }
    __syncthreads();
    for (int r = wave; r < (int)p.Lq * heads; r += NW) {            // row (qi, h): exact softmax statistics
        float* sr = S + r * Lk;
        float mx = -INFINITY;
        for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, sr[j]);
        mx = wave_max(mx);
        float l = 0.f;
        for (int j = lane; j < Lk; j += 64) {
            const float e = mx == -INFINITY ? 0.f : __expf(sr[j] - mx);
            sr[j] = e;
            l += e;
        }
        l = wave_sum(l);
        if (lane == 0) {
            const int qi = r / heads, hh = r - qi * heads;
            invs[r] = l > 0.f ? 1.f / l : 0.f;
            p.lse[(bb * p.heads_total + p.head0 + hh) * p.Lq + qi] = l > 0.f ? mx + __logf(l) : INFINITY;
        }
    }
    __syncthreads();
    {
        float ov[QB][EPL];
#pragma unroll
        for (int qi = 0; qi < QB; ++qi)
#pragma unroll
            for (int e = 0; e < EPL; ++e) ov[qi][e] = 0.f;
        for (int j = wave; j < Lk; j += NW) {
            float vv[EPL];
            load_part<EPL>(p.v + (bb * p.Lk + j) * p.ldv + lane * EPL, vv);
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                if (qi < p.Lq) {
                    float pj = S[(qi * heads + h) * Lk + j] * invs[qi * heads + h];
                    if (p.dropout) pj *= drop_mul(p.drop, (uint32_t)(((bb * p.heads_total + p.head0 + h) * p.Lq + qi) * p.Lk + j));
#pragma unroll
                    for (int e = 0; e < EPL; ++e) ov[qi][e] = fmaf(pj, vv[e], ov[qi][e]);
                }
            }
        }
#pragma unroll
        for (int qi = 0; qi < QB; ++qi)
#pragma unroll
            for (int e = 0; e < EPL; ++e) atomicAdd(&Os[qi * d + lane * EPL + e], ov[qi][e]);
    }
    __syncthreads();
    for (int i = tid; i < (int)p.Lq * d; i += NW * 64) {
        const int qi = i / d, c = i - qi * d;
        p.out[(bb * p.Lq + qi) * p.ldo + c] = (bf16_t)Os[i];
    }
}

// ------------------------------------------------------------------------------------------------ thin-Q backward
// grid b; NW waves stride over the key rows: dK / dV rows written, dQ summed (registers -> LDS -> memory)
template <int EPL, int QB, int NW>
__global__ __launch_bounds__(NW * 64) void thin_q_bwd_kernel(TP p) {
    extern __shared__ float sm[];
    const int d = 64 * EPL, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t bb = blockIdx.x;
    float* dQs = sm;                                // [QB][d]
    for (int i = tid; i < QB * d; i += NW * 64) dQs[i] = 0.f;
    const int h = lane / p.lph;
    float qv[QB][EPL], gv[QB][EPL], L[QB], delta[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        if (qi < p.Lq) {
            float ov[EPL];
            load_part<EPL>(p.q + (bb * p.Lq + qi) * p.ldq + lane * EPL, qv[qi]);
            load_part<EPL>(p.dout + (bb * p.Lq + qi) * p.ldo + lane * EPL, gv[qi]);
            load_part<EPL>(p.o + (bb * p.Lq + qi) * p.ldo + lane * EPL, ov);
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) a = fmaf(gv[qi][e], ov[e], a);
            delta[qi] = head_sum(a, p.lph);
            L[qi] = p.lse[(bb * p.heads_total + p.head0 + h) * p.Lq + qi];
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) { qv[qi][e] = 0.f; gv[qi][e] = 0.f; }
            delta[qi] = 0.f;
            L[qi] = INFINITY;
        }
    }
    float adq[QB][EPL];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int e = 0; e < EPL; ++e) adq[qi][e] = 0.f;
    __syncthreads();
    for (int64_t j = wave; j < p.Lk; j += NW) {
        float kv[EPL], vv[EPL], dkv[EPL], dvv[EPL];
        load_part<EPL>(p.k + (bb * p.Lk + j) * p.ldk + lane * EPL, kv);
        load_part<EPL>(p.v + (bb * p.Lk + j) * p.ldv + lane * EPL, vv);
        const bool masked = p.kmask && p.kmask[bb * p.Lk + j];
#pragma unroll
        for (int e = 0; e < EPL; ++e) { dkv[e] = 0.f; dvv[e] = 0.f; }
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            float a = 0.f, g = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                a = fmaf(qv[qi][e], kv[e], a);
                g = fmaf(gv[qi][e], vv[e], g);
            }
            a = head_sum(a, p.lph) * p.scale;
            g = head_sum(g, p.lph);
            const float pj = masked ? 0.f : __expf(a - L[qi]);
            const float dm = p.dropout ? drop_mul(p.drop, (uint32_t)(((bb * p.heads_total + p.head0 + h) * p.Lq + qi) * p.Lk + j)) : 1.f;
            const float ds = pj * (dm * g - delta[qi]), pt = pj * dm;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                dkv[e] = fmaf(ds, qv[qi][e], dkv[e]);
                dvv[e] = fmaf(pt, gv[qi][e], dvv[e]);
                adq[qi][e] = fmaf(ds, kv[e], adq[qi][e]);
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) dkv[e] *= p.scale;
        store_part<EPL>(p.dk + (bb * p.Lk + j) * p.ldk + lane * EPL, dkv);
        store_part<EPL>(p.dv + (bb * p.Lk + j) * p.ldv + lane * EPL, dvv);
    }
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int e = 0; e < EPL; ++e) atomicAdd(&dQs[qi * d + lane * EPL + e], adq[qi][e] * p.scale);
    __syncthreads();
    for (int i = tid; i < (int)p.Lq * d; i += NW * 64) {
        const int qi = i / d, c = i - qi * d;
        p.dq[(bb * p.Lq + qi) * p.ldq + c] = (bf16_t)dQs[i];
    }
}

template <typename KFn>
static void set_smem(KFn k, size_t) {           // once per instantiation: up to the CU's whole LDS
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
}

static bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace thin

// A problem of width d = heads * hd runs as d / d' column groups of d' in {1024, 768, 512} columns, each a whole number of heads (one launch
// per group with shifted base pointers: the caption interaction at d = 768 is one group of 8 x 96, the MSR-VTT video model's d = 4096 with
// 512-wide heads four groups of 2 heads).  thin side <= 4 rows: any d'; <= 8 rows: d' = 512 only (the backward's accumulators).
static bool thin_plan(const AttnDims& D, int thin_rows, int* dprime, int* groups) {
    const int64_t d = D.heads * D.hd;
    static const int opts[3] = {1024, 768, 512};
    for (int o = (thin_rows <= 4 ? 0 : 2); o < 3; ++o) {
        const int dp = opts[o];
        if (d % dp || dp % D.hd) continue;
        const int hp = (int)(dp / D.hd);
        if (hp > 64 || (64 % hp) != 0) continue;
        // several groups = several launches: only worth it when the long side has rows to stream (the MSR-VTT video model's 8 frames x 4
        // queries at d = 4096 run 7 % faster on the generic kernels: 33.7 against 36.1 ms per step)
        if (d / dp > 1 && (D.Lq > D.Lk ? D.Lq : D.Lk) < 128) return false;
        *dprime = dp;
        *groups = (int)(d / dp);
        return true;
    }
    return false;
}

// which thin form (if any) takes this problem: 1 = thin-K (Lk <= 8), 2 = thin-Q (Lq <= 8), 0 = neither
int attn_thin_kind(const AttnDims& D, int dtype) {
    static const int enabled = [] { const char* e = getenv("LECCR_ATTN_THIN"); return (e && e[0] == '0') ? 0 : 1; }();
    if (!enabled || dtype != LECCR_BF16) return 0;
    if ((D.ldq & 3) || (D.ldk & 3) || (D.ldv & 3) || (D.ldo & 3) || (D.hd & 3)) return 0;
    int dp, groups;
    if (D.Lk <= thin::MAXT && thin_plan(D, (int)D.Lk, &dp, &groups)) return 1;
    if (D.Lq <= thin::MAXT && thin_plan(D, (int)D.Lq, &dp, &groups)) {
        const int64_t qb = D.Lq <= 4 ? 4 : 8, hp = dp / D.hd;
        if ((qb * hp * D.Lk + qb * dp + qb * hp) * 4 <= 144 * 1024) return 2;
    }
    return 0;
}

#define THIN_DISPATCH(EPLV, TB, CALL, WITH_12_8)                                                           \
    do {                                                                                                   \
        if (EPLV == 8 && TB == 4) { CALL(8, 4); } else if (EPLV == 8) { CALL(8, 8); }                      \
        else if (EPLV == 12 && TB == 4) { CALL(12, 4); } else if (EPLV == 12) { WITH_12_8(CALL); }         \
        else { CALL(16, 4); }                                                                              \
    } while (0)
#define THIN_YES(CALL) CALL(12, 8)
#define THIN_NO(CALL) return LECCR_EINVAL

static thin::TP thin_params(const AttnDims& D, int dprime) {
    thin::TP p{};
    p.b = D.b; p.heads = dprime / D.hd; p.heads_total = D.heads; p.head0 = 0; p.Lq = D.Lq; p.Lk = D.Lk;
    p.ldq = D.ldq; p.ldk = D.ldk; p.ldv = D.ldv; p.ldo = D.ldo;
    p.lph = (int)(64 / p.heads); p.nsplit = 1; p.scale = 1.f / sqrtf((float)D.hd); p.drop = D.drop; p.dropout = D.dropout;
    return p;
}

int attn_thin_fwd(int kind, const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                  const AttnDims& D, hipStream_t st) {
    using namespace thin;
    if (!(al8(q) && al8(k) && al8(v) && al8(o))) { set_error("attn_fwd(thin): q/k/v/o must be 8-byte aligned"); return LECCR_EALIGN; }
    int d = 0, groups = 0;
    if (!thin_plan(D, (int)(kind == 1 ? D.Lk : D.Lq), &d, &groups)) { set_error("attn_fwd(thin): no column-group plan"); return LECCR_EINVAL; }
    const int epl = d / 64;
    for (int g = 0; g < groups; ++g) {
        TP p = thin_params(D, d);
        const int64_t c0 = (int64_t)g * d;
        p.head0 = g * (int)p.heads;
        p.q = (const bf16_t*)q + c0; p.k = (const bf16_t*)k + c0; p.v = (const bf16_t*)v + c0; p.out = (bf16_t*)o + c0; p.lse = lse; p.kmask = kmask;
        if (kind == 1) {
            const int kb = D.Lk <= 4 ? 4 : 8;
            const size_t shm = (size_t)2 * kb * d * sizeof(float);
#define CALL(E, T)                                                                                                       \
            do { set_smem(thin_k_fwd_kernel<E, T>, shm);                                                                 \
                 hipLaunchKernelGGL((thin_k_fwd_kernel<E, T>), dim3((unsigned)cdiv(D.Lq, 64), (unsigned)D.b), dim3(256), shm, st, p); } while (0)
            THIN_DISPATCH(epl, kb, CALL, THIN_NO);
#undef CALL
        } else {
            const int qb = D.Lq <= 4 ? 4 : 8;
            const size_t shm = ((size_t)qb * p.heads * D.Lk + (size_t)qb * d + (size_t)qb * p.heads) * sizeof(float);
#define CALL(E, T)                                                                                                       \
            do { constexpr int NW = (T == 4 && E <= 12) ? 16 : 8;                                                        \
                 set_smem(thin_q_fwd_kernel<E, T, NW>, shm);                                                             \
                 hipLaunchKernelGGL((thin_q_fwd_kernel<E, T, NW>), dim3((unsigned)D.b), dim3(NW * 64), shm, st, p); } while (0)
            THIN_DISPATCH(epl, qb, CALL, THIN_NO);
#undef CALL
        }
    }
    return check_launch(kind == 1 ? "thin_k_fwd" : "thin_q_fwd");
}

int attn_thin_bwd(int kind, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                  const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, const AttnDims& D, hipStream_t st) {
    using namespace thin;
    if (!(al8(q) && al8(k) && al8(v) && al8(o) && al8(dout) && al8(dq) && al8(dk) && al8(dv))) {
        set_error("attn_bwd(thin): tensors must be 8-byte aligned");
        return LECCR_EALIGN;
    }
    int d = 0, groups = 0;
    if (!thin_plan(D, (int)(kind == 1 ? D.Lk : D.Lq), &d, &groups)) { set_error("attn_bwd(thin): no column-group plan"); return LECCR_EINVAL; }
    const int epl = d / 64;
    // thin-K: long query sets are cut into row chunks (more workgroups than batch elements in flight); their dK / dV partial sums meet in
    // the caller's f32 workspace (`delta`: 2 b heads roundup(Lq, 64) floats) when 2 b Lk d_total of them fit (a slice per column group)
    const int64_t ws_have = 2 * D.b * D.heads * (cdiv(D.Lq, 64) * 64), ws_group = 2 * D.b * D.Lk * d, ws_need = ws_group * groups;
    const int nsplit = (kind == 1 && D.Lq >= 256 && ws_need <= ws_have) ? 8 : 1;
    if (nsplit > 1 && hipMemsetAsync(delta, 0, sizeof(float) * ws_need, st) != hipSuccess) return check_launch("thin_k_bwd memset");
    for (int g = 0; g < groups; ++g) {
        TP p = thin_params(D, d);
        const int64_t c0 = (int64_t)g * d;
        p.head0 = g * (int)p.heads;
        p.q = (const bf16_t*)q + c0; p.k = (const bf16_t*)k + c0; p.v = (const bf16_t*)v + c0; p.o = (const bf16_t*)o + c0;
        p.dout = (const bf16_t*)dout + c0; p.dq = (bf16_t*)dq + c0; p.dk = (bf16_t*)dk + c0; p.dv = (bf16_t*)dv + c0;
        p.lse = const_cast<float*>(lse); p.kmask = kmask;
        if (kind == 1) {
            const int kb = D.Lk <= 4 ? 4 : 8;
            p.nsplit = nsplit;
            p.ws = delta + g * ws_group;
            const size_t shm = (size_t)4 * kb * d * sizeof(float);
#define CALL(E, T)                                                                                                       \
            do { set_smem(thin_k_bwd_kernel<E, T>, shm);                                                                 \
                 hipLaunchKernelGGL((thin_k_bwd_kernel<E, T>), dim3((unsigned)p.nsplit, (unsigned)D.b), dim3(256), shm, st, p); } while (0)
            THIN_DISPATCH(epl, kb, CALL, THIN_NO);
#undef CALL
            if (p.nsplit > 1)
                hipLaunchKernelGGL(thin_k_finish_kernel, dim3((unsigned)cdiv(D.b * D.Lk * d, 256)), dim3(256), 0, st, p, d);
        } else {
            const int qb = D.Lq <= 4 ? 4 : 8;
            const size_t shm = (size_t)qb * d * sizeof(float);
#define CALL(E, T)                                                                                                       \
            do { constexpr int NW = (E == 16 || T == 8) ? 4 : 8;       /* 128 q / dO registers: one wave per SIMD */             \
                 hipLaunchKernelGGL((thin_q_bwd_kernel<E, T, NW>), dim3((unsigned)D.b), dim3(NW * 64), shm, st, p); } while (0)
            THIN_DISPATCH(epl, qb, CALL, THIN_NO);
#undef CALL
        }
    }
    return check_launch(kind == 1 ? "thin_k_bwd" : "thin_q_bwd");
}

}  // namespace leccr
