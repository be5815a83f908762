// attention_generic.hip - reference-grade attention for any head size / dtype (f32 arithmetic, one wave per row).
// Used by the f32 parity path and for head sizes the MFMA flash kernels (attention_mfma.hip) do not cover.
//   fwd : o_i = softmax_j(scale * q_i.k_j + mask_j) v_j ; lse_i = logsumexp_j
//   bwd : delta_i = dO_i.O_i ; dS_ij = p_ij (dO_i.v_j - delta_i) ; dq_i = scale sum_j dS_ij k_j ;
//         dk_j = scale sum_i dS_ij q_i ; dv_j = sum_i p_ij dO_i        (p recomputed from lse, never stored)
#include "attention.h"
#include <stdlib.h>

namespace leccr {

// dynamic LDS per wave: hd floats (the row's q or k) + L floats (p) [+ L floats (dS)]
template <typename T>
__global__ __launch_bounds__(256) void attn_generic_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ v, T* __restrict__ o,
                                                               float* __restrict__ lse,
                                                               const uint8_t* __restrict__ kmask, AttnDims D,
                                                               float scale) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;  // over b*heads*Lq
    if (row >= D.b * D.heads * D.Lq) return;
    const int64_t i = row % D.Lq, h = (row / D.Lq) % D.heads, bb = row / (D.Lq * D.heads);
    float* qs = sm + (size_t)wave * (D.hd + D.Lk);
    float* ps = qs + D.hd;
    const T* qp = q + (bb * D.Lq + i) * D.ldq + h * D.hd;
    for (int d = lane; d < D.hd; d += 64) qs[d] = to_f32(qp[d]);
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int64_t j = lane; j < D.Lk; j += 64) {
        float s = -INFINITY;
        if ((!kmask || !kmask[bb * D.Lk + j]) && !(D.causal && j > i)) {
            const T* kp = k + (bb * D.Lk + j) * D.ldk + h * D.hd;
            float acc = 0.f;
            for (int d = 0; d < D.hd; ++d) acc = fmaf(qs[d], to_f32(kp[d]), acc);
            s = acc * scale;
        }
        ps[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float l = 0.f;
    for (int64_t j = lane; j < D.Lk; j += 64) {
        const float p = mx == -INFINITY ? 0.f : __expf(ps[j] - mx);
        ps[j] = p;
        l += p;
    }
    l = wave_sum(l);
    __builtin_amdgcn_wave_barrier();
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (lane == 0) lse[row] = l > 0.f ? mx + __logf(l) : INFINITY;
    if (D.dropout) {      // dropout on the normalised probabilities; the row sum above used the undropped values
        for (int64_t j = lane; j < D.Lk; j += 64) ps[j] *= drop_mul(D.drop, (uint32_t)(row * D.Lk + j));
        __builtin_amdgcn_wave_barrier();
    }
    T* op = o + (bb * D.Lq + i) * D.ldo + h * D.hd;
    for (int d = lane; d < D.hd; d += 64) {
        float acc = 0.f;
        const T* vp = v + bb * D.Lk * D.ldv + h * D.hd + d;
        for (int64_t j = 0; j < D.Lk; ++j) acc = fmaf(ps[j], to_f32(vp[j * D.ldv]), acc);
        op[d] = from_f32<T>(acc * inv);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                         float* __restrict__ delta, AttnDims D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= D.b * D.heads * D.Lq) return;
    const int64_t i = row % D.Lq, h = (row / D.Lq) % D.heads, bb = row / (D.Lq * D.heads);
    const int64_t off = (bb * D.Lq + i) * D.ldo + h * D.hd;
    float s = 0.f;
    for (int d = lane; d < D.hd; d += 64) s += to_f32(o[off + d]) * to_f32(dout[off + d]);
    s = wave_sum(s);
    if (lane == 0) delta[row] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_generic_dq_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                              const T* __restrict__ v, const T* __restrict__ dout,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta,
                                                              const uint8_t* __restrict__ kmask, T* __restrict__ dq,
                                                              AttnDims D, float scale) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= D.b * D.heads * D.Lq) return;
    const int64_t i = row % D.Lq, h = (row / D.Lq) % D.heads, bb = row / (D.Lq * D.heads);
    float* qs = sm + (size_t)wave * (2 * D.hd + D.Lk);
    float* dos = qs + D.hd;
    float* ds = dos + D.hd;
    const T* qp = q + (bb * D.Lq + i) * D.ldq + h * D.hd;
    const T* dop = dout + (bb * D.Lq + i) * D.ldo + h * D.hd;
    for (int d = lane; d < D.hd; d += 64) {
        qs[d] = to_f32(qp[d]);
        dos[d] = to_f32(dop[d]);
    }
    __builtin_amdgcn_wave_barrier();
    const float L = lse[row], dl = delta[row];
    for (int64_t j = lane; j < D.Lk; j += 64) {
        float dsv = 0.f;
        if ((!kmask || !kmask[bb * D.Lk + j]) && !(D.causal && j > i)) {
            const T* kp = k + (bb * D.Lk + j) * D.ldk + h * D.hd;
            const T* vp = v + (bb * D.Lk + j) * D.ldv + h * D.hd;
            float s = 0.f, dp = 0.f;
            for (int d = 0; d < D.hd; ++d) {
                s = fmaf(qs[d], to_f32(kp[d]), s);
                dp = fmaf(dos[d], to_f32(vp[d]), dp);
            }
            const float p = __expf(s * scale - L);
            const float dm = D.dropout ? drop_mul(D.drop, (uint32_t)(row * D.Lk + j)) : 1.f;
            dsv = p * (dm * dp - dl);
        }
        ds[j] = dsv;
    }
    __builtin_amdgcn_wave_barrier();
    T* dqp = dq + (bb * D.Lq + i) * D.ldq + h * D.hd;
    for (int d = lane; d < D.hd; d += 64) {
        float acc = 0.f;
        const T* kp = k + bb * D.Lk * D.ldk + h * D.hd + d;
        for (int64_t j = 0; j < D.Lk; ++j) acc = fmaf(ds[j], to_f32(kp[j * D.ldk]), acc);
        dqp[d] = from_f32<T>(acc * scale);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_generic_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ v, const T* __restrict__ dout,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               const uint8_t* __restrict__ kmask, T* __restrict__ dk,
                                                               T* __restrict__ dv, AttnDims D, float scale) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;  // over b*heads*Lk
    if (row >= D.b * D.heads * D.Lk) return;
    const int64_t j = row % D.Lk, h = (row / D.Lk) % D.heads, bb = row / (D.Lk * D.heads);
    float* ks = sm + (size_t)wave * (2 * D.hd + 2 * D.Lq);
    float* vs = ks + D.hd;
    float* ps = vs + D.hd;
    float* ds = ps + D.Lq;
    const T* kp = k + (bb * D.Lk + j) * D.ldk + h * D.hd;
    const T* vp = v + (bb * D.Lk + j) * D.ldv + h * D.hd;
    for (int d = lane; d < D.hd; d += 64) {
        ks[d] = to_f32(kp[d]);
        vs[d] = to_f32(vp[d]);
    }
    __builtin_amdgcn_wave_barrier();
    const bool masked = kmask && kmask[bb * D.Lk + j];
    for (int64_t i = lane; i < D.Lq; i += 64) {
        float p = 0.f, dsv = 0.f;
        if (!masked && !(D.causal && j > i)) {
            const T* qp = q + (bb * D.Lq + i) * D.ldq + h * D.hd;
            const T* dop = dout + (bb * D.Lq + i) * D.ldo + h * D.hd;
            float s = 0.f, dp = 0.f;
            for (int d = 0; d < D.hd; ++d) {
                s = fmaf(ks[d], to_f32(qp[d]), s);
                dp = fmaf(vs[d], to_f32(dop[d]), dp);
            }
            const int64_t r = (bb * D.heads + h) * D.Lq + i;
            p = __expf(s * scale - lse[r]);
            const float dm = D.dropout ? drop_mul(D.drop, (uint32_t)(r * D.Lk + j)) : 1.f;
            dsv = p * (dm * dp - delta[r]);
            p *= dm;                                  // dV uses the dropped probabilities
        }
        ps[i] = p;
        ds[i] = dsv;
    }
    __builtin_amdgcn_wave_barrier();
    T* dkp = dk + (bb * D.Lk + j) * D.ldk + h * D.hd;
    T* dvp = dv + (bb * D.Lk + j) * D.ldv + h * D.hd;
    for (int d = lane; d < D.hd; d += 64) {
        float ak = 0.f, av = 0.f;
        const T* qp = q + bb * D.Lq * D.ldq + h * D.hd + d;
        const T* dop = dout + bb * D.Lq * D.ldo + h * D.hd + d;
        for (int64_t i = 0; i < D.Lq; ++i) {
            ak = fmaf(ds[i], to_f32(qp[i * D.ldq]), ak);
            av = fmaf(ps[i], to_f32(dop[i * D.ldo]), av);
        }
        dkp[d] = from_f32<T>(ak * scale);
        dvp[d] = from_f32<T>(av);
    }
}

// ------------------------------------------------------------------------------------------------ few rows, wide heads
// The reference's shipped video geometry (configs/cclm-base-ft/Retrieval_msrvtt.yaml:36-37: 4096-wide frame features, nn.MultiheadAttention(4096, 8)
// = 8 heads of 512; models/attention.py:90,107, video_model_retrieval_caption.py:26-34): 8-20 frames, 2-4 caption queries, 64 caption
// tokens.  A problem of a few rows and a very wide head: the kernels above put the KEYS (queries) on the lanes - 8 of 64 lanes busy, each
// walking a 512-long dot product over its own strided row.  Here the HEAD DIMENSION is on the lanes (EPL = hd / 64 contiguous elements per
// lane, vector loads of whole rows), a wave owns one softmax row (forward, dQ) or one key (dK / dV), walks the <= 64 rows of the other
// side and reduces each dot product across the wave; dQ is ONE pass over the keys (s and dP reduced together), dK / dV one pass over the
// queries.  Same arithmetic, masks, dropout counter and outputs as the generic kernels.
template <typename T, int EPL>
__device__ __forceinline__ void load_slice(const T* __restrict__ p, float (&f)[EPL]) {
    if constexpr (EPL >= 4) {
#pragma unroll
        for (int c = 0; c < EPL / 4; ++c) {
            const f32x4 v = load4(p + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * c + e] = v[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) f[e] = to_f32(p[e]);
    }
}
template <typename T, int EPL>
__device__ __forceinline__ void store_slice(T* __restrict__ p, const float (&f)[EPL]) {
    if constexpr (EPL >= 4) {
#pragma unroll
        for (int c = 0; c < EPL / 4; ++c) store4(p + 4 * c, f32x4{f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]});
    } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e) p[e] = from_f32<T>(f[e]);
    }
}
template <int EPL>
__device__ __forceinline__ float dot_slice(const float (&a)[EPL], const float (&b)[EPL]) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s = fmaf(a[e], b[e], s);
    return s;
}

template <typename T, int EPL>
__global__ __launch_bounds__(256) void attn_wide_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                            T* __restrict__ o, float* __restrict__ lse, const uint8_t* __restrict__ kmask,
                                                            AttnDims D, float scale) {
    __shared__ float ps_all[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;          // over b * heads * Lq
    if (row >= D.b * D.heads * D.Lq) return;
    const int64_t i = row % D.Lq, h = (row / D.Lq) % D.heads, bb = row / (D.Lq * D.heads);
    float* ps = ps_all[wave];
    float qf[EPL], kf[EPL];
    load_slice<T, EPL>(q + (bb * D.Lq + i) * D.ldq + h * D.hd + lane * EPL, qf);
    float mine = -INFINITY;                                      // lane j keeps score j
    for (int64_t j = 0; j < D.Lk; ++j) {
        float s = -INFINITY;
        if (!kmask || !kmask[bb * D.Lk + j]) {                   // (wave-uniform)
            load_slice<T, EPL>(k + (bb * D.Lk + j) * D.ldk + h * D.hd + lane * EPL, kf);
            s = wave_sum(dot_slice<EPL>(qf, kf)) * scale;
        }
        if (lane == j) mine = s;
    }
    const float mx = wave_max(mine);
    float p = (lane < D.Lk && mx != -INFINITY) ? __expf(mine - mx) : 0.f;
    const float l = wave_sum(p);
    if (lane == 0) lse[row] = l > 0.f ? mx + __logf(l) : INFINITY;
    p = l > 0.f ? p / l : 0.f;
    if (D.dropout && lane < D.Lk) p *= drop_mul(D.drop, (uint32_t)(row * D.Lk + lane));
    ps[lane] = p;
    __builtin_amdgcn_wave_barrier();
    float of[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) of[e] = 0.f;
    for (int64_t j = 0; j < D.Lk; ++j) {
        const float pj = ps[j];
        if (pj != 0.f) {                                         // (wave-uniform: a masked or dropped key contributes nothing)
            load_slice<T, EPL>(v + (bb * D.Lk + j) * D.ldv + h * D.hd + lane * EPL, kf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) of[e] = fmaf(pj, kf[e], of[e]);
        }
    }
    store_slice<T, EPL>(o + (bb * D.Lq + i) * D.ldo + h * D.hd + lane * EPL, of);
}

// dq_i = scale sum_j dS_ij k_j, dS_ij = p_ij (drop_ij dO_i.v_j - delta_i); delta_i = dO_i.O_i is computed here and kept for the dK / dV kernel
template <typename T, int EPL>
__global__ __launch_bounds__(256) void attn_wide_dq_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                           const T* __restrict__ o, const T* __restrict__ dout, const float* __restrict__ lse,
                                                           const uint8_t* __restrict__ kmask, T* __restrict__ dq, float* __restrict__ delta,
                                                           AttnDims D, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= D.b * D.heads * D.Lq) return;
    const int64_t i = row % D.Lq, h = (row / D.Lq) % D.heads, bb = row / (D.Lq * D.heads);
    float qf[EPL], gf[EPL], kf[EPL], vf[EPL], acc[EPL];
    load_slice<T, EPL>(q + (bb * D.Lq + i) * D.ldq + h * D.hd + lane * EPL, qf);
    load_slice<T, EPL>(dout + (bb * D.Lq + i) * D.ldo + h * D.hd + lane * EPL, gf);
    load_slice<T, EPL>(o + (bb * D.Lq + i) * D.ldo + h * D.hd + lane * EPL, kf);
    const float dl = wave_sum(dot_slice<EPL>(gf, kf));
    if (lane == 0) delta[row] = dl;
    const float L = lse[row];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int64_t j = 0; j < D.Lk; ++j) {
        if (kmask && kmask[bb * D.Lk + j]) continue;
        load_slice<T, EPL>(k + (bb * D.Lk + j) * D.ldk + h * D.hd + lane * EPL, kf);
        load_slice<T, EPL>(v + (bb * D.Lk + j) * D.ldv + h * D.hd + lane * EPL, vf);
        const float s = wave_sum(dot_slice<EPL>(qf, kf)), dp = wave_sum(dot_slice<EPL>(gf, vf));
        const float p = __expf(s * scale - L);
        const float dm = D.dropout ? drop_mul(D.drop, (uint32_t)(row * D.Lk + j)) : 1.f;
        const float ds = p * (dm * dp - dl);
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = fmaf(ds, kf[e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] *= scale;
    store_slice<T, EPL>(dq + (bb * D.Lq + i) * D.ldq + h * D.hd + lane * EPL, acc);
}

template <typename T, int EPL>
__global__ __launch_bounds__(256) void attn_wide_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                            const T* __restrict__ dout, const float* __restrict__ lse,
                                                            const float* __restrict__ delta, const uint8_t* __restrict__ kmask,
                                                            T* __restrict__ dk, T* __restrict__ dv, AttnDims D, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);          // over b * heads * Lk
    if (row >= D.b * D.heads * D.Lk) return;
    const int64_t j = row % D.Lk, h = (row / D.Lk) % D.heads, bb = row / (D.Lk * D.heads);
    float kf[EPL], vf[EPL], qf[EPL], gf[EPL], ak[EPL], av[EPL];
    load_slice<T, EPL>(k + (bb * D.Lk + j) * D.ldk + h * D.hd + lane * EPL, kf);
    load_slice<T, EPL>(v + (bb * D.Lk + j) * D.ldv + h * D.hd + lane * EPL, vf);
#pragma unroll
    for (int e = 0; e < EPL; ++e) ak[e] = av[e] = 0.f;
    if (!(kmask && kmask[bb * D.Lk + j])) {
        for (int64_t i = 0; i < D.Lq; ++i) {
            load_slice<T, EPL>(q + (bb * D.Lq + i) * D.ldq + h * D.hd + lane * EPL, qf);
            load_slice<T, EPL>(dout + (bb * D.Lq + i) * D.ldo + h * D.hd + lane * EPL, gf);
            const float s = wave_sum(dot_slice<EPL>(kf, qf)), dp = wave_sum(dot_slice<EPL>(vf, gf));
            const int64_t r = (bb * D.heads + h) * D.Lq + i;
            const float p = __expf(s * scale - lse[r]);
            const float dm = D.dropout ? drop_mul(D.drop, (uint32_t)(r * D.Lk + j)) : 1.f;
            const float ds = p * (dm * dp - delta[r]), pd = p * dm;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                ak[e] = fmaf(ds, qf[e], ak[e]);
                av[e] = fmaf(pd, gf[e], av[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) ak[e] *= scale;
    store_slice<T, EPL>(dk + (bb * D.Lk + j) * D.ldk + h * D.hd + lane * EPL, ak);
    store_slice<T, EPL>(dv + (bb * D.Lk + j) * D.ldv + h * D.hd + lane * EPL, av);
}

// hd = 64 EPL with EPL in {2, 4, 8, 16}, at most 64 keys and 64 queries, rows aligned for the vector loads, no causal mask
static int wide_epl(const AttnDims& D, const void* const* ptrs, int nptrs, size_t esz) {
    static const bool on = [] { const char* e = getenv("LECCR_ATTN_WIDE"); return !(e && e[0] == '0'); }();
    if (!on || D.causal || D.Lq > 64 || D.Lk > 64 || D.hd % 64) return 0;
    const int epl = (int)(D.hd / 64);
    if (epl != 2 && epl != 4 && epl != 8 && epl != 16) return 0;
    if ((D.ldq | D.ldk | D.ldv | D.ldo) & 3) return 0;
    for (int i = 0; i < nptrs; ++i)
        if (reinterpret_cast<uintptr_t>(ptrs[i]) & (4 * esz - 1)) return 0;
    return epl;
}

#define WIDE_SWITCH(EPL_, CALL)                         \
    switch (EPL_) {                                     \
        case 2: { constexpr int E = 2; CALL; } break;   \
        case 4: { constexpr int E = 4; CALL; } break;   \
        case 8: { constexpr int E = 8; CALL; } break;   \
        default: { constexpr int E = 16; CALL; } break; \
    }

template <typename T>
int attn_generic_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                     const AttnDims& D, hipStream_t st) {
    {
        const void* ptrs[4] = {q, k, v, o};
        if (const int epl = wide_epl(D, ptrs, 4, sizeof(T))) {
            const float sc = 1.f / sqrtf((float)D.hd);
            const dim3 grid((unsigned)cdiv(D.b * D.heads * D.Lq, 4));
            WIDE_SWITCH(epl, hipLaunchKernelGGL((attn_wide_fwd_kernel<T, E>), grid, dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)v,
                                                (T*)o, lse, kmask, D, sc));
            return check_launch("attn_wide_fwd");
        }
    }
    const size_t shm = 4 * (size_t)(D.hd + D.Lk) * sizeof(float);
    if (shm > 64 * 1024) {
        set_error("attn_fwd(generic): Lk=%lld too long for the generic kernel", (long long)D.Lk);
        return LECCR_EINVAL;
    }
    const float scale = 1.f / sqrtf((float)D.hd);
    hipLaunchKernelGGL((attn_generic_fwd_kernel<T>), dim3((unsigned)cdiv(D.b * D.heads * D.Lq, 4)), dim3(256), shm, st,
                       (const T*)q, (const T*)k, (const T*)v, (T*)o, lse, kmask, D, scale);
    return check_launch("attn_generic_fwd");
}

template <typename T>
int attn_generic_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                     const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, const AttnDims& D,
                     hipStream_t st) {
    {
        const void* ptrs[8] = {q, k, v, o, dout, dq, dk, dv};
        if (const int epl = wide_epl(D, ptrs, 8, sizeof(T))) {
            const float sc = 1.f / sqrtf((float)D.hd);
            const dim3 gq((unsigned)cdiv(D.b * D.heads * D.Lq, 4)), gk((unsigned)cdiv(D.b * D.heads * D.Lk, 4));
            WIDE_SWITCH(epl, hipLaunchKernelGGL((attn_wide_dq_kernel<T, E>), gq, dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)v,
                                                (const T*)o, (const T*)dout, lse, kmask, (T*)dq, delta, D, sc));
            WIDE_SWITCH(epl, hipLaunchKernelGGL((attn_wide_dkv_kernel<T, E>), gk, dim3(256), 0, st, (const T*)q, (const T*)k, (const T*)v,
                                                (const T*)dout, lse, (const float*)delta, kmask, (T*)dk, (T*)dv, D, sc));
            return check_launch("attn_wide_bwd");
        }
    }
    const size_t shm1 = 4 * (size_t)(2 * D.hd + D.Lk) * sizeof(float);
    const size_t shm2 = 4 * (size_t)(2 * D.hd + 2 * D.Lq) * sizeof(float);
    if (shm1 > 64 * 1024 || shm2 > 64 * 1024) {
        set_error("attn_bwd(generic): sequence too long for the generic kernel");
        return LECCR_EINVAL;
    }
    const float scale = 1.f / sqrtf((float)D.hd);
    hipLaunchKernelGGL((attn_delta_kernel<T>), dim3((unsigned)cdiv(D.b * D.heads * D.Lq, 4)), dim3(256), 0, st,
                       (const T*)o, (const T*)dout, delta, D);
    hipLaunchKernelGGL((attn_generic_dq_kernel<T>), dim3((unsigned)cdiv(D.b * D.heads * D.Lq, 4)), dim3(256), shm1, st,
                       (const T*)q, (const T*)k, (const T*)v, (const T*)dout, lse, delta, kmask, (T*)dq, D, scale);
    hipLaunchKernelGGL((attn_generic_dkv_kernel<T>), dim3((unsigned)cdiv(D.b * D.heads * D.Lk, 4)), dim3(256), shm2, st,
                       (const T*)q, (const T*)k, (const T*)v, (const T*)dout, lse, delta, kmask, (T*)dk, (T*)dv, D,
                       scale);
    return check_launch("attn_generic_bwd");
}

template int attn_generic_fwd<float>(const void*, const void*, const void*, void*, float*, const uint8_t*, const AttnDims&, hipStream_t);
template int attn_generic_fwd<bf16_t>(const void*, const void*, const void*, void*, float*, const uint8_t*, const AttnDims&, hipStream_t);
template int attn_generic_bwd<float>(const void*, const void*, const void*, const void*, const void*, const float*, const uint8_t*, void*, void*, void*, float*, const AttnDims&, hipStream_t);
template int attn_generic_bwd<bf16_t>(const void*, const void*, const void*, const void*, const void*, const float*, const uint8_t*, void*, void*, void*, float*, const AttnDims&, hipStream_t);

}  // namespace leccr
