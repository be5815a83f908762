// common.h - shared device helpers for the gfx950 kernels (wave64, bf16 <-> f32, reductions, error plumbing).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/leccr_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

namespace leccr {

constexpr int kWave = 64;

// host-side error text (thread local)
void set_error(const char* fmt, ...);
int check_launch(const char* what);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Optional in-library timing (leccr_prof_*): brackets a launch with hipEvents on the launch stream.
// kinds 0-4: work = algorithmic FLOPs (MFMA-bound launches); kinds 5-10: work = algorithmic HBM bytes (every operand read once,
// every result written once) of the HBM-bound launches - bench.py's `hbm_kernels` list divides them by the event time.
enum ProfKind { PROF_GEMM_NT = 0, PROF_GEMM_TN = 1, PROF_GEMM_GENERIC = 2, PROF_ATTN_FWD = 3, PROF_ATTN_BWD = 4,
                PROF_LN_FWD = 5, PROF_LN_BWD = 6, PROF_COLSUM = 7, PROF_EMBED = 8, PROF_ADAMW = 9, PROF_DROPOUT = 10,
                PROF_GEMM_FP8_NT = 11, PROF_GEMM_FP8_TN = 12,       // fp8-operand launches (forward / dX; weight gradients): priced against the fp8 peak
                PROF_SPLIT = 13,                                     // leccr_split3 passes (fp32 operands as bf16 hi / lo pairs): bytes
                PROF_KINDS = 14, PROF_KINDS_V1 = 5 };
struct ProfScope {
    int slot;
    hipStream_t st;
    ProfScope(int kind, double work, hipStream_t stream);
    ~ProfScope();
};

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int code = LECCR_F32; };
template <> struct DT<bf16_t> { static constexpr int code = LECCR_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return static_cast<bf16_t>(v); }

// 4-wide vector load/store of T as f32 (pointer must be aligned to 4*sizeof(T))
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
    bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `sh` needs >= 16 floats; every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}
__device__ __forceinline__ float block_min(float v, float* sh) {
    v = wave_min(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < nw; ++i) r = fminf(r, sh[i]);
    return r;
}

// activations (clip/model.py:166-168 QuickGELU; erf GELU of attention.py:16-17 / HF BERT) and their derivatives.
// Epilogue-hot (128 evaluations per lane per 256x256 output tile): hardware exp2/rcp instead of libm expf / IEEE
// division, and erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far inside the f32 parity tolerance) whose
// exp(-x^2/2) factor is shared with the Gaussian pdf the GELU derivative needs.
__device__ __forceinline__ float fast_sigmoid(float z) {      // 1 / (1 + e^-z)
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
__device__ __forceinline__ float quick_gelu(float x) { return x * fast_sigmoid(1.702f * x); }
__device__ __forceinline__ float dquick_gelu(float x) {
    const float s = fast_sigmoid(1.702f * x);
    return s * (1.f + 1.702f * x * (1.f - s));
}
// returns erf(x / sqrt 2) and e = exp(-x^2 / 2)
__device__ __forceinline__ float erf_half(float x, float& e) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.f - poly * e;
    return x < 0.f ? -r : r;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float e;
    return 0.5f * x * (1.f + erf_half(x, e));
}
__device__ __forceinline__ float dgelu_erf(float x) {
    float e;
    const float cdf = 0.5f * (1.f + erf_half(x, e));
    return cdf + x * 0.3989422804014327f * e;
}

template <int EPI> __device__ __forceinline__ float apply_epi(float v, float aux) {
    if constexpr (EPI == LECCR_EPI_QUICKGELU || EPI == LECCR_EPI_QUICKGELU_D) return quick_gelu(v);
    else if constexpr (EPI == LECCR_EPI_GELU || EPI == LECCR_EPI_GELU_D) return gelu_erf(v);
    else if constexpr (EPI == LECCR_EPI_DQUICKGELU) return v * dquick_gelu(aux);
    else if constexpr (EPI == LECCR_EPI_DGELU) return v * dgelu_erf(aux);
    else if constexpr (EPI == LECCR_EPI_MULAUX) return v * aux;
    else return v;
}
// epilogues that read aux / that save the derivative in aux_out
template <int EPI> constexpr bool epi_needs_aux = EPI == LECCR_EPI_DQUICKGELU || EPI == LECCR_EPI_DGELU || EPI == LECCR_EPI_MULAUX;
template <int EPI> constexpr bool epi_saves_deriv = EPI == LECCR_EPI_QUICKGELU_D || EPI == LECCR_EPI_GELU_D;
// the _D forms: activation and its derivative from one sigmoid / one erf (the value saved for LECCR_EPI_MULAUX)
struct ActD { float a, d; };
template <int EPI> __device__ __forceinline__ ActD apply_epi_d(float v) {
    if constexpr (EPI == LECCR_EPI_QUICKGELU_D) {
        const float s = fast_sigmoid(1.702f * v);
        return ActD{v * s, s * (1.f + 1.702f * v * (1.f - s))};
    } else {
        float e;
        const float cdf = 0.5f * (1.f + erf_half(v, e));
        return ActD{v * cdf, cdf + v * 0.3989422804014327f * e};
    }
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Counter-based dropout RNG: keep(element) depends only on (seed, element index), so the backward pass and the
// recomputing attention kernels regenerate the forward mask without storing it.  Two rounds of a 32-bit integer
// finaliser (lowbias32) keyed by the two seed halves.
struct DropCfg {
    uint32_t s0, s1, thr;   // keep iff hash >= thr, thr = p * 2^32
    float scale;            // 1 / (1 - p)
};
__host__ __device__ inline uint32_t drop_hash(uint32_t s0, uint32_t s1, uint32_t idx) {
    uint32_t x = idx ^ s0;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    x += s1 * 0x9e3779b9u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float drop_mul(const DropCfg& d, uint32_t idx) {   // scale or 0
    return drop_hash(d.s0, d.s1, idx) >= d.thr ? d.scale : 0.f;
}
inline DropCfg make_drop(float p, uint64_t seed) {
    DropCfg d;
    d.s0 = (uint32_t)seed;
    d.s1 = (uint32_t)(seed >> 32);
    const double t = (double)p * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    d.scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
    return d;
}

}  // namespace leccr

#define LECCR_REQUIRE(cond, code, ...)          \
    do {                                        \
        if (!(cond)) {                          \
            leccr::set_error(__VA_ARGS__);      \
            return (code);                      \
        }                                       \
    } while (0)
