// optim.hip - flat-arena optimizer and dtype plumbing (HBM-bound streaming kernels, 16-byte accesses).
//   adamw           transformers.AdamW semantics (optim.py:1,58-63): Adam update with bias correction, THEN decoupled
//                   decay p -= lr*wd*p on the updated value; eps added to sqrt(v) before bias correction.
//   cast_bf16       f32 arena -> bf16 compute copy
//   transpose_cast  [R,C] f32 -> [C,R] T  (K-contiguous copies of weights for the dX GEMMs)
//   axpy            y += a*x
#include "common.h"

namespace leccr {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                    const int64_t* __restrict__ seg_end,
                                                    const float* __restrict__ seg_wd, int n_seg, float lr, float beta1,
                                                    float beta2, float eps, float step_size, float gscale) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        // segment lookup (binary search over the cumulative ends; segments are 4-aligned so one lookup per quad)
        int lo = 0, hi = n_seg - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (seg_end[mid] > i) hi = mid; else lo = mid + 1;
        }
        const float wd = seg_wd[lo];
        if (wd < 0.f) continue;  // segment never receives a gradient: the reference optimizer skips it (grad is None)
        const int cnt = n - i < 4 ? (int)(n - i) : 4;
        for (int j = 0; j < cnt; ++j) {
            const float gr = g[i + j] * gscale;
            const float mm = beta1 * m[i + j] + (1.f - beta1) * gr;
            const float vv = beta2 * v[i + j] + (1.f - beta2) * gr * gr;
            m[i + j] = mm;
            v[i + j] = vv;
            float pp = p[i + j] - step_size * mm / (sqrtf(vv) + eps);
            pp -= lr * wd * pp;
            p[i + j] = pp;
        }
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) store4(dst + i, load4(src + i));
        else for (int64_t j = i; j < n; ++j) dst[j] = from_f32<bf16_t>(src[j]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < C) tile[k][tx] = src[(int64_t)(r0 + k) * C + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && r0 + tx < R) dst[(int64_t)(c0 + k) * R + r0 + tx] = from_f32<T>(tile[tx][k]);
}

template <typename T>
__global__ __launch_bounds__(256) void axpy_kernel(const T* __restrict__ x, T* __restrict__ y, float a, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = from_f32<T>(to_f32(y[i]) + a * to_f32(x[i]));
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, DropCfg d) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = from_f32<T>(to_f32(x[i]) * drop_mul(d, (uint32_t)i));
}

static unsigned stream_grid(int64_t n_items) {
    int64_t nb = cdiv(n_items, 256);
    return (unsigned)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb));
}

}  // namespace leccr

using namespace leccr;

extern "C" int leccr_adamw(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                           const float* seg_wd, int64_t n_seg, float lr, float beta1, float beta2, float eps, int64_t step,
                           float gscale, void* stream) {
    LECCR_REQUIRE(p && g && m && v && seg_end && seg_wd && n_seg > 0 && step >= 1, LECCR_EINVAL, "adamw: bad argument");
    if (n == 0) return LECCR_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
    hipLaunchKernelGGL(adamw_kernel, dim3(stream_grid(cdiv(n, 4))), dim3(256), 0, as_stream(stream), p, g, m, v, n, seg_end,
                       seg_wd, (int)n_seg, lr, beta1, beta2, eps, step_size, gscale);
    return check_launch("adamw");
}

extern "C" int leccr_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
    LECCR_REQUIRE(src && dst, LECCR_EINVAL, "cast_bf16: null pointer");
    LECCR_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, LECCR_EALIGN, "cast_bf16: misaligned");
    if (n == 0) return LECCR_OK;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(stream_grid(cdiv(n, 4))), dim3(256), 0, as_stream(stream), src, (bf16_t*)dst, n);
    return check_launch("cast_bf16");
}

extern "C" int leccr_transpose_cast(const float* src, void* dst, int64_t R, int64_t C, int dtype, void* stream) {
    LECCR_REQUIRE(src && dst && R > 0 && C > 0, LECCR_EINVAL, "transpose_cast: bad argument");
    dim3 grid((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32));
    if (dtype == LECCR_F32) hipLaunchKernelGGL((transpose_cast_kernel<float>), grid, dim3(256), 0, as_stream(stream), src, (float*)dst, (int)R, (int)C);
    else if (dtype == LECCR_BF16) hipLaunchKernelGGL((transpose_cast_kernel<bf16_t>), grid, dim3(256), 0, as_stream(stream), src, (bf16_t*)dst, (int)R, (int)C);
    else { set_error("transpose_cast: bad dtype"); return LECCR_EINVAL; }
    return check_launch("transpose_cast");
}

extern "C" int leccr_axpy(const void* x, void* y, float a, int64_t n, int dtype, void* stream) {
    LECCR_REQUIRE(x && y, LECCR_EINVAL, "axpy: null pointer");
    if (n == 0) return LECCR_OK;
    if (dtype == LECCR_F32) hipLaunchKernelGGL((axpy_kernel<float>), dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, a, n);
    else if (dtype == LECCR_BF16) hipLaunchKernelGGL((axpy_kernel<bf16_t>), dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, a, n);
    else { set_error("axpy: bad dtype"); return LECCR_EINVAL; }
    return check_launch("axpy");
}

extern "C" int leccr_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
    LECCR_REQUIRE(x && y && p >= 0.f && p < 1.f, LECCR_EINVAL, "dropout: bad argument");
    LECCR_REQUIRE(n < (1ll << 32), LECCR_EINVAL, "dropout: more than 2^32 elements");
    if (n == 0) return LECCR_OK;
    const DropCfg d = make_drop(p, seed);
    if (dtype == LECCR_F32) hipLaunchKernelGGL((dropout_kernel<float>), dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, n, d);
    else if (dtype == LECCR_BF16) hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, n, d);
    else { set_error("dropout: bad dtype"); return LECCR_EINVAL; }
    return check_launch("dropout");
}
