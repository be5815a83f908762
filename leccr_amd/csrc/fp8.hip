// fp8.hip - per-tensor scaled OCP fp8 copies of GEMM operands (BASELINE configs[4]: fp8 MFMA GEMM inputs), delayed scaling.
// See include/leccr_hip.h ("fp8 GEMM operands").  HBM-bound passes: read 2 bytes, write 1 per element.
#include "common.h"

namespace leccr {
namespace {

__device__ __forceinline__ void atomic_max_pos(float* addr, float v) {      // v >= 0: float order == unsigned order of the bits
    atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// One atomic per BLOCK, and only when it would raise the running maximum: 65 536 waves adding to one address serialise at the
// memory side (~10 ns each: 0.5 ms of a 0.25 ms pass, measured through rocprofv3 on the config-E step).
__device__ __forceinline__ void block_amax(float m, float* __restrict__ site) {
    __shared__ float wm[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > __builtin_nontemporal_load(site)) atomic_max_pos(site, m);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fp8_amax_kernel(const T* __restrict__ x, int64_t rows, int cols, int64_t ldx,
                                                       float* __restrict__ site) {
    float m = 0.f;
    const int c8 = cols >> 3;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const T* row = x + r * ldx;
        for (int c = threadIdx.x; c < c8; c += 256) {
            if constexpr (sizeof(T) == 2) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + c * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf((float)v[i]));
            } else {
                const f32x4 v0 = load4(row + c * 8), v1 = load4(row + c * 8 + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) m = fmaxf(m, fmaxf(fabsf(v0[i]), fabsf(v1[i])));
            }
        }
    }
    block_amax(m, site);
}

__global__ void fp8_update_kernel(float* __restrict__ sites, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* s = sites + 4 * i;
    const float amax = s[0];
    if (amax > 0.f && isfinite(amax)) {
        const float fmax = s[3] != 0.f ? 57344.f : 448.f;
        // a power of two: scaling is then exact in bf16 / f32, and the headroom (< 2x) absorbs next step's growth
        // exponent clamped so that sc and 1 / sc stay finite and normal for a (nearly) all-zero tensor (amax ~ 1e-38: fmax / amax = inf)
        const float sc = exp2f(fminf(floorf(log2f(fmax) - log2f(amax)), 100.f));
        s[1] = sc;
        s[2] = 1.f / sc;
    }
    s[0] = 0.f;
}

// 8 elements per thread: 16-byte (bf16) loads, 8-byte stores
template <typename T, int FMT>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const T* __restrict__ x, uint8_t* __restrict__ y, int64_t rows, int cols,
                                                           int64_t ldx, int64_t ldy, float* __restrict__ site) {
    const float sc = site[1];
    constexpr float FMAX = FMT ? 57344.f : 448.f;
    float m = 0.f;
    const int c8 = cols >> 3;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const T* row = x + r * ldx;
        uint8_t* out = y + r * ldy;
        for (int c = threadIdx.x; c < c8; c += 256) {
            float v[8];
            if constexpr (sizeof(T) == 2) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(row + c * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
            } else {
                const f32x4 v0 = load4(row + c * 8), v1 = load4(row + c * 8 + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = v0[i]; v[4 + i] = v1[i]; }
            }
            int w[2] = {0, 0};
            float q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                m = fmaxf(m, fabsf(v[i]));
                q[i] = fminf(fmaxf(v[i] * sc, -FMAX), FMAX);
            }
            if constexpr (FMT == 0) {
                w[0] = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w[0], false);
                w[0] = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w[0], true);
                w[1] = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w[1], false);
                w[1] = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w[1], true);
            } else {
                w[0] = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w[0], false);
                w[0] = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w[0], true);
                w[1] = __builtin_amdgcn_cvt_pk_bf8_f32(q[4], q[5], w[1], false);
                w[1] = __builtin_amdgcn_cvt_pk_bf8_f32(q[6], q[7], w[1], true);
            }
            *reinterpret_cast<int2*>(out + c * 8) = int2{w[0], w[1]};
        }
    }
    block_amax(m, site);
}

}  // namespace
}  // namespace leccr

using namespace leccr;

static unsigned rows_grid(int64_t rows) { return (unsigned)(rows < 4096 ? (rows < 1 ? 1 : rows) : 4096); }
// contiguous rows (ld == cols): the tensor is re-cut into pseudo-rows of 2048 elements so that every thread of a block has one
// 8-element chunk (a 768- or 1024-wide row leaves half of a 256-thread block idle: 3.3 TB/s measured against 4.8 on 4096-wide rows)
static void recut(int64_t& rows, int64_t& cols, int64_t& ldx, int64_t* ldy) {
    if (ldx != cols || (ldy && *ldy != cols)) return;
    const int64_t total = rows * cols;
    if (total % 2048 == 0 && cols < 2048) {
        rows = total / 2048; cols = 2048; ldx = 2048;
        if (ldy) *ldy = 2048;
    }
}

extern "C" int leccr_fp8_amax(const void* x, int64_t rows, int64_t cols, int64_t ldx, int dtype, float* site, void* stream) {
    LECCR_REQUIRE(x && site && rows >= 0 && cols > 0 && (cols & 7) == 0 && ldx >= cols && (ldx & 7) == 0, LECCR_EINVAL, "fp8_amax: bad argument (cols, ldx multiples of 8)");
    LECCR_REQUIRE(((uintptr_t)x & 15) == 0, LECCR_EALIGN, "fp8_amax: x must be 16-byte aligned");
    if (rows == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    recut(rows, cols, ldx, nullptr);
    if (dtype == LECCR_BF16) hipLaunchKernelGGL((fp8_amax_kernel<bf16_t>), dim3(rows_grid(rows)), dim3(256), 0, st, (const bf16_t*)x, rows, (int)cols, ldx, site);
    else if (dtype == LECCR_F32) hipLaunchKernelGGL((fp8_amax_kernel<float>), dim3(rows_grid(rows)), dim3(256), 0, st, (const float*)x, rows, (int)cols, ldx, site);
    else { set_error("fp8_amax: bad dtype"); return LECCR_EINVAL; }
    return check_launch("fp8_amax");
}

extern "C" int leccr_fp8_update(float* sites, int64_t n, void* stream) {
    LECCR_REQUIRE(sites && n >= 0, LECCR_EINVAL, "fp8_update: bad argument");
    if (n == 0) return LECCR_OK;
    hipLaunchKernelGGL(fp8_update_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), sites, n);
    return check_launch("fp8_update");
}

extern "C" int leccr_fp8_quantize(const void* x, void* y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, int dtype,
                                  int fmt, float* site, void* stream) {
    LECCR_REQUIRE(x && y && site && rows >= 0 && cols > 0 && (cols & 7) == 0 && ldx >= cols && ldy >= cols && (ldx & 7) == 0 && (ldy & 7) == 0,
                  LECCR_EINVAL, "fp8_quantize: bad argument (cols, ldx, ldy multiples of 8)");
    LECCR_REQUIRE(fmt == 0 || fmt == 1, LECCR_EINVAL, "fp8_quantize: fmt 0 (e4m3) or 1 (e5m2)");
    LECCR_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 7) == 0, LECCR_EALIGN, "fp8_quantize: misaligned");
    if (rows == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    recut(rows, cols, ldx, &ldy);
    const dim3 g(rows_grid(rows)), b(256);
    if (dtype == LECCR_BF16 && fmt == 0) hipLaunchKernelGGL((fp8_quantize_kernel<bf16_t, 0>), g, b, 0, st, (const bf16_t*)x, (uint8_t*)y, rows, (int)cols, ldx, ldy, site);
    else if (dtype == LECCR_BF16) hipLaunchKernelGGL((fp8_quantize_kernel<bf16_t, 1>), g, b, 0, st, (const bf16_t*)x, (uint8_t*)y, rows, (int)cols, ldx, ldy, site);
    else if (dtype == LECCR_F32 && fmt == 0) hipLaunchKernelGGL((fp8_quantize_kernel<float, 0>), g, b, 0, st, (const float*)x, (uint8_t*)y, rows, (int)cols, ldx, ldy, site);
    else if (dtype == LECCR_F32) hipLaunchKernelGGL((fp8_quantize_kernel<float, 1>), g, b, 0, st, (const float*)x, (uint8_t*)y, rows, (int)cols, ldx, ldy, site);
    else { set_error("fp8_quantize: bad dtype"); return LECCR_EINVAL; }
    return check_launch("fp8_quantize");
}
