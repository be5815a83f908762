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
                                                    float beta2, float eps, float step_size, float gscale,
                                                    bf16_t* __restrict__ lp) {
    // 16-byte accesses on all four streams (the arena and every segment are 4-element aligned); the bf16 compute copy
    // of the updated weights is written from the same registers (lp != NULL) instead of a second pass over p.
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        // segment lookup (binary search over the cumulative ends; segments are 4-aligned so one lookup per quad)
        int lo = 0, hi = n_seg - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (seg_end[mid] > i) hi = mid; else lo = mid + 1;
        }
        const float wd = seg_wd[lo];
        if (i + 4 <= n) {
            f32x4 pp = load4(p + i);
            if (wd >= 0.f) {         // wd < 0: the segment never receives a gradient, the reference optimizer skips it
                const f32x4 gr = load4(g + i) * gscale;
                f32x4 mm = load4(m + i), vv = load4(v + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mm[j] = beta1 * mm[j] + (1.f - beta1) * gr[j];
                    vv[j] = beta2 * vv[j] + (1.f - beta2) * gr[j] * gr[j];
                    pp[j] = pp[j] - step_size * mm[j] / (sqrtf(vv[j]) + eps);
                    pp[j] -= lr * wd * pp[j];
                }
                store4(m + i, mm);
                store4(v + i, vv);
                store4(p + i, pp);
            }
            if (lp) store4(lp + i, pp);
        } else {
            for (int64_t k = i; k < n; ++k) {
                float pp = p[k];
                if (wd >= 0.f) {
                    const float gr = g[k] * gscale;
                    const float mm = beta1 * m[k] + (1.f - beta1) * gr;
                    const float vv = beta2 * v[k] + (1.f - beta2) * gr * gr;
                    m[k] = mm;
                    v[k] = vv;
                    pp = pp - step_size * mm / (sqrtf(vv) + eps);
                    pp -= lr * wd * pp;
                    p[k] = pp;
                }
                if (lp) lp[k] = (bf16_t)pp;
            }
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
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, DropCfg d,
                                                      int64_t base) {      // element i of this call is element base + i of the tensor
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = from_f32<T>(to_f32(x[i]) * drop_mul(d, (uint32_t)(base + i)));
}

// bf16, 16 bytes per lane: 8 consecutive elements per thread (the mask of element i depends on i only: same masks as above)
__global__ __launch_bounds__(256) void dropout8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n8, DropCfg d) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n8; q += stride) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + q * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)v[j] * drop_mul(d, (uint32_t)(q * 8 + j)));
        *reinterpret_cast<bf16x8*>(y + q * 8) = o;
    }
}

static unsigned stream_grid(int64_t n_items) {
    int64_t nb = cdiv(n_items, 256);
    return (unsigned)(nb > 4096 ? 4096 : (nb < 1 ? 1 : nb));
}

}  // namespace leccr

using namespace leccr;

extern "C" int leccr_adamw(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                           const float* seg_wd, int64_t n_seg, float lr, float beta1, float beta2, float eps, int64_t step,
                           float gscale, void* lp_bf16, void* stream) {
    LECCR_REQUIRE(p && g && m && v && seg_end && seg_wd && n_seg > 0 && step >= 1, LECCR_EINVAL, "adamw: bad argument");
    LECCR_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)lp_bf16 & 7) == 0,
                  LECCR_EALIGN, "adamw: arenas must be 16-byte aligned");
    if (n == 0) return LECCR_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
    ProfScope prof(PROF_ADAMW, (double)n * (28.0 + (lp_bf16 ? 2.0 : 0.0)), as_stream(stream));   // p,g,m,v read; p,m,v (+bf16) written
    hipLaunchKernelGGL(adamw_kernel, dim3(stream_grid(cdiv(n, 4))), dim3(256), 0, as_stream(stream), p, g, m, v, n, seg_end,
                       seg_wd, (int)n_seg, lr, beta1, beta2, eps, step_size, gscale, (bf16_t*)lp_bf16);
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

// One launch for every [in,out] compute copy of the arena (209 matrices at ViT-B/16 + mBERT: 209 launches of ~5 us
// were 1.2 ms per step).  table[i] = {first tile, src pointer (f32 [R,C]), dst pointer (bf16 [C,R]), R, C}; a block = one 64 x 64 tile.
// Round 5: 16-byte accesses on both sides.  The first version moved 32 x 32 tiles by 4-byte reads and 2-byte stores (64 bytes of a
// destination row per wave instruction: ~1 TB/s, 2 x 483 us per step).  Now a tile is read as float4 rows (16 lanes = one 256-byte
// source row piece), and thread (c = tid >> 2, q = tid & 3) stores 16 consecutive r of destination row c as two bf16x8: four adjacent
// lanes write one whole 128-byte line.  LDS tile [64][65] floats: the column reads (lane stride 65 words down a column group, q * 16 * 65
// = q * 16 mod 64 across the quarter) touch 64 distinct banks.  Tiles on an edge, odd shapes and misaligned pointers take the scalar loop.
__global__ __launch_bounds__(256) void transpose_cast_batched_kernel(const int64_t* __restrict__ table, int n) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n - 1;                                  // last matrix whose first tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 5] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* e = table + lo * 5;
    const float* src = reinterpret_cast<const float*>(e[1]);
    bf16_t* dst = reinterpret_cast<bf16_t*>(e[2]);
    const int R = (int)e[3], C = (int)e[4];
    const int t = (int)(blockIdx.x - e[0]), tiles_x = (C + 63) >> 6;
    const int r0 = (t / tiles_x) * 64, c0 = (t % tiles_x) * 64;
    const int tid = threadIdx.x;
    const bool full = r0 + 64 <= R && c0 + 64 <= C && (C & 3) == 0 && (R & 7) == 0 &&
                      ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if (full) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 4) + 16 * i, c4 = (tid & 15) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (int64_t)(r0 + r) * C + c0 + c4);
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[r][c4 + k] = v[k];
        }
    } else {
        for (int k = tid; k < 64 * 64; k += 256) {
            const int r = k >> 6, c = k & 63;
            tile[r][c] = (r0 + r < R && c0 + c < C) ? src[(int64_t)(r0 + r) * C + c0 + c] : 0.f;
        }
    }
    __syncthreads();
    const int c = tid >> 2, q = tid & 3;                     // destination row c0 + c, source rows r0 + 16 q .. + 15
    if (full) {
        bf16_t* d = dst + (int64_t)(c0 + c) * R + r0 + 16 * q;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)tile[16 * q + 8 * k + j][c];
            *reinterpret_cast<bf16x8*>(d + 8 * k) = o;
        }
    } else if (c0 + c < C) {
        for (int j = 0; j < 16; ++j) {
            const int r = r0 + 16 * q + j;
            if (r >= R) break;
            dst[(int64_t)(c0 + c) * R + r] = (bf16_t)tile[16 * q + j][c];
        }
    }
}

extern "C" int leccr_transpose_cast_batched(const int64_t* table, int64_t n, int64_t total_tiles, void* stream) {
    LECCR_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), LECCR_EINVAL, "transpose_cast_batched: bad argument");
    hipLaunchKernelGGL(transpose_cast_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, as_stream(stream), table, (int)n);
    return check_launch("transpose_cast_batched");
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
    ProfScope prof(PROF_DROPOUT, (double)n * 2.0 * (dtype == LECCR_BF16 ? 2.0 : 4.0), as_stream(stream));
    if (dtype == LECCR_F32) hipLaunchKernelGGL((dropout_kernel<float>), dim3(stream_grid(n)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, n, d, (int64_t)0);
    else if (dtype == LECCR_BF16) {
        const bf16_t *xb = (const bf16_t*)x;
        bf16_t* yb = (bf16_t*)y;
        int64_t n8 = (((uintptr_t)x | (uintptr_t)y) & 15) == 0 ? n / 8 : 0;          // 16-byte part, then the scalar tail
        if (n8) hipLaunchKernelGGL(dropout8_kernel, dim3(stream_grid(n8)), dim3(256), 0, as_stream(stream), xb, yb, n8, d);
        if (n8 * 8 < n) {
            hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(stream_grid(n - n8 * 8)), dim3(256), 0, as_stream(stream), xb + n8 * 8,
                               yb + n8 * 8, n - n8 * 8, d, n8 * 8);
        }
    }
    else { set_error("dropout: bad dtype"); return LECCR_EINVAL; }
    return check_launch("dropout");
}
