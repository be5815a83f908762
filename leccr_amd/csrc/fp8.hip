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

__global__ void fp8_update_kernel(float* __restrict__ sites, int64_t n, float keep) {
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
        // keep > 0 (leccr_fp8_update_keep): the next step's running maximum starts at keep x this step's instead of 0.  Every producer
        // compares its wave's maximum with the stored amax before issuing the (same-address) atomic: starting from 0, the ~5000 resident
        // waves of the step's first launch on a site ALL find theirs above it - 86 us of serialised atomics in front of a 171 us LayerNorm
        // pass, 70 us in front of a 30 us one (scripts/ln_ex_bench.py).  From 0.9 x the previous maximum only the rows that come within
        // 10 % of it do (~1 % of them), and the recorded amax = max(0.9 x previous, this step's own): never below the true one (no
        // overflow), at most 10 % above it, decaying 10 % per step when a tensor shrinks (its power-of-two scale follows within 7 steps).
        s[0] = keep * amax;
        return;
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

// ---- weights: one batched requantisation per optimizer step (current scaling) ----------------------------------------------------
// table[i] = {first tile, bf16 source [R, C], e4m3 copy [R, C], e4m3 copy [C, R], R, C, site}: 64 x 64 tiles, one workgroup each.
// Three launches for ALL weights (config E: 192 Linears were 768 launches of 18-58 us, 48 ms of kernel time per step):
// amax per tile -> site.amax; scale from it (site.amax back to 0); both layouts written from one read of the tile.
constexpr int WT = 64, WTAB = 7;

__device__ __forceinline__ const int64_t* weight_entry(const int64_t* __restrict__ table, int n) {
    int lo = 0, hi = n - 1;                                  // last weight whose first tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * WTAB] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    return table + lo * WTAB;
}

// thread -> row (tid >> 3) and (tid >> 3) + 32 of the tile, 8 columns at 8 (tid & 7)
template <int STAGE>
__global__ __launch_bounds__(256) void fp8_weights_kernel(const int64_t* __restrict__ table, int n) {
    const int64_t* e = weight_entry(table, n);
    const bf16_t* src = reinterpret_cast<const bf16_t*>(e[1]);
    const int R = (int)e[4], C = (int)e[5];
    float* site = reinterpret_cast<float*>(e[6]);
    const int t = (int)(blockIdx.x - e[0]), tiles_x = C / WT;
    const int r0 = (t / tiles_x) * WT, c0 = (t % tiles_x) * WT;
    const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;
    bf16x8 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) v[h] = *reinterpret_cast<const bf16x8*>(src + (int64_t)(r0 + tr + 32 * h) * C + c0 + tc);
    if constexpr (STAGE == 0) {
        float m = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i) m = fmaxf(m, fabsf((float)v[h][i]));
        block_amax(m, site);
    } else {
        __shared__ uint8_t tile[WT][WT + 8];                  // [row][column] bytes; 72-byte rows: the column reads below spread over banks
        uint8_t* q8 = reinterpret_cast<uint8_t*>(e[2]);
        uint8_t* q8t = reinterpret_cast<uint8_t*>(e[3]);
        const float sc = site[1];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float q[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = fminf(fmaxf((float)v[h][i] * sc, -448.f), 448.f);
            int w[2] = {0, 0};
            w[0] = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w[0], false);
            w[0] = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w[0], true);
            w[1] = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w[1], false);
            w[1] = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w[1], true);
            const int r = tr + 32 * h;
            *reinterpret_cast<int2*>(q8 + (int64_t)(r0 + r) * C + c0 + tc) = int2{w[0], w[1]};
            *reinterpret_cast<int2*>(&tile[r][tc]) = int2{w[0], w[1]};
        }
        __syncthreads();
        // transposed copy: thread -> column c = tid >> 2 of the tile (a row of q8t), 16 consecutive r at 16 (tid & 3)
        const int c = threadIdx.x >> 2, rb = (threadIdx.x & 3) * 16;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = (uint32_t)tile[rb + 4 * j][c] | ((uint32_t)tile[rb + 4 * j + 1][c] << 8) | ((uint32_t)tile[rb + 4 * j + 2][c] << 16) |
                   ((uint32_t)tile[rb + 4 * j + 3][c] << 24);
        *reinterpret_cast<uint4*>(q8t + (int64_t)(c0 + c) * R + r0 + rb) = uint4{o[0], o[1], o[2], o[3]};
    }
}

__global__ void fp8_update_table_kernel(const int64_t* __restrict__ table, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* s = reinterpret_cast<float*>(table[i * WTAB + 6]);
    const float amax = s[0];
    if (amax > 0.f && isfinite(amax)) {
        const float sc = exp2f(fminf(floorf(log2f(448.f) - log2f(amax)), 100.f));      // as fp8_update_kernel, e4m3
        s[1] = sc;
        s[2] = 1.f / sc;
    }
    s[0] = 0.f;
}

}  // namespace
}  // namespace leccr

using namespace leccr;

extern "C" int leccr_fp8_requantize_weights(const int64_t* table, int64_t n, int64_t total_tiles, void* stream) {
    LECCR_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31) && n < (1ll << 24), LECCR_EINVAL, "fp8_requantize_weights: bad argument");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(fp8_weights_kernel<0>, dim3((unsigned)total_tiles), dim3(256), 0, st, table, (int)n);
    hipLaunchKernelGGL(fp8_update_table_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, table, (int)n);
    hipLaunchKernelGGL(fp8_weights_kernel<1>, dim3((unsigned)total_tiles), dim3(256), 0, st, table, (int)n);
    return check_launch("fp8_requantize_weights");
}

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

extern "C" int leccr_fp8_update_keep(float* sites, int64_t n, float keep, void* stream) {
    LECCR_REQUIRE(sites && n >= 0 && keep >= 0.f && keep < 1.f, LECCR_EINVAL, "fp8_update: bad argument (0 <= keep < 1)");
    if (n == 0) return LECCR_OK;
    hipLaunchKernelGGL(fp8_update_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), sites, n, keep);
    return check_launch("fp8_update");
}

extern "C" int leccr_fp8_update(float* sites, int64_t n, void* stream) { return leccr_fp8_update_keep(sites, n, 0.f, stream); }

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
