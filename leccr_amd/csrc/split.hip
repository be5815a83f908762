// split.hip - fp32 GEMM operands as bf16 (hi, lo) pairs: the reference-precision path on the bf16 matrix cores ("bf16x3").
//
// The reference computes in fp32 throughout (LECCR/models/xvlm.py:88 `clip_encoder.float()`, clip/model.py:171-192).  gfx950's
// f32 MFMA (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s; the bf16 MFMA at 2.5 PFLOP/s.  An f32 value x is hi + lo + e with
// hi = bf16(x), lo = bf16(x - hi), |e| <= 2^-18 |x|, so
//        x * w  =  hi_x hi_w + hi_x lo_w + lo_x hi_w  + O(2^-17 |x w|)
// and the three bf16 products - accumulated in f32 by the MFMA - are ONE bf16 GEMM over a reduction index three times as long:
//        [ hi_x | hi_x | lo_x ] . [ hi_w | lo_w | hi_w ]          (segment by segment: hh, hl, lh)
// The copies written here carry the segments side by side in every row, [rows, 3 * cols]:
//   form 0 ("X": activations, the [in, out] weight copy of the dX product):  hi | hi | lo
//   form 1 ("G": gradients, the [out, in] weight of the forward product):    hi | lo | hi
// An X-form operand against a G-form operand pairs up as (hh, hl, lh) whichever of the two is A or B.  The same memory viewed as
// [3 * rows, cols] is the row-interleaved concatenation the weight-gradient product dW = dY^T X needs (reduction over rows): no
// second copy for the TN form.
//
// transpose = 1 writes the split of src^T ([cols, 3 * rows]): the K-contiguous [in, out] weight copy for dX.
// HBM-bound: 4 bytes read + 6 written per element.
#include "common.h"

namespace leccr {

__device__ __forceinline__ void split2(float x, bf16_t& hi, bf16_t& lo) {
    hi = static_cast<bf16_t>(x);                       // round to nearest even
    lo = static_cast<bf16_t>(x - static_cast<float>(hi));   // exact difference, rounded once
}

// one thread: 8 consecutive elements of a row -> three 16-byte stores
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows,
                                                     int64_t cols, int64_t ld, int form) {
    const int64_t chunks = cols >> 3;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= rows * chunks) return;
    const int64_t r = id / chunks, c = (id - r * chunks) << 3;
    const float* s = src + r * ld + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
    bf16x8 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16_t hh, ll;
        split2(a[i], hh, ll); h[i] = hh; l[i] = ll;
        split2(b[i], hh, ll); h[4 + i] = hh; l[4 + i] = ll;
    }
    bf16_t* d = dst + r * 3 * cols + c;
    *reinterpret_cast<bf16x8*>(d) = h;
    *reinterpret_cast<bf16x8*>(d + cols) = form == 0 ? h : l;
    *reinterpret_cast<bf16x8*>(d + 2 * cols) = form == 0 ? l : h;
}

// The same with the column sums of src riding along (colsum[c] += sum_r src[r][c]: the bias gradient when src is a dY - one pass
// over dY instead of the split pass plus leccr_colsum's).  A thread keeps ONE 8-column chunk for all the rows it visits (block =
// rpb rows x `chunks` threads, blocks stride over the rows), so its eight partial sums stay in registers; the rpb row-threads of a
// chunk are added up in LDS and each block issues one f32 atomic per column.
__global__ __launch_bounds__(1024) void split3_colsum_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows,
                                                             int64_t cols, int64_t ld, int form, float* __restrict__ colsum, int rpb) {
    extern __shared__ float part[];                    // [rpb][cols]
    const int chunks = (int)(cols >> 3);
    const int ch = threadIdx.x % chunks, rr = threadIdx.x / chunks;
    const int64_t c = (int64_t)ch << 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t r = (int64_t)blockIdx.x * rpb + rr; r < rows; r += (int64_t)gridDim.x * rpb) {
        const float* s = src + r * ld + c;
        const f32x4 a = *reinterpret_cast<const f32x4*>(s), b = *reinterpret_cast<const f32x4*>(s + 4);
        bf16x8 h, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16_t hh, ll;
            split2(a[i], hh, ll); h[i] = hh; l[i] = ll; acc[i] += a[i];
            split2(b[i], hh, ll); h[4 + i] = hh; l[4 + i] = ll; acc[4 + i] += b[i];
        }
        bf16_t* d = dst + r * 3 * cols + c;
        *reinterpret_cast<bf16x8*>(d) = h;
        *reinterpret_cast<bf16x8*>(d + cols) = form == 0 ? h : l;
        *reinterpret_cast<bf16x8*>(d + 2 * cols) = form == 0 ? l : h;
    }
    float* mine = part + (size_t)rr * cols + c;
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[i] = acc[i];
    __syncthreads();
    for (int64_t col = threadIdx.x; col < cols; col += blockDim.x) {
        float t = 0.f;
        for (int j = 0; j < rpb; ++j) t += part[(size_t)j * cols + col];
        atomicAdd(colsum + col, t);
    }
}

// any cols (scalar accesses): small / odd operands
__global__ __launch_bounds__(256) void split3_scalar_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows,
                                                            int64_t cols, int64_t ld, int form) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= rows * cols) return;
    const int64_t r = id / cols, c = id - r * cols;
    bf16_t h, l;
    split2(src[r * ld + c], h, l);
    bf16_t* d = dst + r * 3 * cols + c;
    d[0] = h;
    d[cols] = form == 0 ? h : l;
    d[2 * cols] = form == 0 ? l : h;
}

// dst[c][s * rows + r] = part_s(src[r][c]): 32 x 32 tiles through LDS (weights: small, once per product)
__global__ __launch_bounds__(256) void split3_t_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows,
                                                       int64_t cols, int64_t ld, int form) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[r * ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) {
            bf16_t h, l;
            split2(tile[tx][ty + 8 * i], h, l);
            bf16_t* d = dst + c * 3 * rows + r;
            d[0] = h;
            d[rows] = form == 0 ? h : l;
            d[2 * rows] = form == 0 ? l : h;
        }
    }
}

}  // namespace leccr

using namespace leccr;

extern "C" int leccr_split3(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld, int form, int transpose, float* colsum,
                            void* stream) {
    LECCR_REQUIRE(src && dst, LECCR_EINVAL, "split3: null pointer");
    LECCR_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols, LECCR_EINVAL, "split3: bad sizes");
    LECCR_REQUIRE(form == 0 || form == 1, LECCR_EINVAL, "split3: form must be 0 (hi|hi|lo) or 1 (hi|lo|hi)");
    if (rows == 0 || cols == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PROF_SPLIT, 10.0 * (double)rows * (double)cols, st);
    bf16_t* d = reinterpret_cast<bf16_t*>(dst);
    const bool vec = (cols & 7) == 0 && (ld & 3) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0;
    if (colsum) {
        LECCR_REQUIRE(!transpose && vec && (cols >> 3) <= 1024, LECCR_EINVAL, "split3: colsum needs transpose = 0, cols %% 8 == 0, cols <= 8192, 16-byte aligned rows");
        const int chunks = (int)(cols >> 3);
        int rpb = 512 / chunks;
        if (rpb < 1) rpb = 1;
        int64_t grid = cdiv(rows, (int64_t)rpb * 8);              // >= 8 rows per thread amortise the atomics
        if (grid > 2048) grid = 2048;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(split3_colsum_kernel, dim3((unsigned)grid), dim3(chunks * rpb), (size_t)rpb * cols * sizeof(float), st, src, d, rows,
                           cols, ld, form, colsum, rpb);
        return check_launch("split3");
    }
    if (transpose) {
        hipLaunchKernelGGL(split3_t_kernel, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)), dim3(256), 0, st, src, d, rows, cols, ld, form);
    } else if (vec) {
        hipLaunchKernelGGL(split3_kernel, dim3((unsigned)cdiv(rows * (cols >> 3), 256)), dim3(256), 0, st, src, d, rows, cols, ld, form);
    } else {
        hipLaunchKernelGGL(split3_scalar_kernel, dim3((unsigned)cdiv(rows * cols, 256)), dim3(256), 0, st, src, d, rows, cols, ld, form);
    }
    return check_launch("split3");
}
