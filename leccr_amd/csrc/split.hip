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

// Six segments (leccr_split_n, nseg = 6): x = p0 + p1 + p2 + e, three bf16 pieces, |e| <= 2^-27 |x|, and the six products of order >= 2^-16
//        x w ~ p0 q0 + p0 q1 + p1 q0 + p0 q2 + p1 q1 + p2 q0          (2^-24 relative: fp32-grade, at a sixth of the bf16 rate)
// as segments   X form: p0 | p0 | p1 | p0 | p1 | p2      G form: p0 | p1 | p0 | p2 | p1 | p0     (the 3-segment forms are their first halves).
__device__ __forceinline__ int seg_piece(int form, int s) {       // which piece (0 = hi, 1 = mid, 2 = lo) segment s of the form carries
    constexpr int X[6] = {0, 0, 1, 0, 1, 2}, G[6] = {0, 1, 0, 2, 1, 0};
    return form == 0 ? X[s] : G[s];
}
__device__ __forceinline__ void split3p(float x, bf16_t (&p)[3]) {
    p[0] = static_cast<bf16_t>(x);
    const float r1 = x - static_cast<float>(p[0]);
    p[1] = static_cast<bf16_t>(r1);
    p[2] = static_cast<bf16_t>(r1 - static_cast<float>(p[1]));
}

// nseg = 6, any shape / alignment (a precision option, not a tuned path): one element per thread, optional transpose and column sums
__global__ __launch_bounds__(256) void split6_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows, int64_t cols,
                                                     int64_t ld, int form, int transpose, float* __restrict__ colsum) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= rows * cols) return;
    // consecutive threads walk the DESTINATION's contiguous dimension
    const int64_t r = transpose ? id % rows : id / cols, c = transpose ? id / rows : id % cols;
    const float x = src[r * ld + c];
    bf16_t p[3];
    split3p(x, p);
    const int64_t seg = transpose ? rows : cols;
    bf16_t* d = transpose ? dst + c * 6 * rows + r : dst + r * 6 * cols + c;
#pragma unroll
    for (int s = 0; s < 6; ++s) d[s * seg] = p[seg_piece(form, s)];
    if (colsum) atomicAdd(colsum + c, x);
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

// dst[c][s * rows + r] = part_s(src[r][c]) (weights: [out, in] -> the K-contiguous [in, 3 out] copy of the dX product, once per product).
// 64 x 64 tiles through LDS: 16-byte row reads of the source, and 16-byte stores of the copy - thread (c, g) owns 16 consecutive r of one
// destination row c (two bf16x8 per segment).  (The first version moved 2-byte pieces from 32 x 32 tiles: 34 us per weight, 4.6 ms of the
// fp32 step for 135 launches.)  Tails in either direction take the scalar path of the same tile loop.
__global__ __launch_bounds__(256) void split3_t_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows,
                                                       int64_t cols, int64_t ld, int form) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const bool full = r0 + 64 <= rows && c0 + 64 <= cols && (ld & 3) == 0 && (rows & 7) == 0 &&
                      ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if (full) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 4) + 16 * i, c4 = (tid & 15) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (r0 + r) * ld + c0 + c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[r][c4 + e] = v[e];
        }
    } else {
        for (int e = tid; e < 64 * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(r0 + r) * ld + c0 + c] : 0.f;
        }
    }
    __syncthreads();
    const int c = tid & 63, g = tid >> 6;              // destination row c0 + c, source rows r0 + 16 g .. + 15
    if (full) {
        bf16_t* d = dst + (c0 + c) * 3 * rows + r0 + 16 * g;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            bf16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bf16_t hh, ll;
                split2(tile[16 * g + 8 * k + j][c], hh, ll);
                h[j] = hh; l[j] = ll;
            }
            *reinterpret_cast<bf16x8*>(d + 8 * k) = h;
            *reinterpret_cast<bf16x8*>(d + rows + 8 * k) = form == 0 ? h : l;
            *reinterpret_cast<bf16x8*>(d + 2 * rows + 8 * k) = form == 0 ? l : h;
        }
    } else if (c0 + c < cols) {
        for (int j = 0; j < 16; ++j) {
            const int64_t r = r0 + 16 * g + j;
            if (r >= rows) break;
            bf16_t h, l;
            split2(tile[16 * g + j][c], h, l);
            bf16_t* d = dst + (c0 + c) * 3 * rows + r;
            d[0] = h;
            d[rows] = form == 0 ? h : l;
            d[2 * rows] = form == 0 ? l : h;
        }
    }
}

}  // namespace leccr

using namespace leccr;

// Every weight's two split copies in ONE launch (round 5): table[i] = {first tile, src (f32 [R, C], contiguous), dstG (bf16 [R, 3 C], form 1: the
// [out, in] weight of the forward product), dstT (bf16 [C, 3 R], form 0 of src^T: the K-contiguous [in, out] copy of the dX product), R, C};
// a block = one 64 x 64 tile.  The fp32 step used to split each weight at every use - 230 launches of 40 - 90 us for 6 - 28 MB of traffic each.
// The tile moves as in split3_t_kernel (the first batched version went back to 32 x 32 tiles and 2-byte pieces): float4 source rows; the G copy
// straight from those registers (8-byte stores, 16 lanes = 128 bytes of a segment row); the transposed copy from the LDS tile, thread
// (c = tid >> 2, q = tid & 3) = 16 consecutive r of destination row c, two bf16x8 per segment (four adjacent lanes = one 128-byte line).
__global__ __launch_bounds__(256) void split3_weights_kernel(const int64_t* __restrict__ table, int n) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n - 1;                                  // last matrix whose first tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 6] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const int64_t* e = table + lo * 6;
    const float* src = reinterpret_cast<const float*>(e[1]);
    bf16_t* dg = reinterpret_cast<bf16_t*>(e[2]);
    bf16_t* dt = reinterpret_cast<bf16_t*>(e[3]);
    const int64_t R = e[4], C = e[5];
    const int t = (int)(blockIdx.x - e[0]), tiles_x = (int)((C + 63) >> 6);
    const int64_t r0 = (int64_t)(t / tiles_x) * 64, c0 = (int64_t)(t % tiles_x) * 64;
    const int tid = threadIdx.x;
    const bool full = r0 + 64 <= R && c0 + 64 <= C && (C & 3) == 0 && (R & 7) == 0 &&
                      ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dg) | reinterpret_cast<uintptr_t>(dt)) & 15) == 0;
    if (full) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 4) + 16 * i, c4 = (tid & 15) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (r0 + r) * C + c0 + c4);
            bf16x4 h, l;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tile[r][c4 + k] = v[k];
                bf16_t hh, ll;
                split2(v[k], hh, ll);
                h[k] = hh; l[k] = ll;
            }
            bf16_t* d = dg + (r0 + r) * 3 * C + c0 + c4;                    // form 1: hi | lo | hi
            *reinterpret_cast<bf16x4*>(d) = h;
            *reinterpret_cast<bf16x4*>(d + C) = l;
            *reinterpret_cast<bf16x4*>(d + 2 * C) = h;
        }
    } else {
        for (int k = tid; k < 64 * 64; k += 256) {
            const int r = k >> 6, c = k & 63;
            const bool in = r0 + r < R && c0 + c < C;
            const float x = in ? src[(r0 + r) * C + c0 + c] : 0.f;
            tile[r][c] = x;
            if (in) {
                bf16_t h, l;
                split2(x, h, l);
                bf16_t* d = dg + (r0 + r) * 3 * C + c0 + c;
                d[0] = h; d[C] = l; d[2 * C] = h;
            }
        }
    }
    __syncthreads();
    const int c = tid >> 2, q = tid & 3;                     // destination row c0 + c, source rows r0 + 16 q .. + 15
    if (full) {
        bf16_t* d = dt + (c0 + c) * 3 * R + r0 + 16 * q;                    // form 0 of the transpose: hi | hi | lo
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            bf16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bf16_t hh, ll;
                split2(tile[16 * q + 8 * k + j][c], hh, ll);
                h[j] = hh; l[j] = ll;
            }
            *reinterpret_cast<bf16x8*>(d + 8 * k) = h;
            *reinterpret_cast<bf16x8*>(d + R + 8 * k) = h;
            *reinterpret_cast<bf16x8*>(d + 2 * R + 8 * k) = l;
        }
    } else if (c0 + c < C) {
        for (int j = 0; j < 16; ++j) {
            const int64_t r = r0 + 16 * q + j;
            if (r >= R) break;
            bf16_t h, l;
            split2(tile[16 * q + j][c], h, l);
            bf16_t* d = dt + (c0 + c) * 3 * R + r;
            d[0] = h; d[R] = h; d[2 * R] = l;
        }
    }
}

extern "C" int leccr_split3_weights_batched(const int64_t* table, int64_t n, int64_t total_tiles, void* stream) {
    LECCR_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), LECCR_EINVAL, "split3_weights_batched: bad argument");
    hipStream_t st = as_stream(stream);
    ProfScope prof(PROF_SPLIT, 16.0 * 4096.0 * (double)total_tiles, st);
    hipLaunchKernelGGL(split3_weights_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st, table, (int)n);
    return check_launch("split3_weights_batched");
}

extern "C" int leccr_split_n(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld, int form, int nseg, int transpose,
                             float* colsum, void* stream) {
    if (nseg == 3) return leccr_split3(src, dst, rows, cols, ld, form, transpose, colsum, stream);
    LECCR_REQUIRE(nseg == 6, LECCR_EINVAL, "split_n: nseg must be 3 or 6");
    LECCR_REQUIRE(src && dst, LECCR_EINVAL, "split_n: null pointer");
    LECCR_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols && (form == 0 || form == 1), LECCR_EINVAL, "split_n: bad sizes / form");
    LECCR_REQUIRE(!(colsum && transpose), LECCR_EINVAL, "split_n: colsum with transpose");
    if (rows == 0 || cols == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    ProfScope prof(PROF_SPLIT, 16.0 * (double)rows * (double)cols, st);
    hipLaunchKernelGGL(split6_kernel, dim3((unsigned)cdiv(rows * cols, 256)), dim3(256), 0, st, src, reinterpret_cast<bf16_t*>(dst), rows, cols, ld,
                       form, transpose, colsum);
    return check_launch("split_n");
}

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
        hipLaunchKernelGGL(split3_t_kernel, dim3((unsigned)cdiv(cols, 64), (unsigned)cdiv(rows, 64)), dim3(256), 0, st, src, d, rows, cols, ld, form);
    } else if (vec) {
        hipLaunchKernelGGL(split3_kernel, dim3((unsigned)cdiv(rows * (cols >> 3), 256)), dim3(256), 0, st, src, d, rows, cols, ld, form);
    } else {
        hipLaunchKernelGGL(split3_scalar_kernel, dim3((unsigned)cdiv(rows * cols, 256)), dim3(256), 0, st, src, d, rows, cols, ld, form);
    }
    return check_launch("split3");
}
