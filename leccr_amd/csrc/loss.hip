// loss.hip - all-pairs similarity + contrastive / distillation losses of the LECCR step, forward and backward.
// f32 features in, f32 losses out.  Forward score matrices are f32 matrix-core tiles (f32_nt_tile_kernel) written to a G x G
// scratch and reduced per row; the backward kernels recompute their scores from LDS-staged rows (a wave owns one row, lanes own
// columns) over a slice of the other operand.  (Reference sequences: include/leccr_hip.h.)
#include "common.h"
#include <algorithm>

namespace leccr {

// dot(a_row, b_row) over D floats; a in LDS (broadcast), b from global (each lane its own row, L2 resident)
__device__ __forceinline__ float dot_lds_glb(const float* __restrict__ a_lds, const float* __restrict__ b, int D) {
    float acc = 0.f;
    int d = 0;
    for (; d + 3 < D; d += 4) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + d);
        acc = fmaf(a_lds[d], bv[0], acc);
        acc = fmaf(a_lds[d + 1], bv[1], acc);
        acc = fmaf(a_lds[d + 2], bv[2], acc);
        acc = fmaf(a_lds[d + 3], bv[3], acc);
    }
    for (; d < D; ++d) acc = fmaf(a_lds[d], b[d], acc);
    return acc;
}


// The other operand's rows j0..j0+63 staged once per block in LDS (coalesced loads, row stride D + 4 floats so that the 16
// lanes of a ds_read_b128 group hit distinct banks): lanes reading "their own row" straight from global touched 64
// different cache lines per load instruction - at G = 2048 (8 ranks) that made these kernels several ms each.
__device__ __forceinline__ void stage_rows64(const float* __restrict__ B, int j0, int G, int D, float* __restrict__ tile) {
    const int ld = D + 4, q4 = D >> 2;
    for (int e = threadIdx.x; e < 64 * q4; e += 256) {
        const int r = e / q4, c = (e - r * q4) * 4;
        const int j = j0 + r < G ? j0 + r : G - 1;
        *reinterpret_cast<f32x4*>(tile + r * ld + c) = *reinterpret_cast<const f32x4*>(B + (int64_t)j * D + c);
    }
}

__device__ __forceinline__ float dot_lds_lds(const float* __restrict__ a_lds, const float* __restrict__ b_lds, int D) {
    float acc = 0.f;
    for (int d = 0; d < D; d += 4) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a_lds + d), bv = *reinterpret_cast<const f32x4*>(b_lds + d);
        acc = fmaf(av[0], bv[0], acc);
        acc = fmaf(av[1], bv[1], acc);
        acc = fmaf(av[2], bv[2], acc);
        acc = fmaf(av[3], bv[3], acc);
    }
    return acc;
}

// ----------------------------------------------------------------------------------------------- f32 score tiles
// C[M,N] = alpha * A[M,K] . B[N,K]^T in f32 on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products at the vector
// rate, but a 32 x 32 tile of scores costs 64 operand floats per k instead of 2048 LDS reads).  Block = 64 x 64 scores, 4 waves
// of 32 x 32; both operands' 64 rows are staged for a K chunk of up to 256 (row stride K chunk + 4 floats).  The MFMA's two
// k slots may hold ANY two k of the chunk as long as A and B agree, so lane (row, half) walks the contiguous half
// [half * KC/2, (half + 1) * KC/2) of its row with ds_read_b128.  MODE 0: store, 1: C = max(C, .) (max over the caption queries),
// 2: the same with the index q of the launch that holds the maximum kept in arg (same leading dimension as C).
// (The all-pairs loss kernels spent 2 LDS reads per FMA: 0.5 ms for one G = 2048 similarity, 1.6 ms for the three of dstl.)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr int kScoreKC = 256;

template <int MODE>
__global__ __launch_bounds__(256) void f32_nt_tile_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                          int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N,
                                                          int K, float alpha, const float* __restrict__ inv_alpha_dev,
                                                          int last, int32_t* __restrict__ arg, int q) {
    // alpha: the scale of the stored result; inv_alpha_dev (optional) = DEVICE scalar whose reciprocal replaces it (the learnable
    // temperature, read at run time: no host read of a device scalar in the training step).  MODE 1 / 2 take the running maximum of
    // the RAW products over the launches q = 0, 1, .. and scale only in the launch flagged `last` - max first, then / temp, as
    // model_retrieval_caption.py:148 orders it (right for either sign of temp).
    extern __shared__ float sm[];
    if (inv_alpha_dev) alpha = 1.f / inv_alpha_dev[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    constexpr int LD = kScoreKC + 4;
    float* As = sm;
    float* Bs = sm + 64 * LD;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += kScoreKC) {
        const int kc = min(kScoreKC, K - k0), q4 = kc >> 2;          // K % 8 == 0 (launcher)
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * q4; e += 256) {
            const int r = e / q4, c = (e - r * q4) * 4;
            const int ra = m0 + r < M ? m0 + r : M - 1, rb = n0 + r < N ? n0 + r : N - 1;
            *reinterpret_cast<f32x4*>(As + r * LD + c) = *reinterpret_cast<const f32x4*>(A + (int64_t)ra * lda + k0 + c);
            *reinterpret_cast<f32x4*>(Bs + r * LD + c) = *reinterpret_cast<const f32x4*>(B + (int64_t)rb * ldb + k0 + c);
        }
        __syncthreads();
        const int half = kc >> 1;
        const float* ap = As + (wr * 32 + (lane & 31)) * LD + (lane >> 5) * half;
        const float* bp = Bs + (wc * 32 + (lane & 31)) * LD + (lane >> 5) * half;
        for (int s4 = 0; s4 < half; s4 += 4) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(ap + s4), bv = *reinterpret_cast<const f32x4*>(bp + s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
        }
    }
    // acc[i] = C[m0 + 32 wr + 8 (i >> 2) + 4 (lane >> 5) + (i & 3)][n0 + 32 wc + (lane & 31)]
    const int n = n0 + wc * 32 + (lane & 31);
    if (n >= N) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + wr * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        if (m < M) {
            float* cp = C + (int64_t)m * ldc + n;
            if constexpr (MODE == 2) {               // running max over launches q = 0, 1, ..: the first maximum wins (torch.max)
                float cur = q == 0 ? acc[i] : *cp;
                if (q == 0 || acc[i] > cur) {
                    cur = acc[i];
                    arg[(int64_t)m * ldc + n] = q;
                }
                *cp = last ? cur * alpha : cur;
            } else if constexpr (MODE == 1) {
                const float cur = fmaxf(*cp, acc[i]);
                *cp = last ? cur * alpha : cur;
            } else {
                *cp = alpha * acc[i];
            }
        }
    }
}

static int launch_f32_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                         int64_t K, float alpha, bool take_max, hipStream_t st, int32_t* arg = nullptr, int q = 0, int last = 1,
                         const float* inv_alpha_dev = nullptr) {
    const size_t lds = (size_t)2 * 64 * (kScoreKC + 4) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)f32_nt_tile_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)f32_nt_tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)f32_nt_tile_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 64));
    if (arg) hipLaunchKernelGGL(f32_nt_tile_kernel<2>, grid, dim3(256), lds, st, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha, inv_alpha_dev, last, arg, q);
    else if (take_max) hipLaunchKernelGGL(f32_nt_tile_kernel<1>, grid, dim3(256), lds, st, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha, inv_alpha_dev, last, arg, q);
    else hipLaunchKernelGGL(f32_nt_tile_kernel<0>, grid, dim3(256), lds, st, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, alpha, inv_alpha_dev, last, arg, q);
    return check_launch("f32_nt_tile");
}

// Row statistics of a stored G x G logits matrix with the soft labels of get_contrastive_loss (lab_ij = [idx_i == idx_j] / cnt_i):
// lse_i, and loss += scale * (lse_i - mean of the matching logits).  One wave per row, coalesced.
__global__ __launch_bounds__(256) void sim_rowstat_kernel(const float* __restrict__ logits, const int64_t* __restrict__ idx, int G,
                                                          float* __restrict__ lse_out, float* __restrict__ loss, float scale) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= G) return;
    const float* row = logits + (int64_t)i * G;
    const int64_t my = idx ? idx[i] : 0;
    float m = -INFINITY, pos = 0.f, cnt = 0.f;
    for (int j = lane; j < G; j += 64) {
        const float s = row[j];
        m = fmaxf(m, s);
        if (idx ? (idx[j] == my) : (j == i)) {
            pos += s;
            cnt += 1.f;
        }
    }
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < G; j += 64) l += __expf(row[j] - m);
    l = wave_sum(l);
    pos = wave_sum(pos);
    cnt = wave_sum(cnt);
    if (lane == 0) {
        const float lse = m + __logf(l);
        lse_out[i] = lse;
        atomicAdd(loss, scale * (lse - pos / cnt));
    }
}


// ----------------------------------------------------------------------------------------------- sim_infonce forward, one product
// get_contrastive_loss's forward (models/xvlm.py:273-292) without a G x G logits matrix in memory: ONE f32 matrix-core product
// a b^T serves both softmax directions.  A block computes a 64 x 64 score tile (the f32_nt_tile_kernel loop), parks it in the
// LDS the operands occupied and reduces it twice - along its rows (image -> text direction: row i over the tile's 64 columns) and
// along its columns (text -> image: column j over the tile's 64 rows) - into online-softmax partials (max, sum of exp, sum of the
// matching logits, match count).  sim_merge_kernel combines a row's (column's) partials over the tiles into lse and the loss.
// Workspace: 2 G (lse) + 2 * G * ceil(G / 64) * 4 floats (2 MB at G = 2048, against 16.8 MB of logits written and read twice).
__global__ __launch_bounds__(256) void sim_tile_stats_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                             const int64_t* __restrict__ idx, int G, int D, float inv_temp, const float* __restrict__ temp_dev,
                                                             f32x4* __restrict__ part, int both_dirs) {
    if (temp_dev) inv_temp = 1.f / temp_dev[0];          // the learnable temperature, read on the device
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int T = gridDim.x;                                  // tiles per row of the score matrix (= per column: it is square)
    constexpr int LD = kScoreKC + 4;
    float* As = sm;
    float* Bs = sm + 64 * LD;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < D; k0 += kScoreKC) {
        const int kc = min(kScoreKC, D - k0), q4 = kc >> 2;          // D % 8 == 0 (launcher)
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * q4; e += 256) {
            const int r = e / q4, c = (e - r * q4) * 4;
            const int ra = m0 + r < G ? m0 + r : G - 1, rb = n0 + r < G ? n0 + r : G - 1;
            *reinterpret_cast<f32x4*>(As + r * LD + c) = *reinterpret_cast<const f32x4*>(A + (int64_t)ra * D + k0 + c);
            *reinterpret_cast<f32x4*>(Bs + r * LD + c) = *reinterpret_cast<const f32x4*>(B + (int64_t)rb * D + k0 + c);
        }
        __syncthreads();
        const int half = kc >> 1;
        const float* ap = As + (wr * 32 + (lane & 31)) * LD + (lane >> 5) * half;
        const float* bp = Bs + (wc * 32 + (lane & 31)) * LD + (lane >> 5) * half;
        for (int s4 = 0; s4 < half; s4 += 4) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(ap + s4), bv = *reinterpret_cast<const f32x4*>(bp + s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
        }
    }
    __syncthreads();                                          // every wave is done with the operand tiles: reuse their LDS
    constexpr int SL = 65;                                    // score tile row stride (floats)
    float* S = sm;
    int64_t* ids = reinterpret_cast<int64_t*>(sm + 64 * SL);             // [0..63] the tile's rows, [64..127] its columns (byte 16640)
    {
        const int n = wc * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 16; ++i) S[(wr * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)) * SL + n] = acc[i] * inv_temp;
        if (threadIdx.x < 128) {
            const int g = (threadIdx.x < 64 ? m0 : n0 - 64) + threadIdx.x;
            ids[threadIdx.x] = g < G ? (idx ? idx[g] : (int64_t)g) : (int64_t)-1 - threadIdx.x;      // padding matches nothing
        }
    }
    __syncthreads();
    // thread t: line t >> 2 (a row in pass 0, a column in pass 1), elements 16 (t & 3) .. + 15 of it; 4 threads merge by shuffle
    const int line = threadIdx.x >> 2, q = threadIdx.x & 3;
    for (int pass = 0; pass < (both_dirs ? 2 : 1); ++pass) {
        const int gl = (pass == 0 ? m0 : n0) + line;          // global index of this line
        const int o0 = pass == 0 ? n0 : m0;                   // global index of the other axis' first element
        const int64_t my = ids[pass * 64 + line];
        const int64_t* other = ids + (1 - pass) * 64;
        float m = -INFINITY, l = 0.f, pos = 0.f, cnt = 0.f;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int o = q * 16 + e;
            const float s = pass == 0 ? S[line * SL + o] : S[o * SL + line];
            v[e] = o0 + o < G ? s : -INFINITY;
            m = fmaxf(m, v[e]);
            if (other[o] == my) {
                pos += s;
                cnt += 1.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) l += v[e] == -INFINITY ? 0.f : __expf(v[e] - m);
#pragma unroll
        for (int d = 1; d < 4; d <<= 1) {                     // merge the 4 quarter-lines (adjacent lanes)
            const float m2 = __shfl_xor(m, d), l2 = __shfl_xor(l, d);
            pos += __shfl_xor(pos, d);
            cnt += __shfl_xor(cnt, d);
            const float mn = fmaxf(m, m2);
            l = (m == -INFINITY ? 0.f : l * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : l2 * __expf(m2 - mn));
            m = mn;
        }
        if (q == 0 && gl < G) {
            const int tile = pass == 0 ? blockIdx.x : blockIdx.y;
            part[((int64_t)pass * G + gl) * T + tile] = f32x4{m, l, pos, cnt};
        }
    }
}

// lse and loss of line i (a row: dir 0, a column: dir 1) from its per-tile partials.  One wave per line.
__global__ __launch_bounds__(256) void sim_merge_kernel(const f32x4* __restrict__ part, int G, int T, float* __restrict__ ws,
                                                        float* __restrict__ loss, float scale) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), dir = blockIdx.y;
    if (i >= G) return;
    const f32x4* p = part + ((int64_t)dir * G + i) * T;
    float m = -INFINITY, l = 0.f, pos = 0.f, cnt = 0.f;
    for (int t = lane; t < T; t += 64) {
        const f32x4 v = p[t];
        const float mn = fmaxf(m, v[0]);
        l = (m == -INFINITY ? 0.f : l * __expf(m - mn)) + (v[0] == -INFINITY ? 0.f : v[1] * __expf(v[0] - mn));
        m = mn;
        pos += v[2];
        cnt += v[3];
    }
    const float M = wave_max(m);
    l = wave_sum(m == -INFINITY ? 0.f : l * __expf(m - M));
    pos = wave_sum(pos);
    cnt = wave_sum(cnt);
    if (lane == 0) {
        const float lse = M + __logf(l);
        ws[dir * G + i] = lse;
        atomicAdd(loss, scale * (lse - pos / cnt));
    }
}

// ----------------------------------------------------------------------------------------------- sim_infonce
// dir 0: rows of a vs all b ; dir 1: rows of b vs all a.   For row i: lse_i, pos_i = sum_j lab_ij logit_ij.
// loss += scale * (lse_i - pos_i).   ws: [lse dir0 (G) | lse dir1 (G)]
__global__ __launch_bounds__(256) void sim_rowlse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const int64_t* __restrict__ idx, float inv_temp, const float* __restrict__ temp_dev, int G, int D,
                                                         float* __restrict__ loss, float* __restrict__ ws,
                                                         float scale) {
    if (temp_dev) inv_temp = 1.f / temp_dev[0];          // the learnable temperature, read on the device
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dir = blockIdx.y;
    const int i_raw = blockIdx.x * 4 + wave;
    const int i = i_raw < G ? i_raw : G - 1;          // no early exit: the block synchronises on the staged tiles
    const float* A = dir == 0 ? a : b;
    const float* B = dir == 0 ? b : a;
    float* ar = sm + wave * D;
    float* tile = sm + 4 * D;
    for (int d = lane; d < D; d += 64) ar[d] = A[(int64_t)i * D + d];
    const int64_t my = idx ? idx[i] : 0;
    float m = -INFINITY, l = 0.f, pos = 0.f, cnt = 0.f;
    for (int j0 = 0; j0 < G; j0 += 64) {
        __syncthreads();
        stage_rows64(B, j0, G, D, tile);
        __syncthreads();
        const int j = j0 + lane;
        float s = -INFINITY;
        if (j < G) {
            s = dot_lds_lds(ar, tile + lane * (D + 4), D) * inv_temp;
            const bool match = idx ? (idx[j] == my) : (j == i);
            if (match) {
                pos += s;
                cnt += 1.f;
            }
        }
        const float mn = fmaxf(m, wave_max(s));
        l = l * __expf(m - mn) + wave_sum(j < G ? __expf(s - mn) : 0.f);
        m = mn;
    }
    if (i_raw >= G) return;
    pos = wave_sum(pos);
    cnt = wave_sum(cnt);
    if (lane == 0) {
        const float lse = m + __logf(l);
        ws[dir * G + i] = lse;
        atomicAdd(loss, scale * (lse - pos / cnt));
    }
}

// Backward of both directions for the rows of one operand.
//  side 0: da_i = sum_j g_ij b_j / temp   side 1: db_j = sum_i g_ij a_i / temp
//  g_ij = gs * [ w_r (softmax_row(i,j) - lab_ij) + w_c (softmax_col(i,j) - lab_ij) ],   w_r = w_c = 1/(2G) (both
//  dirs) or w_r = 1/G, w_c = 0 (single direction).  dtemp += -sum_ij g_ij logit_ij / temp   (side 0 only).
__global__ __launch_bounds__(256) void sim_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const int64_t* __restrict__ idx, float inv_temp, const float* __restrict__ temp_dev, int G, int D,
                                                      const float* __restrict__ ws, float w_r, float w_c,
                                                      float* __restrict__ da, float* __restrict__ db,
                                                      float* __restrict__ dtemp, int row0, int row_end, int slice) {
    if (temp_dev) inv_temp = 1.f / temp_dev[0];          // the learnable temperature, read on the device
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int side = blockIdx.y;
    // blockIdx.z: a slice of the other operand's rows j (gridDim.z > 1: the slices add into zeroed output rows, so that a shard
    // of 256 rows against G = 2048 is 1024 workgroups instead of 128 walking all of G)
    const int j_beg = blockIdx.z * slice, j_end = min(G, j_beg + slice);
    const int i_raw = row0 + blockIdx.x * 4 + wave;   // only rows [row0, row_end) get their gradient (the caller's shard)
    const int i = i_raw < row_end ? i_raw : row_end - 1;          // no early exit: the block synchronises on the staged tiles
    const float* A = side == 0 ? a : b;
    const float* B = side == 0 ? b : a;
    float* out = side == 0 ? da : db;
    // row statistics for "my" row i and for the other operand's rows
    const float* lse_mine = ws + side * G;        // side 0: row lse ; side 1: column lse
    const float* lse_other = ws + (1 - side) * G;
    const float w_mine = side == 0 ? w_r : w_c, w_oth = side == 0 ? w_c : w_r;
    float* ar = sm + wave * (D + 64);
    float* gbuf = ar + D;
    float* tile = sm + 4 * (D + 64);
    const int ldt = D + 4;
    for (int d = lane; d < D; d += 64) ar[d] = A[(int64_t)i * D + d];
    const int64_t my = idx ? idx[i] : 0;
    // label normaliser: cnt_i (the label matrix is symmetric: lab_ij = eq/cnt_i, lab_ji = eq/cnt_j, cnt_i == cnt_j on a match)
    float cnt = 0.f;
    for (int j = lane; j < G; j += 64) cnt += (idx ? (idx[j] == my) : (j == i)) ? 1.f : 0.f;
    cnt = wave_sum(cnt);
    const float li = w_mine != 0.f ? lse_mine[i] : 0.f;
    float acc[8];  // D <= 512: lane owns columns lane, lane+64, ...
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    float dt = 0.f;
    for (int j0 = j_beg; j0 < j_end; j0 += 64) {
        __syncthreads();
        stage_rows64(B, j0, G, D, tile);
        __syncthreads();
        const int j = j0 + lane;
        float g = 0.f;
        if (j < j_end) {
            const float s = dot_lds_lds(ar, tile + lane * ldt, D) * inv_temp;
            const float lab = (idx ? (idx[j] == my) : (j == i)) ? 1.f / cnt : 0.f;
            if (w_mine != 0.f) g += w_mine * (__expf(s - li) - lab);
            if (w_oth != 0.f) g += w_oth * (__expf(s - lse_other[j]) - lab);  // unused direction: statistics never written
            dt -= g * s;
        }
        gbuf[lane] = g;
        __syncthreads();
        const int jn = j_end - j0 < 64 ? j_end - j0 : 64;
        for (int jj = 0; jj < jn; ++jj) {
            const float gj = gbuf[jj];
            const float* br = tile + jj * ldt;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int d = lane + 64 * q;
                if (d < D) acc[q] = fmaf(gj, br[d], acc[q]);
            }
        }
    }
    if (i_raw >= row_end) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int d = lane + 64 * q;
        if (d < D) {
            if (gridDim.z > 1) atomicAdd(out + (int64_t)i * D + d, acc[q] * inv_temp);
            else out[(int64_t)i * D + d] = acc[q] * inv_temp;
        }
    }
    if (side == 0 && dtemp) {
        dt = wave_sum(dt);
        if (lane == 0) atomicAdd(dtemp, dt * inv_temp);
    }
}

// The same gradient for wide rows (D > 512: the pooled cv vectors of a ViT-L/14 tower, D = 768; the 4096-wide frame features of
// the video model), where a 64-row LDS tile of the other operand no longer fits: a wave owns one row (kept in LDS), each lane
// scores one row of the other operand straight from global memory (G x D is L2-resident) - the coefficients of the whole slice
// (<= 512 rows: 8 per lane) stay in registers - then the wave walks the slice once per 1024-column chunk in 4-row load batches
// with the coefficient broadcast from its lane.  Always adds into zeroed output rows.
__global__ __launch_bounds__(256) void sim_bwd_wide_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const int64_t* __restrict__ idx, float inv_temp, const float* __restrict__ temp_dev, int G, int D,
                                                           const float* __restrict__ ws, float w_r, float w_c,
                                                           float* __restrict__ da, float* __restrict__ db,
                                                           float* __restrict__ dtemp, int row0, int row_end, int slice) {
    if (temp_dev) inv_temp = 1.f / temp_dev[0];          // the learnable temperature, read on the device
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int side = blockIdx.y;
    const int j_beg = blockIdx.z * slice, j_end = min(G, j_beg + slice);       // slice <= 512 (launcher)
    const int i = row0 + blockIdx.x * 4 + wave;
    if (i >= row_end) return;                          // no block-wide synchronisation in this kernel
    const float* A = side == 0 ? a : b;
    const float* B = side == 0 ? b : a;
    float* out = side == 0 ? da : db;
    const float* lse_mine = ws + side * G;
    const float* lse_other = ws + (1 - side) * G;
    const float w_mine = side == 0 ? w_r : w_c, w_oth = side == 0 ? w_c : w_r;
    float* ar = sm + wave * D;
    for (int d = lane; d < D; d += 64) ar[d] = A[(int64_t)i * D + d];
    __builtin_amdgcn_wave_barrier();
    const int64_t my = idx ? idx[i] : 0;
    float cnt = 0.f;
    for (int j = lane; j < G; j += 64) cnt += (idx ? (idx[j] == my) : (j == i)) ? 1.f : 0.f;
    cnt = wave_sum(cnt);
    const float li = w_mine != 0.f ? lse_mine[i] : 0.f;
    float gs[8];
    float dt = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int j = j_beg + 64 * t + lane;
        float g = 0.f;
        if (j < j_end) {
            const float s = dot_lds_glb(ar, B + (int64_t)j * D, D) * inv_temp;
            const float lab = (idx ? (idx[j] == my) : (j == i)) ? 1.f / cnt : 0.f;
            if (w_mine != 0.f) g += w_mine * (__expf(s - li) - lab);
            if (w_oth != 0.f) g += w_oth * (__expf(s - lse_other[j]) - lab);
            dt -= g * s;
        }
        gs[t] = g;
    }
    for (int c0 = 0; c0 < D; c0 += 1024) {
        float acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j0 = j_beg + 64 * t;
            if (j0 >= j_end) continue;
            const int jn = min(64, j_end - j0);
#pragma unroll 1
            for (int u0 = 0; u0 < jn; u0 += 4) {
                float rv[4][16];
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) {
                    const float* row = B + (int64_t)min(j0 + u0 + uu, G - 1) * D + c0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int d = lane + 64 * q;
                        rv[uu][q] = c0 + d < D ? row[d] : 0.f;
                    }
                }
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) {
                    const float gu = __shfl(gs[t], u0 + uu);               // lanes past jn hold g = 0
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = fmaf(gu, rv[uu][q], acc[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int d = c0 + lane + 64 * q;
            if (d < D) atomicAdd(out + (int64_t)i * D + d, acc[q] * inv_temp);
        }
    }
    if (side == 0 && dtemp) {
        dt = wave_sum(dt);
        if (lane == 0) atomicAdd(dtemp, dt * inv_temp);
    }
}

// ----------------------------------------------------------------------------------------------- capsim
// logits[i,j] = max_n <c[i,n,:], t[j,:]> / temp  (b x b, local batch).  ws: [logits b*b | lse_row b | lse_col b]
__global__ __launch_bounds__(256) void capsim_logits_kernel(const float* __restrict__ c, const float* __restrict__ t,
                                                            float inv_temp, const float* __restrict__ temp_dev, int b, int n, int D,
                                                            float* __restrict__ logits, int32_t* __restrict__ argmax) {
    if (temp_dev) inv_temp = 1.f / temp_dev[0];          // the learnable temperature, read on the device
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= b) return;
    float* cr = sm + wave * n * D;
    for (int d = lane; d < n * D; d += 64) cr[d] = c[(int64_t)i * n * D + d];
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < b; j += 64) {
        float best = -INFINITY;
        int am = 0;
        for (int q = 0; q < n; ++q) {
            const float s = dot_lds_glb(cr + q * D, t + (int64_t)j * D, D);
            if (s > best) {  // first maximum wins, as torch.max
                best = s;
                am = q;
            }
        }
        logits[(int64_t)i * b + j] = best * inv_temp;
        argmax[(int64_t)i * b + j] = am;
    }
}

// row / column log-sum-exp of a stored [R,C] matrix + hard-label CE:  loss += scale * (lse_r - x[r, r])
__global__ __launch_bounds__(256) void lse_ce_kernel(const float* __restrict__ x, int R, int C, int64_t rs, int64_t cs,
                                                     float* __restrict__ lse_out, float* __restrict__ loss, float scale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float m = -INFINITY;
    for (int j = lane; j < C; j += 64) m = fmaxf(m, x[r * rs + j * cs]);
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < C; j += 64) l += __expf(x[r * rs + j * cs] - m);
    l = wave_sum(l);
    if (lane == 0) {
        const float lse = m + __logf(l);
        lse_out[r] = lse;
        if (loss) atomicAdd(loss, scale * (lse - x[r * rs + r * cs]));
    }
}

// g_ij = gs/(2b) [ (softmax_row - delta_ij) + (softmax_col - delta_ij) ]
__device__ __forceinline__ float capsim_g(const float* logits, const float* lse_r, const float* lse_c, int b, int i,
                                          int j, float w) {
    const float s = logits[(int64_t)i * b + j];
    const float dl = i == j ? 1.f : 0.f;
    return w * ((__expf(s - lse_r[i]) - dl) + (__expf(s - lse_c[j]) - dl));
}

// side 0: wave per i -> dc[i, n, :] ; side 1: wave per j -> dt[j, :].   D <= 512, n <= 8.
// blockIdx.z: a slice of the other index; per 64 of them the lanes compute the coefficients g (and fetch the arg-max query) in
// parallel, then the wave walks the 64 operand rows with g broadcast from its lane; slices add into the zeroed outputs.
__global__ __launch_bounds__(256) void capsim_bwd_kernel(const float* __restrict__ c, const float* __restrict__ t,
                                                         float inv_temp, const float* __restrict__ temp_dev, int b, int n, int D,
                                                         const int32_t* __restrict__ argmax,
                                                         const float* __restrict__ ws, float w, float* __restrict__ dc,
                                                         float* __restrict__ dt, float* __restrict__ dtemp, int slice) {
    if (temp_dev) inv_temp = 1.f / temp_dev[0];          // the learnable temperature, read on the device
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int side = blockIdx.y;
    const int r = blockIdx.x * 4 + wave;
    if (r >= b) return;
    const float* logits = ws;
    const float* lse_r = ws + (int64_t)b * b;
    const float* lse_c = lse_r + b;
    const int o_beg = blockIdx.z * slice, o_end = min(b, o_beg + slice);
    if (side == 0) {
        float acc[8][8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[q][u] = 0.f;
        float dtl = 0.f;
        for (int j0 = o_beg; j0 < o_end; j0 += 64) {
            const int j = j0 + lane;
            float g = 0.f;
            int am = 0;
            if (j < o_end) {
                g = capsim_g(logits, lse_r, lse_c, b, r, j, w);
                am = argmax[(int64_t)r * b + j];
                dtl -= g * logits[(int64_t)r * b + j];
            }
            const int cnt = min(64, o_end - j0);
            for (int u = 0; u < cnt; ++u) {
                const float gu = __shfl(g, u);
                const int qu = __builtin_amdgcn_readlane(am, u);          // wave-uniform query index
                const float* row = t + (int64_t)(j0 + u) * D;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (q == qu) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int d = lane + 64 * k;
                            if (d < D) acc[q][k] = fmaf(gu, row[d], acc[q][k]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < n) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int d = lane + 64 * k;
                    if (d < D) atomicAdd(dc + ((int64_t)r * n + q) * D + d, acc[q][k] * inv_temp);
                }
            }
        }
        dtl = wave_sum(dtl);
        if (lane == 0 && dtemp) atomicAdd(dtemp, dtl * inv_temp);
    } else {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i0 = o_beg; i0 < o_end; i0 += 64) {
            const int i = i0 + lane;
            float g = 0.f;
            int am = 0;
            if (i < o_end) {
                g = capsim_g(logits, lse_r, lse_c, b, i, r, w);
                am = argmax[(int64_t)i * b + r];
            }
            const int cnt = min(64, o_end - i0);
            for (int u = 0; u < cnt; ++u) {
                const float gu = __shfl(g, u);
                const int qu = __builtin_amdgcn_readlane(am, u);
                const float* row = c + ((int64_t)(i0 + u) * n + qu) * D;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int d = lane + 64 * k;
                    if (d < D) acc[k] = fmaf(gu, row[d], acc[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int d = lane + 64 * k;
            if (d < D) atomicAdd(dt + (int64_t)r * D + d, acc[k] * inv_temp);
        }
    }
}

// ----------------------------------------------------------------------------------------------- dstl
// ws layout (floats): [tv G*G | lab G*G (sv, then labels) | sc G*G | minmax 4 (as ordered ints) | lse_tv G]
__device__ __forceinline__ int f2ord(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void dstl_init_kernel(int* mm) {
    mm[0] = f2ord(INFINITY);   // min sv
    mm[1] = f2ord(-INFINITY);  // max sv
    mm[2] = f2ord(INFINITY);   // min sc
    mm[3] = f2ord(-INFINITY);  // max sc
}

// global min / max of the two stored score matrices (norm_score's ranges), grid-stride
__global__ __launch_bounds__(256) void dstl_minmax_kernel(const float* __restrict__ sv, const float* __restrict__ sc, int64_t n,
                                                          int* __restrict__ mm) {
    float mn_sv = INFINITY, mx_sv = -INFINITY, mn_sc = INFINITY, mx_sc = -INFINITY;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float a = sv[e], b = sc[e];
        mn_sv = fminf(mn_sv, a);
        mx_sv = fmaxf(mx_sv, a);
        mn_sc = fminf(mn_sc, b);
        mx_sc = fmaxf(mx_sc, b);
    }
    __shared__ float red[4][4];
    mn_sv = wave_min(mn_sv);
    mx_sv = wave_max(mx_sv);
    mn_sc = wave_min(mn_sc);
    mx_sc = wave_max(mx_sc);
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        red[0][wv] = mn_sv; red[1][wv] = mx_sv; red[2][wv] = mn_sc; red[3][wv] = mx_sc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                      // one set of atomics per workgroup (4 addresses)
        atomicMin(mm + 0, f2ord(fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]))));
        atomicMax(mm + 1, f2ord(fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]))));
        atomicMin(mm + 2, f2ord(fminf(fminf(red[2][0], red[2][1]), fminf(red[2][2], red[2][3]))));
        atomicMax(mm + 3, f2ord(fmaxf(fmaxf(red[3][0], red[3][1]), fmaxf(red[3][2], red[3][3]))));
    }
}

template <bool STAGED>
__global__ __launch_bounds__(256) void dstl_scores_kernel(const float* __restrict__ v, const float* __restrict__ c,
                                                          const float* __restrict__ s, const float* __restrict__ t,
                                                          int G, int n, int D, float* __restrict__ tv,
                                                          float* __restrict__ sv, float* __restrict__ sc,
                                                          int* __restrict__ mm) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i_raw = blockIdx.x * 4 + wave;
    if (!STAGED && i_raw >= G) return;
    const int i = i_raw < G ? i_raw : G - 1;     // STAGED: no early exit, the block synchronises on the tiles
    float* tr = sm + wave * (2 + n) * D;  // t_i | s_i | c_i[0..n)
    float* sr = tr + D;
    float* cr = sr + D;
    float* vtile = sm + 4 * (2 + n) * D;  // STAGED: rows j0..j0+63 of v and of s (row stride D + 4)
    float* stile = vtile + 64 * (D + 4);
    for (int d = lane; d < D; d += 64) {
        tr[d] = t[(int64_t)i * D + d];
        sr[d] = s[(int64_t)i * D + d];
    }
    for (int d = lane; d < n * D; d += 64) cr[d] = c[(int64_t)i * n * D + d];
    __builtin_amdgcn_wave_barrier();
    float mn_sv = INFINITY, mx_sv = -INFINITY, mn_sc = INFINITY, mx_sc = -INFINITY;
    for (int j0 = 0; j0 < G; j0 += 64) {
        if constexpr (STAGED) {
            __syncthreads();
            stage_rows64(v, j0, G, D, vtile);
            stage_rows64(s, j0, G, D, stile);
            __syncthreads();
        }
        const int j = j0 + lane;
        if (j >= G || i_raw >= G) continue;
        float a, bsv, best = -INFINITY;
        if constexpr (STAGED) {
            const float* vr = vtile + lane * (D + 4);
            const float* srow = stile + lane * (D + 4);
            a = dot_lds_lds(tr, vr, D);                                // logits_tv[i,j] = <t_i, v_j>
            bsv = dot_lds_lds(sr, vr, D);                              // logits_sv[i,j] = <s_i, v_j>
            for (int q = 0; q < n; ++q) best = fmaxf(best, dot_lds_lds(cr + q * D, srow, D));   // max_n <c_i,n , s_j>
        } else {
            a = dot_lds_glb(tr, v + (int64_t)j * D, D);
            bsv = dot_lds_glb(sr, v + (int64_t)j * D, D);
            for (int q = 0; q < n; ++q) best = fmaxf(best, dot_lds_glb(cr + q * D, s + (int64_t)j * D, D));
        }
        tv[(int64_t)i * G + j] = a;
        sv[(int64_t)i * G + j] = bsv;
        sc[(int64_t)i * G + j] = best;
        mn_sv = fminf(mn_sv, bsv);
        mx_sv = fmaxf(mx_sv, bsv);
        mn_sc = fminf(mn_sc, best);
        mx_sc = fmaxf(mx_sc, best);
    }
    if (i_raw >= G) return;
    mn_sv = wave_min(mn_sv);
    mx_sv = wave_max(mx_sv);
    mn_sc = wave_min(mn_sc);
    mx_sc = wave_max(mx_sc);
    if (lane == 0) {
        atomicMin(mm + 0, f2ord(mn_sv));
        atomicMax(mm + 1, f2ord(mx_sv));
        atomicMin(mm + 2, f2ord(mn_sc));
        atomicMax(mm + 3, f2ord(mx_sc));
    }
}

// per row: labels = softmax(alpha*norm(sv) + (1-alpha)*norm(sc)) (stored over sv), lse of tv, KL contribution
__global__ __launch_bounds__(256) void dstl_kl_kernel(const float* __restrict__ tv, float* __restrict__ lab,
                                                      const float* __restrict__ sc, const int* __restrict__ mm,
                                                      float alpha, int G, float* __restrict__ lse_tv,
                                                      float* __restrict__ loss) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= G) return;
    const float mn_sv = ord2f(mm[0]), mx_sv = ord2f(mm[1]), mn_sc = ord2f(mm[2]), mx_sc = ord2f(mm[3]);
    // norm_score: (x - min) / max(x - min)
    const float k_sv = alpha / (mx_sv - mn_sv), k_sc = (1.f - alpha) / (mx_sc - mn_sc);
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int j = lane; j < G; j += 64) {
        const int64_t o = (int64_t)i * G + j;
        const float x = (lab[o] - mn_sv) * k_sv + (sc[o] - mn_sc) * k_sc;
        lab[o] = x;
        m1 = fmaxf(m1, x);
        m2 = fmaxf(m2, tv[o]);
    }
    m1 = wave_max(m1);
    m2 = wave_max(m2);
    float l1 = 0.f, l2 = 0.f;
    for (int j = lane; j < G; j += 64) {
        const int64_t o = (int64_t)i * G + j;
        l1 += __expf(lab[o] - m1);
        l2 += __expf(tv[o] - m2);
    }
    l1 = wave_sum(l1);
    l2 = wave_sum(l2);
    const float lse1 = m1 + __logf(l1), lse2 = m2 + __logf(l2);
    float kl = 0.f;
    for (int j = lane; j < G; j += 64) {
        const int64_t o = (int64_t)i * G + j;
        const float loglab = lab[o] - lse1, p = __expf(loglab);
        lab[o] = p;
        kl += p > 0.f ? p * (loglab - (tv[o] - lse2)) : 0.f;
    }
    kl = wave_sum(kl);
    if (lane == 0) {
        lse_tv[i] = lse2;
        atomicAdd(loss, kl / G);
    }
}

// g_ij = gs/G (softmax(tv)_ij - lab_ij);  side 0: dt_i = sum_j g_ij v_j ; side 1: dv_j = sum_i g_ij t_i
// One wave = one output row r (of the caller's shard), one side, one slice of the reduction index o.  Per 64 o's the lanes
// compute the 64 coefficients g in parallel (one exp per lane instead of the same exp in all 64), then the wave walks the 64
// operand rows with the coefficient broadcast from its lane; slices meet by atomicAdd into the zeroed output rows.
// (One wave walking all G o's serially with a redundant exp per lane took 2.6 ms at G = 2048 for 256 rows.)
__global__ __launch_bounds__(256) void dstl_bwd_kernel(const float* __restrict__ v, const float* __restrict__ t, int G,
                                                       int D, const float* __restrict__ tv,
                                                       const float* __restrict__ lab, const float* __restrict__ lse_tv,
                                                       float w, float* __restrict__ dv, float* __restrict__ dt,
                                                       int row0, int row_end, int slice) {
    const int lane = threadIdx.x & 63;
    const int side = blockIdx.y;
    const int r = row0 + blockIdx.x * 4 + (threadIdx.x >> 6);      // only the caller's shard of rows
    if (r >= row_end) return;
    const int o_beg = blockIdx.z * slice, o_end = min(G, o_beg + slice);
    const float* src = side == 0 ? v : t;
    const float lse_r = lse_tv[side == 0 ? r : 0];                 // side 0: the row's own lse; side 1: per o below
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int o0 = o_beg; o0 < o_end; o0 += 64) {
        const int o = o0 + lane;
        float g = 0.f;
        if (o < o_end) {
            const int64_t off = side == 0 ? (int64_t)r * G + o : (int64_t)o * G + r;
            g = w * (__expf(tv[off] - (side == 0 ? lse_r : lse_tv[o])) - lab[off]);
        }
        const int cnt = min(64, o_end - o0);
        // 8 operand rows per step: their loads are independent and in flight together (one row per step was one L2 round
        // trip per o); lanes past cnt hold g = 0 and rows are clamped, so the tail needs no separate loop
#pragma unroll 1
        for (int u0 = 0; u0 < cnt; u0 += 8) {
            float rv[8][8];
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) {
                const int o = min(o0 + u0 + uu, G - 1);
                const float* row = src + (int64_t)o * D;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int d = lane + 64 * k;
                    rv[uu][k] = d < D ? row[d] : 0.f;
                }
            }
#pragma unroll
            for (int uu = 0; uu < 8; ++uu) {
                const float gu = __shfl(g, u0 + uu);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = fmaf(gu, rv[uu][k], acc[k]);
            }
        }
    }
    float* out = (side == 0 ? dt : dv) + (int64_t)r * D;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int d = lane + 64 * k;
        if (d < D) atomicAdd(out + d, acc[k]);
    }
}

// ----------------------------------------------------------------------------------------------- cvpool / capreg
// x [b, L, d]: per (b, column): S = sum_l x, nrm = max(sqrt(sum_l x^2), 1e-12); pooled = S / (L nrm)
// F.normalize(dim = 1) over the TOKEN axis followed by the mean over tokens, per (sample, column):
// pooled[b,c] = (sum_l x[b,l,c]) / max(||x[b,:,c]||, eps) / L.   Block = one sample x 256 columns; a thread owns 4
// consecutive columns (one 8/16-byte access) and every 4th token, the 4 token slices meet in LDS.  (One thread per column
// walking all L tokens serially ran at a quarter of the HBM rate.)
template <typename T>
__global__ __launch_bounds__(256) void cvpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ pooled,
                                                         float* __restrict__ inv_norm, int L, int d) {
    __shared__ float red[2][4][256];
    const int cg = threadIdx.x & 63, ls = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + cg * 4;
    const int64_t bi = blockIdx.y;
    f32x4 S = {0.f, 0.f, 0.f, 0.f}, Q = S;
    if (col < d) {
        const T* p = x + bi * L * (int64_t)d + col;
        for (int l = ls; l < L; l += 4) {
            const f32x4 v = load4(p + (int64_t)l * d);
            S += v;
            Q += v * v;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][ls][cg * 4 + j] = S[j];
        red[1][ls][cg * 4 + j] = Q[j];
    }
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < d) {
        const int t = threadIdx.x;
        const float s = red[0][0][t] + red[0][1][t] + red[0][2][t] + red[0][3][t];
        const float q = red[1][0][t] + red[1][1][t] + red[1][2][t] + red[1][3][t];
        const float inv = 1.f / fmaxf(sqrtf(q), 1e-12f);
        inv_norm[bi * d + c] = inv;
        pooled[bi * d + c] = s * inv / L;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cvpool_bwd_kernel(const T* __restrict__ x, const float* __restrict__ inv_norm,
                                                         const float* __restrict__ dpooled, T* __restrict__ dx, int L,
                                                         int d) {
    __shared__ float red[4][256];
    const int cg = threadIdx.x & 63, ls = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + cg * 4;
    const int64_t bi = blockIdx.y;
    const bool live = col < d;
    const T* p = x + bi * L * (int64_t)d + col;
    T* o = dx + bi * L * (int64_t)d + col;
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
    if (live)
        for (int l = ls; l < L; l += 4) S += load4(p + (int64_t)l * d);
#pragma unroll
    for (int j = 0; j < 4; ++j) red[ls][cg * 4 + j] = S[j];
    __syncthreads();
    if (!live) return;
    f32x4 inv, g, k;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = cg * 4 + j;
        const float s = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        inv[j] = inv_norm[bi * d + col + j];
        g[j] = dpooled[bi * d + col + j] / L;
        k[j] = s * inv[j] * inv[j] * inv[j];
    }
    for (int l = ls; l < L; l += 4) store4(o + (int64_t)l * d, g * (inv - k * load4(p + (int64_t)l * d)));
}

// caption_regularization: loss = (sum_b |sum_p c_hat[b,p]|^2 - b n) / (b n^2).  One wave per sample, d <= 4096, n <= 8.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void capreg_kernel(const T* __restrict__ c, float* __restrict__ loss,
                                                     T* __restrict__ dc, float gscale, int b, int n, int d) {
    const int lane = threadIdx.x & 63;
    const int bi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bi >= b) return;
    const float w = 1.f / ((float)b * n * n);
    float inv[8];
    for (int p = 0; p < n; ++p) {
        float q = 0.f;
        for (int e = lane; e < d; e += 64) {
            const float v = to_f32(c[((int64_t)bi * n + p) * d + e]);
            q = fmaf(v, v, q);
        }
        inv[p] = 1.f / fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    }
    if constexpr (!BWD) {
        float uu = 0.f;
        for (int e = lane; e < d; e += 64) {
            float u = 0.f;
            for (int p = 0; p < n; ++p) u += to_f32(c[((int64_t)bi * n + p) * d + e]) * inv[p];
            uu = fmaf(u, u, uu);
        }
        uu = wave_sum(uu);
        if (lane == 0) atomicAdd(loss, (uu - n) * w);
    } else {
        // g_p = 2 w u ;  dc_p = inv_p (g - chat_p <g, chat_p>)
        float gdot[8];
        for (int p = 0; p < n; ++p) gdot[p] = 0.f;
        for (int e = lane; e < d; e += 64) {
            float u = 0.f;
            for (int p = 0; p < n; ++p) u += to_f32(c[((int64_t)bi * n + p) * d + e]) * inv[p];
            for (int p = 0; p < n; ++p) gdot[p] += u * to_f32(c[((int64_t)bi * n + p) * d + e]) * inv[p];
        }
        for (int p = 0; p < n; ++p) gdot[p] = wave_sum(gdot[p]);
        const float k = 2.f * w * gscale;
        for (int e = lane; e < d; e += 64) {
            float u = 0.f;
            for (int p = 0; p < n; ++p) u += to_f32(c[((int64_t)bi * n + p) * d + e]) * inv[p];
            for (int p = 0; p < n; ++p) {
                const float ch = to_f32(c[((int64_t)bi * n + p) * d + e]) * inv[p];
                dc[((int64_t)bi * n + p) * d + e] = from_f32<T>(k * inv[p] * (u - ch * gdot[p]));
            }
        }
    }
}

}  // namespace leccr

using namespace leccr;

extern "C" int leccr_sim_infonce_fwd(const float* a, const float* b, const int64_t* idx, float temp, int64_t G, int64_t D,
                                     int both_dirs, float* loss, float* ws, const float* temp_dev, void* stream) {
    LECCR_REQUIRE(a && b && loss && ws && G > 0 && D > 0 && D <= 4096 && (temp_dev || temp != 0.f), LECCR_EINVAL, "sim_infonce_fwd: bad argument");
    if (temp_dev) temp = 1.f;               // (the kernels take 1 / *temp_dev)
    LECCR_REQUIRE((D & 3) == 0, LECCR_EALIGN, "sim_infonce_fwd: D must be a multiple of 4");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) return check_launch("sim_infonce memset");
    const float scale = both_dirs ? 0.5f / G : 1.f / G;
    if ((D & 7) == 0) {
        // one product a b^T / temp on the matrix cores, reduced tile by tile along rows AND columns (both softmax directions), then
        // a merge of the per-tile partials: no G x G logits in memory (ws: [lse dir0 G | lse dir1 G | partials 2 x G x T x 4])
        const int T = (int)cdiv(G, 64);
        f32x4* part = reinterpret_cast<f32x4*>(ws + ((2 * G + 3) & ~(int64_t)3));
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)sim_tile_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr = true;
        }
        const size_t lds = (size_t)2 * 64 * (kScoreKC + 4) * sizeof(float);
        hipLaunchKernelGGL(sim_tile_stats_kernel, dim3((unsigned)T, (unsigned)T), dim3(256), lds, st, a, b, idx, (int)G, (int)D,
                           1.f / temp, temp_dev, part, both_dirs);
        hipLaunchKernelGGL(sim_merge_kernel, dim3((unsigned)cdiv(G, 4), both_dirs ? 2 : 1), dim3(256), 0, st, part, (int)G, T, ws,
                           loss, scale);
        return check_launch("sim_infonce_fwd");
    }
    const size_t lds_f = (size_t)(4 * D + 64 * (D + 4)) * sizeof(float);      // 4 own rows + the staged 64-row tile
    (void)hipFuncSetAttribute((const void*)sim_rowlse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(sim_rowlse_kernel, dim3((unsigned)cdiv(G, 4), both_dirs ? 2 : 1), dim3(256), lds_f, st,
                       a, b, idx, 1.f / temp, temp_dev, (int)G, (int)D, loss, ws, scale);
    return check_launch("sim_infonce_fwd");
}

extern "C" int leccr_sim_infonce_bwd(const float* a, const float* b, const int64_t* idx, float temp, int64_t G, int64_t D,
                                     int both_dirs, const float* ws, float gscale, float* da, float* db, float* dtemp,
                                     int64_t row0, int64_t nrows, const float* temp_dev, void* stream) {
    LECCR_REQUIRE(a && b && ws && da && db && G > 0 && D > 0 && D <= 8192 && (D & 3) == 0 && (temp_dev || temp != 0.f), LECCR_EINVAL, "sim_infonce_bwd: bad argument (D <= 8192)");
    if (temp_dev) temp = 1.f;
    LECCR_REQUIRE(row0 >= 0 && nrows > 0 && row0 + nrows <= G, LECCR_EINVAL, "sim_infonce_bwd: row range outside [0, G)");
    hipStream_t st = as_stream(stream);
    const float w_r = gscale * (both_dirs ? 0.5f / G : 1.f / G), w_c = both_dirs ? w_r : 0.f;
    const size_t lds_b = (size_t)(4 * (D + 64) + 64 * (D + 4)) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)sim_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // slices of the reduction index so that the launch has >= ~512 workgroups (slice = multiple of the 64-row staging tile)
    int nslice = (int)cdiv(512, 2 * cdiv(nrows, 4));
    if (nslice > cdiv(G, 256)) nslice = (int)cdiv(G, 256);
    if (nslice < 1) nslice = 1;
    const int slice = (int)(cdiv(cdiv(G, nslice), 64) * 64);
    nslice = (int)cdiv(G, slice);
    if ((nslice > 1 || D > 512) && (hipMemsetAsync(da + row0 * D, 0, sizeof(float) * nrows * D, st) != hipSuccess ||
                                    hipMemsetAsync(db + row0 * D, 0, sizeof(float) * nrows * D, st) != hipSuccess))
        return check_launch("sim_infonce_bwd memset");
    if (D > 512) {
        const int wslice = slice > 512 ? 512 : slice;                  // the wide kernel keeps a slice's coefficients in 8 registers
        (void)hipFuncSetAttribute((const void*)sim_bwd_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(sim_bwd_wide_kernel, dim3((unsigned)cdiv(nrows, 4), 2, (unsigned)cdiv(G, wslice)), dim3(256),
                           (size_t)4 * D * sizeof(float), st, a, b, idx, 1.f / temp, temp_dev, (int)G, (int)D, ws, w_r, w_c, da, db, dtemp,
                           (int)row0, (int)(row0 + nrows), wslice);
        return check_launch("sim_infonce_bwd");
    }
    hipLaunchKernelGGL(sim_bwd_kernel, dim3((unsigned)cdiv(nrows, 4), 2, (unsigned)nslice), dim3(256), lds_b, st, a, b, idx,
                       1.f / temp, temp_dev, (int)G, (int)D, ws, w_r, w_c, da, db, dtemp, (int)row0, (int)(row0 + nrows), slice);
    return check_launch("sim_infonce_bwd");
}

extern "C" int leccr_capsim_fwd(const float* c, const float* t, float temp, int64_t b, int64_t n, int64_t D, float* loss,
                                int32_t* argmax, float* ws, const float* temp_dev, void* stream) {
    LECCR_REQUIRE(c && t && loss && argmax && ws && b > 0 && n > 0 && n <= 8 && D > 0 && (D & 3) == 0 && n * D <= 3072 && (temp_dev || temp != 0.f),
                  LECCR_EINVAL, "capsim_fwd: bad argument");
    if (temp_dev) temp = 1.f;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) return check_launch("capsim memset");
    float* logits = ws;
    float* lse_r = ws + b * b;
    float* lse_c = lse_r + b;
    if ((D & 7) == 0) {      // max_n <c_n, t> as n matrix-core tile passes keeping the running maximum (of the raw products) and its n
        for (int q = 0; q < n; ++q) {
            const int rc = launch_f32_nt(c + q * D, n * D, t, D, logits, b, b, b, D, 1.f / temp, true, st, argmax, q, q == n - 1, temp_dev);
            if (rc != LECCR_OK) return rc;
        }
    } else {
        hipLaunchKernelGGL(capsim_logits_kernel, dim3((unsigned)cdiv(b, 4)), dim3(256), 4 * n * D * sizeof(float), st, c, t,
                           1.f / temp, temp_dev, (int)b, (int)n, (int)D, logits, argmax);
    }
    hipLaunchKernelGGL(lse_ce_kernel, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, st, logits, (int)b, (int)b, (int64_t)b, (int64_t)1, lse_r, loss, 0.5f / b);
    hipLaunchKernelGGL(lse_ce_kernel, dim3((unsigned)cdiv(b, 4)), dim3(256), 0, st, logits, (int)b, (int)b, (int64_t)1, (int64_t)b, lse_c, loss, 0.5f / b);
    return check_launch("capsim_fwd");
}

extern "C" int leccr_capsim_bwd(const float* c, const float* t, float temp, int64_t b, int64_t n, int64_t D,
                                const int32_t* argmax, const float* ws, float gscale, float* dc, float* dt, float* dtemp,
                                const float* temp_dev, void* stream) {
    LECCR_REQUIRE(c && t && argmax && ws && dc && dt && b > 0 && n > 0 && n <= 8 && D > 0 && D <= 512 && (temp_dev || temp != 0.f), LECCR_EINVAL,
                  "capsim_bwd: bad argument");
    if (temp_dev) temp = 1.f;
    hipStream_t st = as_stream(stream);
    int nslice = (int)cdiv(1024, 2 * cdiv(b, 4));
    if (nslice > cdiv(b, 64)) nslice = (int)cdiv(b, 64);
    if (nslice < 1) nslice = 1;
    const int slice = (int)(cdiv(cdiv(b, nslice), 64) * 64);
    nslice = (int)cdiv(b, slice);
    if (hipMemsetAsync(dc, 0, sizeof(float) * b * n * D, st) != hipSuccess || hipMemsetAsync(dt, 0, sizeof(float) * b * D, st) != hipSuccess)
        return check_launch("capsim_bwd memset");
    hipLaunchKernelGGL(capsim_bwd_kernel, dim3((unsigned)cdiv(b, 4), 2, (unsigned)nslice), dim3(256), 0, st, c, t, 1.f / temp, temp_dev,
                       (int)b, (int)n, (int)D, argmax, ws, gscale * 0.5f / b, dc, dt, dtemp, slice);
    return check_launch("capsim_bwd");
}

extern "C" int leccr_dstl_fwd(const float* v, const float* c, const float* s, const float* t, float alpha, int64_t G,
                              int64_t n, int64_t D, float* loss, float* ws, void* stream) {
    LECCR_REQUIRE(v && c && s && t && loss && ws && G > 0 && n > 0 && n <= 8 && D > 0 && (D & 3) == 0 && (2 + n) * D <= 3072, LECCR_EINVAL, "dstl_fwd: bad argument");
    hipStream_t st = as_stream(stream);
    float* tv = ws;
    float* lab = ws + G * G;
    float* sc = lab + G * G;
    int* mm = reinterpret_cast<int*>(sc + G * G);
    float* lse_tv = reinterpret_cast<float*>(mm + 4);
    if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) return check_launch("dstl memset");
    hipLaunchKernelGGL(dstl_init_kernel, dim3(1), dim3(1), 0, st, mm);
    const size_t own = (size_t)4 * (2 + n) * D * sizeof(float), staged = own + (size_t)2 * 64 * (D + 4) * sizeof(float);
    if ((D & 7) == 0) {
        // the 2 + n score products as f32 matrix-core tiles: tv = t v^T, sv = s v^T, sc = max_q c_q s^T (c is [G, n, D]: row
        // stride n D), then one pass for the global ranges
        int rc = launch_f32_nt(t, D, v, D, tv, G, G, G, D, 1.f, false, st);
        if (rc == LECCR_OK) rc = launch_f32_nt(s, D, v, D, lab, G, G, G, D, 1.f, false, st);
        for (int q = 0; q < n && rc == LECCR_OK; ++q) rc = launch_f32_nt(c + q * D, n * D, s, D, sc, G, G, G, D, 1.f, q > 0, st);
        if (rc != LECCR_OK) return rc;
        hipLaunchKernelGGL(dstl_minmax_kernel, dim3((unsigned)std::min<int64_t>(512, cdiv(G * G, 1024))), dim3(256), 0, st, lab, sc,
                           G * G, mm);
    } else if (staged <= 160 * 1024 && (D & 3) == 0) {       // both 64-row tiles fit next to the block's own rows
        (void)hipFuncSetAttribute((const void*)dstl_scores_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(dstl_scores_kernel<true>, dim3((unsigned)cdiv(G, 4)), dim3(256), staged, st, v, c, s, t, (int)G, (int)n,
                           (int)D, tv, lab, sc, mm);
    } else {
        hipLaunchKernelGGL(dstl_scores_kernel<false>, dim3((unsigned)cdiv(G, 4)), dim3(256), own, st, v, c, s, t, (int)G, (int)n,
                           (int)D, tv, lab, sc, mm);
    }
    hipLaunchKernelGGL(dstl_kl_kernel, dim3((unsigned)cdiv(G, 4)), dim3(256), 0, st, tv, lab, sc, mm, alpha, (int)G, lse_tv, loss);
    return check_launch("dstl_fwd");
}

extern "C" int leccr_dstl_bwd(const float* v, const float* t, int64_t G, int64_t D, const float* ws, float gscale, float* dv,
                              float* dt, int64_t row0, int64_t nrows, void* stream) {
    LECCR_REQUIRE(v && t && ws && dv && dt && G > 0 && D > 0 && D <= 512, LECCR_EINVAL, "dstl_bwd: bad argument");
    LECCR_REQUIRE(row0 >= 0 && nrows > 0 && row0 + nrows <= G, LECCR_EINVAL, "dstl_bwd: row range outside [0, G)");
    hipStream_t st = as_stream(stream);
    const float* tv = ws;
    const float* lab = ws + G * G;
    const float* lse_tv = ws + 3 * G * G + 4;
    // the reduction over G is cut into slices so that the launch has >= ~2048 waves; the slices add into zeroed rows
    int nslice = (int)cdiv(2048, 2 * nrows);
    if (nslice > cdiv(G, 64)) nslice = (int)cdiv(G, 64);
    if (nslice < 1) nslice = 1;
    const int slice = (int)(cdiv(cdiv(G, nslice), 64) * 64);
    nslice = (int)cdiv(G, slice);
    if (hipMemsetAsync(dv + row0 * D, 0, sizeof(float) * nrows * D, st) != hipSuccess ||
        hipMemsetAsync(dt + row0 * D, 0, sizeof(float) * nrows * D, st) != hipSuccess)
        return check_launch("dstl_bwd memset");
    hipLaunchKernelGGL(dstl_bwd_kernel, dim3((unsigned)cdiv(nrows, 4), 2, (unsigned)nslice), dim3(256), 0, st, v, t, (int)G, (int)D,
                       tv, lab, lse_tv, gscale / G, dv, dt, (int)row0, (int)(row0 + nrows), slice);
    return check_launch("dstl_bwd");
}

extern "C" int leccr_cvpool_fwd(const void* x, float* pooled, float* inv_norm, int64_t b, int64_t L, int64_t d, int dtype,
                                void* stream) {
    LECCR_REQUIRE(x && pooled && inv_norm && L > 0 && d > 0 && (d & 3) == 0, LECCR_EINVAL, "cvpool_fwd: bad argument (d % 4)");
    if (b == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    dim3 grid((unsigned)cdiv(d, 256), (unsigned)b);
    if (dtype == LECCR_F32) hipLaunchKernelGGL((cvpool_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)x, pooled, inv_norm, (int)L, (int)d);
    else if (dtype == LECCR_BF16) hipLaunchKernelGGL((cvpool_fwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, pooled, inv_norm, (int)L, (int)d);
    else { set_error("cvpool_fwd: bad dtype"); return LECCR_EINVAL; }
    return check_launch("cvpool_fwd");
}

extern "C" int leccr_cvpool_bwd(const void* x, const float* inv_norm, const float* dpooled, void* dx, int64_t b, int64_t L,
                                int64_t d, int dtype, void* stream) {
    LECCR_REQUIRE(x && inv_norm && dpooled && dx && L > 0 && d > 0 && (d & 3) == 0, LECCR_EINVAL, "cvpool_bwd: bad argument (d % 4)");
    if (b == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    dim3 grid((unsigned)cdiv(d, 256), (unsigned)b);
    if (dtype == LECCR_F32) hipLaunchKernelGGL((cvpool_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)x, inv_norm, dpooled, (float*)dx, (int)L, (int)d);
    else if (dtype == LECCR_BF16) hipLaunchKernelGGL((cvpool_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, inv_norm, dpooled, (bf16_t*)dx, (int)L, (int)d);
    else { set_error("cvpool_bwd: bad dtype"); return LECCR_EINVAL; }
    return check_launch("cvpool_bwd");
}

extern "C" int leccr_capreg_fwd(const void* c, float* loss, int64_t b, int64_t n, int64_t d, int dtype, void* stream) {
    LECCR_REQUIRE(c && loss && b > 0 && n > 0 && n <= 8 && d > 0, LECCR_EINVAL, "capreg_fwd: bad argument");
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) return check_launch("capreg memset");
    dim3 grid((unsigned)cdiv(b, 4));
    if (dtype == LECCR_F32) hipLaunchKernelGGL((capreg_kernel<float, false>), grid, dim3(256), 0, st, (const float*)c, loss, (float*)nullptr, 1.f, (int)b, (int)n, (int)d);
    else if (dtype == LECCR_BF16) hipLaunchKernelGGL((capreg_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)c, loss, (bf16_t*)nullptr, 1.f, (int)b, (int)n, (int)d);
    else { set_error("capreg_fwd: bad dtype"); return LECCR_EINVAL; }
    return check_launch("capreg_fwd");
}

extern "C" int leccr_capreg_bwd(const void* c, float gscale, void* dc, int64_t b, int64_t n, int64_t d, int dtype, void* stream) {
    LECCR_REQUIRE(c && dc && b > 0 && n > 0 && n <= 8 && d > 0, LECCR_EINVAL, "capreg_bwd: bad argument");
    hipStream_t st = as_stream(stream);
    dim3 grid((unsigned)cdiv(b, 4));
    if (dtype == LECCR_F32) hipLaunchKernelGGL((capreg_kernel<float, true>), grid, dim3(256), 0, st, (const float*)c, (float*)nullptr, (float*)dc, gscale, (int)b, (int)n, (int)d);
    else if (dtype == LECCR_BF16) hipLaunchKernelGGL((capreg_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)c, (float*)nullptr, (bf16_t*)dc, gscale, (int)b, (int)n, (int)d);
    else { set_error("capreg_bwd: bad dtype"); return LECCR_EINVAL; }
    return check_launch("capreg_bwd");
}
