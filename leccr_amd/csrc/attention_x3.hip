// attention_x3.hip - fp32 flash attention on the BF16 matrix cores: every operand as a bf16 (hi, lo) pair, three products per product.
//
// The reference-precision path (clip/model.py:185-192 `nn.MultiheadAttention` in fp32, LECCR/models/xvlm.py:88; HF BertModel's
// attention; models/attention.py:27,47,90,107), head size 64.  attention_f32.hip computes it on v_mfma_f32_32x32x2_f32 - exact products,
// 2048 matrix-pipe cycles per 32 x 32 x 64 product.  Here x = hi + lo (hi = bf16(x), lo = bf16(x - hi), 2^-18 relative left over) and
//        a b  ~  hi_a hi_b + hi_a lo_b + lo_a hi_b                      (2^-17 relative per term, f32 accumulation)
// on v_mfma_f32_32x32x16_bf16: 12 MFMAs = 384 cycles per product, 5.3 x fewer.  Q / K / V / dO are split when they are staged (f32 from
// HBM -> registers -> two bf16 planes in LDS, or -> register fragments), P and dS are split in registers right where they are computed -
// nothing is rounded to a single bf16 anywhere (a bf16 P alone would put 2^-9 on every output: the bf16 kernels' noise floor).
//
// Same formulation as attention_f32.hip ("lane = row of the softmax", score tile transposed for the query-owning kernels, so that the
// probabilities are the next product's B operand as they stand), with the operand shapes of the 16-deep instruction:
//   * a row-read fragment = 8 consecutive bf16 of the lane's row (k-step t, lane half h: columns 16 t + 8 h ..): one ds_read_b128;
//   * an accumulator tile as B operand: registers 8 u .. 8 u + 7 of lane (column, half h) are the k-step-u operand, rows
//     16 u + 8 (j >> 2) + 4 h + (j & 3);
//   * the matching A operand (the transpose of a row-major tile: V^T, K^T, dO^T, Q^T) comes out of ds_read_b64_tr_b16: two transposing
//     reads of 4 rows each - rows 16 u + 4 h .. + 3 and the four rows 8 further down - exactly that row order.
// Backward = dQ kernel (+ delta) and dK / dV kernel as in attention_f32.hip: 3 + 4 products of 12 MFMAs.
//
// LDS: per operand tile two planes (hi, lo) of 32 rows x 64 bf16, 144-byte rows (ds_read_b128 phases conflict-free; the transposing
// reads take a 2-way conflict on half their rows - LDS time is ~1/6 of the matrix-pipe time of a tile either way).
#include "attention.h"
#include <stdlib.h>

namespace leccr {
namespace x3a {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr int RSB = 72;                      // bf16 per LDS row (64 + 8 pad = 144 bytes)
constexpr int PLANE = 32 * RSB;
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

__device__ __forceinline__ void split1(float x, bf16_t& h, bf16_t& l) {
    h = static_cast<bf16_t>(x);
    l = static_cast<bf16_t>(x - static_cast<float>(h));
}

// the lane's row as register fragments: k-step t = columns 16 t + 8 h .. + 7 of the row (p points at column 8 h), scaled, split
__device__ __forceinline__ void load_frags(const float* __restrict__ p, float mul, bf16x8 (&fh)[4], bf16x8 (&fl)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + 16 * t), b = *reinterpret_cast<const f32x4*>(p + 16 * t + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16_t h, l;
            split1(a[e] * mul, h, l); fh[t][e] = h; fl[t][e] = l;
            split1(b[e] * mul, h, l); fh[t][4 + e] = h; fl[t][4 + e] = l;
        }
    }
}

// registers i0 .. i0 + 7 of an f32 array as a (hi, lo) fragment pair
__device__ __forceinline__ void split_frag(const float* v, bf16x8& fh, bf16x8& fl) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bf16_t h, l;
        split1(v[j], h, l);
        fh[j] = h; fl[j] = l;
    }
}

// one 32 x 64 f32 tile in flight: thread t holds chunks e = t and t + 256 (row e >> 4, columns 4 (e & 15) ..)
struct TileRegs { f32x4 v[2]; };

__device__ __forceinline__ TileRegs tile_load(const float* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows) {
    TileRegs t;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = (int)threadIdx.x + 256 * i;
        int64_t r = row0 + (e >> 4);
        r = r < nrows ? r : nrows - 1;              // rows past the end are clamped; their scores are masked
        t.v[i] = *reinterpret_cast<const f32x4*>(base + r * ld + 4 * (e & 15));
    }
    return t;
}

// .. split into the tile's two planes (8-byte LDS stores)
__device__ __forceinline__ void tile_store(bf16_t* __restrict__ planes, const TileRegs& t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = (int)threadIdx.x + 256 * i;
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16_t hh, ll;
            split1(t.v[i][j], hh, ll);
            h[j] = hh; l[j] = ll;
        }
        bf16_t* d = planes + (e >> 4) * RSB + 4 * (e & 15);
        *reinterpret_cast<bf16x4*>(d) = h;
        *reinterpret_cast<bf16x4*>(d + PLANE) = l;
    }
}

// A fragment by rows: row lane & 31 of the plane, k-step t
__device__ __forceinline__ bf16x8 row_frag(const bf16_t* plane, int l32, int h, int t) {
    return *reinterpret_cast<const bf16x8*>(plane + l32 * RSB + 16 * t + 8 * h);
}

// A fragment of the plane's TRANSPOSE: column 32 c + (lane & 31), rows 16 u + 4 h .. + 3 and the four rows 8 further down.
// Lane 16 g + 4 q + pp of a ds_read_b64_tr_b16 addresses row q of its 4-row block, columns 16 (g & 1) + 4 pp .. + 3, and receives the
// block's column (lane & 15) (attention_mfma.hip, gemm.hip use the same instruction).
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* plane, int lane, int c, int u) {
    const int h = lane >> 5, g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const bf16_t* p0 = plane + (16 * u + 4 * h + q) * RSB + 32 * c + 16 * (g & 1) + 4 * pp;
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(p0));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(p0 + 8 * RSB));
    const short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// acc += A B with A = (ah, al) from LDS and B = (bh, bl) in registers: hi hi into `main`, the two cross terms into `cross` (which may be
// `main` itself: f32 accumulation loses nothing by taking the small terms one at a time)
#define MMA3(ah, al, bh, bl, main, cross) \
    do {                                  \
        main = MFMA16(ah, bh, main);      \
        cross = MFMA16(ah, bl, cross);    \
        cross = MFMA16(al, bh, cross);    \
    } while (0)

__device__ __forceinline__ int slot_row(int half, int r) { return 8 * (r >> 2) + 4 * half + (r & 3); }

// Operand copies written / read by the kernels themselves (round 4; leccr_attn_fwd_s3 / leccr_attn_bwd_s3).  The fp32 path's Linears read their
// f32 operands as bf16 (hi, lo) copies in leccr_split3's layout - [rows, 3 seg], hi | hi | lo (form 0, activations) or hi | lo | hi (form 1,
// gradients) - and o / dq / dk / dv are read by nothing else: the epilogues below write those copies from their registers (bit for bit what
// leccr_split3 makes of the f32 tensor) and the f32 tensors are not written; the dQ kernel's delta = dO . O reads O back as hi + lo.
// Every pointer is at column 0 of the tensor inside segment 0; a null pointer = the f32 tensor as before.
struct S3 {
    bf16_t* o3;            // forward: X-form copy of o
    const bf16_t* o3_in;   // backward: that copy, in place of o
    bf16_t* dq3;           // backward: G-form copies
    bf16_t* dk3;
    bf16_t* dv3;
    int64_t ld3, seg;      // row stride and segment width of the output copies (elements)
    float* bq;             // optional [heads * 64] f32 each: += the column sums of dq / dk / dv (the in-projection's bias gradient), summed
    float* bk;             //   from the staged tiles (hi + lo) - one atomic per column and wave tile
    float* bv;
    int64_t ldo3, sego3;   // .. and of o3_in
};

// The wave's 32 x 64 output tile (lane (row l32, half) holds columns 4 half + 8 g + e in a0[4 g + e] and 32 + the same in a1) -> the three
// segments of its operand copy.  Straight from the registers a lane would write 24 eight-byte pieces of its own row (measured: the three
// kernels 435 us slower per layer than with f32 outputs - most of what the removed split passes save); through a wave-private LDS tile
// (two planes of 32 rows x RSB) the wave writes 16-byte pieces, eight lanes per 128-byte row.  `stage`: 2 * PLANE bf16 nobody else touches
// any more (the kernel's operand buffers after its last barrier); `base`: the copy at the wave's first row, the tile's column 0, segment 0.
template <int FORM>
__device__ __forceinline__ void store_tile_s3(bf16_t* __restrict__ stage, bf16_t* __restrict__ base, int64_t ld3, int64_t seg, int rows_valid,
                                              int lane, const float (&a0)[16], const float (&a1)[16], float* __restrict__ colsum = nullptr) {
    const int l32 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bf16x4 h0, l0, h1, l1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // the split of the ROUNDED f32 value, as leccr_split3 would read it from memory (without the fence hipcc contracts the caller's
            // `acc * inv - hi` into one fma of the unrounded product and the lo halves differ from leccr_split3's in the last bit)
            float x = a0[4 * g + e], y = a1[4 * g + e];
            asm volatile("" : "+v"(x), "+v"(y));
            bf16_t hh, ll;
            split1(x, hh, ll); h0[e] = hh; l0[e] = ll;
            split1(y, hh, ll); h1[e] = hh; l1[e] = ll;
        }
        bf16_t* d = stage + l32 * RSB + 4 * half + 8 * g;
        *reinterpret_cast<bf16x4*>(d) = h0;
        *reinterpret_cast<bf16x4*>(d + 32) = h1;
        *reinterpret_cast<bf16x4*>(d + PLANE) = l0;
        *reinterpret_cast<bf16x4*>(d + PLANE + 32) = l1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave's own LDS writes have landed (in-order LDS, one wave)
    if (colsum) {           // column `lane` of the tile, its valid rows, hi + lo (the values the copy's readers see): one atomic per column
        float t = 0.f;
        for (int row = 0; row < rows_valid; ++row)
            t += static_cast<float>(stage[row * RSB + lane]) + static_cast<float>(stage[PLANE + row * RSB + lane]);
        atomicAdd(colsum + lane, t);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane + 64 * j, row = c >> 3, ch = c & 7;
        const bf16x8 hv = *reinterpret_cast<const bf16x8*>(stage + row * RSB + 8 * ch);
        const bf16x8 lv = *reinterpret_cast<const bf16x8*>(stage + PLANE + row * RSB + 8 * ch);
        if (row < rows_valid) {
            bf16_t* d = base + row * ld3 + 8 * ch;
            *reinterpret_cast<bf16x8*>(d) = hv;
            *reinterpret_cast<bf16x8*>(d + seg) = FORM == 0 ? hv : lv;
            *reinterpret_cast<bf16x8*>(d + 2 * seg) = FORM == 0 ? lv : hv;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // .. and its reads have returned before the tile is staged again
}

template <int FORM>
__device__ __forceinline__ void store_split4(bf16_t* __restrict__ p, int64_t seg, const f32x4& v) {
    bf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // the split of the ROUNDED f32 value, as leccr_split3 would read it from memory: without this fence hipcc contracts the epilogue's
        // `acc * inv - hi` into one fma of the unrounded product and the lo halves differ from leccr_split3's in the last bit
        float x = v[j];
        asm volatile("" : "+v"(x));
        bf16_t hh, ll;
        split1(x, hh, ll);
        h[j] = hh; l[j] = ll;
    }
    *reinterpret_cast<bf16x4*>(p) = h;
    *reinterpret_cast<bf16x4*>(p + seg) = FORM == 0 ? h : l;
    *reinterpret_cast<bf16x4*>(p + 2 * seg) = FORM == 0 ? l : h;
}

__device__ __forceinline__ uint32_t valid_bits(const uint8_t* __restrict__ mask_row, int64_t i0, int64_t n, int l32) {
    const int64_t i = i0 + l32;
    const bool ok = i < n && (mask_row == nullptr || mask_row[i] == 0);
    return (uint32_t)__ballot(ok);
}

// ------------------------------------------------------------------------------------------------ forward
template <bool DROP, bool S3OUT>
__global__ __launch_bounds__(256, 2) void fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                  float* __restrict__ o, float* __restrict__ lse, const uint8_t* __restrict__ kmask,
                                                  AttnDims D, float scale, S3 s3) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][2 * PLANE];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[2][2 * PLANE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
    const int64_t nqb = (D.Lq + 127) / 128;
    const int64_t bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
    const int64_t bb = bh / D.heads, h = bh - bb * D.heads;
    const int64_t q0 = qb * 128 + wave * 32, qi = q0 + l32;
    const bool active = q0 < D.Lq, q_ok = qi < D.Lq;
    const int64_t qc = q_ok ? qi : D.Lq - 1;
    const int64_t row = bh * D.Lq + qi;
    const float* kbase = k + bb * D.Lk * D.ldk + h * 64;
    const float* vbase = v + bb * D.Lk * D.ldv + h * 64;
    const uint8_t* mrow = kmask ? kmask + bb * D.Lk : nullptr;

    bf16x8 qh[4], ql[4];
    load_frags(q + (bb * D.Lq + qc) * D.ldq + h * 64 + half * 8, scale * kLog2e, qh, ql);      // scores in the log2 domain
    f32x16_t o0 = zero16(), o1 = zero16();
    float m = -INFINITY, l = 0.f;

    const int nt = (int)((D.Lk + 31) / 32);
    TileRegs tk = tile_load(kbase, D.ldk, 0, D.Lk), tv = tile_load(vbase, D.ldv, 0, D.Lk);
    tile_store(Ks[0], tk);
    tile_store(Vs[0], tv);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const int64_t k0 = (int64_t)t * 32;
        if (t + 1 < nt) {
            tk = tile_load(kbase, D.ldk, k0 + 32, D.Lk);
            tv = tile_load(vbase, D.ldv, k0 + 32, D.Lk);
        }
        if (active) {
            const uint32_t vb = valid_bits(mrow, k0, D.Lk, l32) >> (4 * half);
            f32x16_t sa = zero16(), sb = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kh = row_frag(Ks[buf], l32, half, ks), kl = row_frag(Ks[buf] + PLANE, l32, half, ks);
                MMA3(kh, kl, qh[ks], ql[ks], sa, sb);
            }
            float s[16], mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = (vb >> (8 * (r >> 2) + (r & 3))) & 1u;
                s[r] = ok ? sa[r] + sb[r] : -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // The reference point m of the running sums moves only when some query's maximum has grown by more than 2^8 since it was
            // set (wave-uniform decision): p = exp2(s - m) then stays below 256, harmless in f32 accumulators and - unlike in the bf16
            // kernels, whose P is ONE bf16 - without loss here, because P is carried as a (hi, lo) pair.  Most tiles skip the 32
            // accumulator multiplies (and their AGPR round trip) altogether.
            const bool grow = mx > m + 8.f || (m == -INFINITY && mx > -INFINITY);
            if (__ballot(grow) != 0ull) {
                const float mn = fmaxf(m, mx);
                const float mu = mn == -INFINITY ? 0.f : mn;
                const float alpha = __builtin_amdgcn_exp2f(m - mu);
                m = mn;
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            }
            const float mu = m == -INFINITY ? 0.f : m;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - mu);
                ps += s[r];
            }
            l += ps;
            if constexpr (DROP) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] *= drop_mul(D.drop, (uint32_t)(row * D.Lk + k0 + slot_row(half, r)));
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 ph, pl;
                split_frag(s + 8 * u, ph, pl);
                const bf16x8 v0h = tr_frag(Vs[buf], lane, 0, u), v0l = tr_frag(Vs[buf] + PLANE, lane, 0, u);
                const bf16x8 v1h = tr_frag(Vs[buf], lane, 1, u), v1l = tr_frag(Vs[buf] + PLANE, lane, 1, u);
                MMA3(v0h, v0l, ph, pl, o0, o0);
                MMA3(v1h, v1l, ph, pl, o1, o1);
            }
        }
        if (t + 1 < nt) {
            tile_store(Ks[buf ^ 1], tk);
            tile_store(Vs[buf ^ 1], tv);
        }
        __syncthreads();
    }
    if (!active) return;
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if constexpr (S3OUT) {  // the X-form operand copy of o instead of o (see S3), the wave's 32 rows through its LDS tile
        float a0[16], a1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = o0[r] * inv; a1[r] = o1[r] * inv; }
        const int rows_valid = (int)(D.Lq - q0 < 32 ? D.Lq - q0 : 32);
        store_tile_s3<0>(wave < 2 ? &Ks[wave][0] : &Vs[wave - 2][0], s3.o3 + (bb * D.Lq + q0) * s3.ld3 + h * 64, s3.ld3, s3.seg, rows_valid, lane, a0, a1);
    } else if (q_ok) {
        float* op = o + (bb * D.Lq + qi) * D.ldo + h * 64 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<f32x4*>(op + 8 * g) = f32x4{o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(op + 32 + 8 * g) = f32x4{o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
        }
    }
    if (q_ok && half == 0) lse[row] = l > 0.f ? (m + __builtin_amdgcn_logf(l)) * kLn2 : INFINITY;
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+ delta)
template <bool DROP, bool S3OUT>
__global__ __launch_bounds__(256, 2) void dq_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                 const float* __restrict__ o, const float* __restrict__ dout, const float* __restrict__ lse,
                                                 const uint8_t* __restrict__ kmask, float* __restrict__ dq, float* __restrict__ delta,
                                                 AttnDims D, float scale, S3 s3) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][2 * PLANE];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[2][2 * PLANE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
    const int64_t nqb = (D.Lq + 127) / 128;
    const int64_t bh = blockIdx.x / nqb, qb = blockIdx.x - bh * nqb;
    const int64_t bb = bh / D.heads, h = bh - bb * D.heads;
    const int64_t q0 = qb * 128 + wave * 32, qi = q0 + l32;
    const bool active = q0 < D.Lq, q_ok = qi < D.Lq;
    const int64_t qc = q_ok ? qi : D.Lq - 1;
    const int64_t row = bh * D.Lq + qi;
    const float* kbase = k + bb * D.Lk * D.ldk + h * 64;
    const float* vbase = v + bb * D.Lk * D.ldv + h * 64;
    const uint8_t* mrow = kmask ? kmask + bb * D.Lk : nullptr;

    bf16x8 qh[4], ql[4], gh[4], gl[4];
    load_frags(q + (bb * D.Lq + qc) * D.ldq + h * 64 + half * 8, scale * kLog2e, qh, ql);
    const float* dop = dout + (bb * D.Lq + qc) * D.ldo + h * 64 + half * 8;
    load_frags(dop, 1.f, gh, gl);
    float dl = 0.f;
    if constexpr (S3OUT) { // delta_i = dO_i . O_i with O read back from its X-form copy (hi | hi | lo): O = hi + lo
        const bf16_t* op3 = s3.o3_in + (bb * D.Lq + qc) * s3.ldo3 + h * 64 + half * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 oh = *reinterpret_cast<const bf16x8*>(op3 + 16 * t), ol = *reinterpret_cast<const bf16x8*>(op3 + 2 * s3.sego3 + 16 * t);
            const f32x4 c = *reinterpret_cast<const f32x4*>(dop + 16 * t), d = *reinterpret_cast<const f32x4*>(dop + 16 * t + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dl = fmaf((float)oh[e] + (float)ol[e], c[e], fmaf((float)oh[4 + e] + (float)ol[4 + e], d[e], dl));
        }
    } else {              // .. in f32 (this lane: its half's columns of every k-step)
        const float* op = o + (bb * D.Lq + qc) * D.ldo + h * 64 + half * 8;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(op + 16 * t), b = *reinterpret_cast<const f32x4*>(op + 16 * t + 4);
            const f32x4 c = *reinterpret_cast<const f32x4*>(dop + 16 * t), d = *reinterpret_cast<const f32x4*>(dop + 16 * t + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dl = fmaf(a[e], c[e], fmaf(b[e], d[e], dl));
        }
    }
    dl += __shfl_xor(dl, 32, 64);
    if (q_ok && half == 0) delta[row] = dl;
    const float lse2 = lse[bh * D.Lq + qc] * kLog2e;
    f32x16_t dq0 = zero16(), dq1 = zero16();

    const int nt = (int)((D.Lk + 31) / 32);
    TileRegs tk = tile_load(kbase, D.ldk, 0, D.Lk), tv = tile_load(vbase, D.ldv, 0, D.Lk);
    tile_store(Ks[0], tk);
    tile_store(Vs[0], tv);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const int64_t k0 = (int64_t)t * 32;
        if (t + 1 < nt) {
            tk = tile_load(kbase, D.ldk, k0 + 32, D.Lk);
            tv = tile_load(vbase, D.ldv, k0 + 32, D.Lk);
        }
        if (active) {
            const uint32_t vb = valid_bits(mrow, k0, D.Lk, l32) >> (4 * half);
            f32x16_t s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {         // S^T = K Q^T and dP^T = V dO^T (two chains, interleaved)
                const bf16x8 kh = row_frag(Ks[buf], l32, half, ks), kl = row_frag(Ks[buf] + PLANE, l32, half, ks);
                const bf16x8 vh = row_frag(Vs[buf], l32, half, ks), vl = row_frag(Vs[buf] + PLANE, l32, half, ks);
                s = MFMA16(kh, qh[ks], s);
                dp = MFMA16(vh, gh[ks], dp);
                s = MFMA16(kh, ql[ks], s);
                dp = MFMA16(vh, gl[ks], dp);
                s = MFMA16(kl, qh[ks], s);
                dp = MFMA16(vl, gh[ks], dp);
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool ok = (vb >> (8 * (r >> 2) + (r & 3))) & 1u;
                const float p = ok ? __builtin_amdgcn_exp2f(s[r] - lse2) : 0.f;
                float dpr = dp[r];
                if constexpr (DROP) dpr *= drop_mul(D.drop, (uint32_t)(row * D.Lk + k0 + slot_row(half, r)));
                ds[r] = p * (dpr - dl);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {            // dQ^T += K^T dS^T
                bf16x8 dh, dlo;
                split_frag(ds + 8 * u, dh, dlo);
                const bf16x8 k0h = tr_frag(Ks[buf], lane, 0, u), k0l = tr_frag(Ks[buf] + PLANE, lane, 0, u);
                const bf16x8 k1h = tr_frag(Ks[buf], lane, 1, u), k1l = tr_frag(Ks[buf] + PLANE, lane, 1, u);
                MMA3(k0h, k0l, dh, dlo, dq0, dq0);
                MMA3(k1h, k1l, dh, dlo, dq1, dq1);
            }
        }
        if (t + 1 < nt) {
            tile_store(Ks[buf ^ 1], tk);
            tile_store(Vs[buf ^ 1], tv);
        }
        __syncthreads();
    }
    if (!active) return;
    if constexpr (S3OUT) {     // the G-form operand copy of dq instead of dq (see S3)
        float a0[16], a1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = dq0[r] * scale; a1[r] = dq1[r] * scale; }
        const int rows_valid = (int)(D.Lq - q0 < 32 ? D.Lq - q0 : 32);
        store_tile_s3<1>(wave < 2 ? &Ks[wave][0] : &Vs[wave - 2][0], s3.dq3 + (bb * D.Lq + q0) * s3.ld3 + h * 64, s3.ld3, s3.seg, rows_valid, lane, a0, a1,
                         s3.bq ? s3.bq + h * 64 : nullptr);
        return;
    }
    if (!q_ok) return;
    float* dp_ = dq + (bb * D.Lq + qi) * D.ldq + h * 64 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = dq0[4 * g + e] * scale;
            b[e] = dq1[4 * g + e] * scale;
        }
        *reinterpret_cast<f32x4*>(dp_ + 8 * g) = a;
        *reinterpret_cast<f32x4*>(dp_ + 32 + 8 * g) = b;
    }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <bool DROP, bool S3OUT>
__global__ __launch_bounds__(256, 2) void dkv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                  const float* __restrict__ dout, const float* __restrict__ lse,
                                                  const float* __restrict__ delta, const uint8_t* __restrict__ kmask,
                                                  float* __restrict__ dk, float* __restrict__ dv, AttnDims D, float scale, S3 s3) {
    __shared__ __attribute__((aligned(16))) bf16_t Qs[2][2 * PLANE];
    __shared__ __attribute__((aligned(16))) bf16_t Os[2][2 * PLANE];       // dO tiles
    __shared__ __attribute__((aligned(16))) float Ls[2][32], Ds[2][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l32 = lane & 31;
    const int64_t nkb = (D.Lk + 127) / 128;
    const int64_t bh = blockIdx.x / nkb, kb = blockIdx.x - bh * nkb;
    const int64_t bb = bh / D.heads, h = bh - bb * D.heads;
    const int64_t k0w = kb * 128 + wave * 32, ki = k0w + l32;
    const bool active = k0w < D.Lk, k_in = ki < D.Lk;
    const int64_t kc = k_in ? ki : D.Lk - 1;
    const bool k_ok = k_in && !(kmask && kmask[bb * D.Lk + kc]);
    const float* qbase = q + bb * D.Lq * D.ldq + h * 64;
    const float* obase = dout + bb * D.Lq * D.ldo + h * 64;
    const float* lbase = lse + bh * D.Lq;
    const float* dbase = delta + bh * D.Lq;

    bf16x8 kh[4], kl[4], vh[4], vl[4];
    load_frags(k + (bb * D.Lk + kc) * D.ldk + h * 64 + half * 8, scale * kLog2e, kh, kl);
    load_frags(v + (bb * D.Lk + kc) * D.ldv + h * 64 + half * 8, 1.f, vh, vl);
    f32x16_t dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();

    const int nt = (int)((D.Lq + 31) / 32);
    auto side_load = [&](int64_t i0, float& lv, float& dv_) {           // (threads 0-31) lse and delta of query i0 + l32
        const int64_t i = i0 + l32 < D.Lq ? i0 + l32 : D.Lq - 1;
        lv = lbase[i] * kLog2e;
        dv_ = dbase[i];
    };
    TileRegs tq = tile_load(qbase, D.ldq, 0, D.Lq), to = tile_load(obase, D.ldo, 0, D.Lq);
    float lv = 0.f, dvv = 0.f;
    if (threadIdx.x < 32) side_load(0, lv, dvv);
    tile_store(Qs[0], tq);
    tile_store(Os[0], to);
    if (threadIdx.x < 32) { Ls[0][l32] = lv; Ds[0][l32] = dvv; }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const int64_t i0 = (int64_t)t * 32;
        if (t + 1 < nt) {
            tq = tile_load(qbase, D.ldq, i0 + 32, D.Lq);
            to = tile_load(obase, D.ldo, i0 + 32, D.Lq);
            if (threadIdx.x < 32) side_load(i0 + 32, lv, dvv);
        }
        if (active) {
            f32x16_t s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {         // S = Q K^T and dP = dO V^T (lane = key, registers = queries)
                const bf16x8 ah = row_frag(Qs[buf], l32, half, ks), al = row_frag(Qs[buf] + PLANE, l32, half, ks);
                const bf16x8 bh_ = row_frag(Os[buf], l32, half, ks), bl = row_frag(Os[buf] + PLANE, l32, half, ks);
                s = MFMA16(ah, kh[ks], s);
                dp = MFMA16(bh_, vh[ks], dp);
                s = MFMA16(ah, kl[ks], s);
                dp = MFMA16(bh_, vl[ks], dp);
                s = MFMA16(al, kh[ks], s);
                dp = MFMA16(bl, vh[ks], dp);
            }
            float pd[16], ds[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[buf][8 * g + 4 * half]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&Ds[buf][8 * g + 4 * half]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const int64_t qi = i0 + 8 * g + 4 * half + e;
                    const bool ok = k_ok && qi < D.Lq;
                    const float p = ok ? __builtin_amdgcn_exp2f(s[r] - l4[e]) : 0.f;
                    float dm = 1.f;
                    if constexpr (DROP) dm = drop_mul(D.drop, (uint32_t)((bh * D.Lq + qi) * D.Lk + ki));
                    pd[r] = p * dm;
                    ds[r] = p * (dm * dp[r] - d4[e]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {            // dV^T += dO^T P, dK^T += Q^T dS
                bf16x8 ph, pl, dh, dlo;
                split_frag(pd + 8 * u, ph, pl);
                split_frag(ds + 8 * u, dh, dlo);
                {
                    const bf16x8 o0h = tr_frag(Os[buf], lane, 0, u), o0l = tr_frag(Os[buf] + PLANE, lane, 0, u);
                    const bf16x8 o1h = tr_frag(Os[buf], lane, 1, u), o1l = tr_frag(Os[buf] + PLANE, lane, 1, u);
                    MMA3(o0h, o0l, ph, pl, dv0, dv0);
                    MMA3(o1h, o1l, ph, pl, dv1, dv1);
                }
                {
                    const bf16x8 q0h = tr_frag(Qs[buf], lane, 0, u), q0l = tr_frag(Qs[buf] + PLANE, lane, 0, u);
                    const bf16x8 q1h = tr_frag(Qs[buf], lane, 1, u), q1l = tr_frag(Qs[buf] + PLANE, lane, 1, u);
                    MMA3(q0h, q0l, dh, dlo, dk0, dk0);
                    MMA3(q1h, q1l, dh, dlo, dk1, dk1);
                }
            }
        }
        if (t + 1 < nt) {
            tile_store(Qs[buf ^ 1], tq);
            tile_store(Os[buf ^ 1], to);
            if (threadIdx.x < 32) { Ls[buf ^ 1][l32] = lv; Ds[buf ^ 1][l32] = dvv; }
        }
        __syncthreads();
    }
    if (!active) return;
    if constexpr (S3OUT) {     // the G-form operand copies of dk / dv instead of dk / dv (see S3): one after the other through the wave's LDS tile
        bf16_t* stage = wave < 2 ? &Qs[wave][0] : &Os[wave - 2][0];
        const int rows_valid = (int)(D.Lk - k0w < 32 ? D.Lk - k0w : 32);
        float a0[16], a1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = dk0[r] * scale; a1[r] = dk1[r] * scale; }
        store_tile_s3<1>(stage, s3.dk3 + (bb * D.Lk + k0w) * s3.ld3 + h * 64, s3.ld3, s3.seg, rows_valid, lane, a0, a1, s3.bk ? s3.bk + h * 64 : nullptr);
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = dv0[r]; a1[r] = dv1[r]; }
        store_tile_s3<1>(stage, s3.dv3 + (bb * D.Lk + k0w) * s3.ld3 + h * 64, s3.ld3, s3.seg, rows_valid, lane, a0, a1, s3.bv ? s3.bv + h * 64 : nullptr);
        return;
    }
    if (!k_in) return;
    float* dkp = dk + (bb * D.Lk + ki) * D.ldk + h * 64 + 4 * half;
    float* dvp = dv + (bb * D.Lk + ki) * D.ldv + h * 64 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 a, b, c, d;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = dk0[4 * g + e] * scale;
            b[e] = dk1[4 * g + e] * scale;
            c[e] = dv0[4 * g + e];
            d[e] = dv1[4 * g + e];
        }
        *reinterpret_cast<f32x4*>(dkp + 8 * g) = a;
        *reinterpret_cast<f32x4*>(dkp + 32 + 8 * g) = b;
        *reinterpret_cast<f32x4*>(dvp + 8 * g) = c;
        *reinterpret_cast<f32x4*>(dvp + 32 + 8 * g) = d;
    }
}

}  // namespace x3a

// LECCR_ATTN_F32=exact: the v_mfma_f32_32x32x2_f32 kernels of attention_f32.hip instead (exact products, ~3 x the time)
bool attn_x3_wanted() {
    static const bool on = [] { const char* e = getenv("LECCR_ATTN_F32"); return !(e && e[0] == 'e'); }();
    return on;
}

static x3a::S3 to_s3(const AttnS3* p) {
    x3a::S3 s{};
    if (p) {
        s.o3 = (bf16_t*)p->o3; s.o3_in = (const bf16_t*)p->o3_in;
        s.dq3 = (bf16_t*)p->dq3; s.dk3 = (bf16_t*)p->dk3; s.dv3 = (bf16_t*)p->dv3;
        s.ld3 = p->ld3; s.seg = p->seg; s.ldo3 = p->ldo3; s.sego3 = p->sego3;
        s.bq = p->bq; s.bk = p->bk; s.bv = p->bv;
    }
    return s;
}

int attn_x3_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask, const AttnDims& D, hipStream_t st,
                const AttnS3* s3p) {
    const float scale = 1.f / sqrtf((float)D.hd);
    const dim3 grid((unsigned)(D.b * D.heads * ((D.Lq + 127) / 128)));
    const x3a::S3 s3 = to_s3(s3p);
    const bool out3 = s3.o3 != nullptr;
#define FWD(DROP_, S3_) hipLaunchKernelGGL((x3a::fwd_kernel<DROP_, S3_>), grid, dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (float*)o, lse, kmask, D, scale, s3)
    if (D.dropout) { if (out3) FWD(true, true); else FWD(true, false); }
    else { if (out3) FWD(false, true); else FWD(false, false); }
#undef FWD
    return check_launch("attn_x3_fwd");
}

int attn_x3_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, const uint8_t* kmask,
                void* dq, void* dk, void* dv, float* delta, const AttnDims& D, hipStream_t st, const AttnS3* s3p) {
    const float scale = 1.f / sqrtf((float)D.hd);
    const dim3 gq((unsigned)(D.b * D.heads * ((D.Lq + 127) / 128))), gk((unsigned)(D.b * D.heads * ((D.Lk + 127) / 128)));
    const x3a::S3 s3 = to_s3(s3p);
#define ARGS_Q (const float*)q, (const float*)k, (const float*)v, (const float*)o, (const float*)dout, lse, kmask, (float*)dq, delta, D, scale, s3
#define ARGS_K (const float*)q, (const float*)k, (const float*)v, (const float*)dout, lse, (const float*)delta, kmask, (float*)dk, (float*)dv, D, scale, s3
    const bool out3 = s3.dq3 != nullptr;         // (leccr_attn_bwd_s3: the o copy in, the three gradient copies out - all or none)
#define BWD(DROP_, S3_)                                                                      \
    do {                                                                                     \
        hipLaunchKernelGGL((x3a::dq_kernel<DROP_, S3_>), gq, dim3(256), 0, st, ARGS_Q);      \
        hipLaunchKernelGGL((x3a::dkv_kernel<DROP_, S3_>), gk, dim3(256), 0, st, ARGS_K);     \
    } while (0)
    if (D.dropout) { if (out3) BWD(true, true); else BWD(true, false); }
    else { if (out3) BWD(false, true); else BWD(false, false); }
#undef BWD
#undef ARGS_Q
#undef ARGS_K
    return check_launch("attn_x3_bwd");
}

}  // namespace leccr
