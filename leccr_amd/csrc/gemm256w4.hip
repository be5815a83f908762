// gemm256w4.hip - the 256x256x64 bf16 GEMM tile run by FOUR waves (one per SIMD), forward / dX products (NT, bf16 out).
//
// Why a second structure (measured with scripts/mb/ktile_mix.hip, PFLOP/s-equivalent of the K-tile's instruction mix):
// 8 waves x (128x64 wave tile, 16x16x32 MFMAs): MFMA + reads + DMA + barriers = 1.26-1.42;  4 waves x (128x128 wave
// tile, 32x32x16 MFMAs, accumulators in AGPRs) with ONE barrier per K-tile = 1.57.  Fewer, larger MFMAs per byte read
// from LDS (32 fragment reads per 64 MFMAs of 32 cycles, against 24 per 64 of 16), no second wave contending for the SIMD's
// issue port, and the whole 512-entry register file for one wave: 256 accumulator registers + double-buffered fragments.
//
// Structure
//   * 256 threads = 4 waves = 2 (M) x 2 (N); wave tile 128 x 128 = 4 x 4 accumulators of v_mfma_f32_32x32x16_bf16, issued
//     with swapped roles (the B rows are the A operand) so that a lane ends with 4 consecutive N of one row:
//     D register i of lane l = C[m = 32 mi + (l & 31)][n = 32 ni + 8 (i >> 2) + 4 (l >> 5) + (i & 3)]   (probed: scripts/mb).
//   * LDS: 2 slots x {A0, A1, B0, B1} half-tiles of 16 KiB (128 rows x 64 k), filled by global_load_lds_dwordx4; rows of
//     128 bytes, physical 16-byte chunk = logical chunk ^ ((row >> 1) & 7): with a lane reading row (l & 31), k-group (l >> 5)
//     the four 16-lane groups of a ds_read_b128 then cover all 64 banks (the 8-wave kernel's (row & 7) swizzle does not:
//     rows r and r + 24 would collide).  The swizzle is applied to the SOURCE address.
//   * K-tile = 4 k-steps of 16 MFMAs.  During k-step s the 8 fragments of k-step s + 1 are read into the other fragment
//     buffer (k-step 3 reads k-step 0 of the NEXT K-tile); 16 DMA pieces per wave and K-tile go out during k-steps 3, 0, 1.
//     One barrier per K-tile, between k-steps 2 and 3: before it every wave has finished reading K-tile t (so its slot can
//     take K-tile t+2) and has seen its own pieces of K-tile t+1 land (vmcnt(0)).
//   * Persistent items, XCD-contiguous, continuous K-tile stream across items - as gemm256.hip.
//   * Epilogue: per 32-row accumulator tile and 16-row half, a wave-private LDS transpose (16 rows x 128 f32, XOR-swizzled, no
//     padding: 8 KiB per wave) so that a lane stores 16 bytes of one row; bias / activation (+ saved pre-activation) /
//     derivative-activation (reading it) / residual as in gemm256.hip.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace leccr {
#ifndef W4_BURST
#define W4_BURST 0            // bit ks: bf16 k-step ks issues its fragment reads and DMA pieces in one burst ahead of its 16 MFMAs (experiment: see profiles/r04_fp8_kstep_schedules.txt)
#endif
namespace g4 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 16384, SLOT = 4 * HALF, NTH = 256;
constexpr int STG_WAVE = 16 * 128 * 4;                      // 16 rows x 128 f32 per wave
constexpr int LDS_BYTES = 2 * SLOT + 4 * STG_WAVE;          // 131072 + 32768 = 163840 = all of the CU's LDS
constexpr int F_RES = 2, F_AUXOUT = 4;
// F_FP8: operands are OCP fp8 bytes (B e4m3; A e4m3, or e5m2 with F_BF8A): BASELINE configs[4]'s "fp8 MFMA" GEMM inputs.  A K-tile is
// 128 fp8 = the same 128-byte LDS rows as 64 bf16, so staging, swizzle and fragment read addresses are byte-identical to the bf16
// kernel (the launcher passes K / 2, lda / 2, ldb / 2 in bf16 units); only the MFMA differs - see the K loop.
constexpr int F_FP8 = 32, F_BF8A = 64;
// F_Q8: additionally write an fp8 copy of C for the product that consumes it (leccr_gemm_params.c8): saves that operand's quantise pass
constexpr int F_Q8 = 128, F_Q8_BF8 = 256;       // F_Q8_BF8: the copy is e5m2 (a gradient), else e4m3
// F_BURST1 (fp8): K1 issues its eight fragment reads and eight DMA pieces in one burst AHEAD of its 16 MFMAs instead of one behind each.
// Measured per shape on one box (profiles/r04_fp8_kstep_schedules.txt): faster for the short reductions (K = 768 / 1024: fc1 972 -> 934 us,
// out 257 -> 251, B/16 fc1 501 -> 483), slower for the long ones (K = 3072 / 4096: fc2 705 -> 729) - chosen by K at launch.
constexpr int F_BURST1 = 512;

struct Args {
    const bf16_t* A;
    const bf16_t* B;
    bf16_t* C;
    int64_t M, N, K, lda, ldb, ldc;
    const float* bias;
    const bf16_t* aux;
    int64_t ldaux;
    bf16_t* aux_out;
    const bf16_t* residual;
    int64_t ldr;
    float alpha;
    int tiles_m, tiles_n, nkt, n_items;
    const float* alpha_dev[2];           // device scalars folded into alpha at kernel start (fp8 inverse scales), or null
    uint8_t* c8;                         // F_Q8: fp8 copy of C (row stride ldc8), quantised under c8_site = {amax, scale, 1/scale, fmt}
    int64_t ldc8;
    float* c8_site;
    float* colsum;                       // F_Q8_BF8 (the dX product with a derivative epilogue): [N] += column sums of the f32 epilogue values
};                                       // F_Q8 with C == nullptr: only the fp8 copy (and aux_out) is written

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int int4_t;

template <int IMM>
__device__ __forceinline__ bf16x8 lds_b128(uint32_t addr) {
    int4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return __builtin_bit_cast(bf16x8, v);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define SEG_BARRIER()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
#define WAVE_LDS_FENCE() asm volatile("" ::: "memory")

struct Cursor {                // position of one K-tile of this block's stream (wave-uniform)
    int j, kt;
    int64_t m0, n0, k0;
    const bf16_t* p[4];        // A0, A1, B0, B1 half-tile sources (row 0 of the half, this K-tile's k0)
    int rv[4];                 // last valid row of the half (127 for interior halves)
};

__device__ __forceinline__ void decode(const Args& a, int first_item, int stride, Cursor& c) {
    const int item = first_item + c.j * stride;
    const int tm = item / a.tiles_n;
    c.m0 = (int64_t)tm * BM;
    c.n0 = (int64_t)(item - tm * a.tiles_n) * BN;
    c.k0 = (int64_t)c.kt * BK;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const bool isA = w < 2;
        const int64_t r0 = (isA ? c.m0 : c.n0) + (w & 1) * 128;
        const int64_t ld = isA ? a.lda : a.ldb, rmax = isA ? a.M : a.N;
        const bf16_t* base = isA ? a.A : a.B;
        const int64_t rb = r0 < rmax ? r0 : rmax - 1;          // a half that starts past the end re-reads the last row
        const int64_t left = rmax - 1 - rb;
        c.rv[w] = (int)(left < 127 ? left : 127);
        c.p[w] = base + rb * ld + c.k0;
    }
}

__device__ __forceinline__ void advance(const Args& a, int first_item, int stride, int n_mine, Cursor& c) {
    if (c.kt + 1 < a.nkt) {
        ++c.kt;
        c.k0 += BK;
#pragma unroll
        for (int w = 0; w < 4; ++w) c.p[w] += BK;
    } else if (c.j + 1 < n_mine) {
        ++c.j;
        c.kt = 0;
        decode(a, first_item, stride, c);
    }
}

// One DMA piece: 256 threads x 16 bytes = 32 rows of a half-tile.  WHICH: 0 = A rows 0-127, 1 = A 128-255, 2 = B 0-127,
// 3 = B 128-255; PIECE 0..3.  toff = this thread's byte offset inside piece 0 (row0 * ld * 2 + its logical chunk); interior
// halves add a wave-uniform piece offset to the base pointer, tail halves clamp the row per lane.
template <int WHICH, int PIECE>
__device__ __forceinline__ void issue_piece(const Args& a, const Cursor& c, int row0, uint32_t colb, uint32_t toffA, uint32_t toffB,
                                            uint32_t lds_slot_off, uint32_t wave_off) {
    constexpr bool isA = WHICH < 2;
    const uint32_t ld2 = (uint32_t)(isA ? a.lda : a.ldb) * 2u;
    // branch-free: the row is clamped to the half's last valid row (interior halves: rv = 127, a no-op); 24-bit multiply
    uint32_t row = (uint32_t)(row0 + 32 * PIECE);
    const uint32_t rv = (uint32_t)c.rv[WHICH];
    row = row < rv ? row : rv;
    const char* src = reinterpret_cast<const char*>(c.p[WHICH]) + (__umul24(row, ld2) + colb);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src),
                                     (__attribute__((address_space(3))) void*)(size_t)(lds_slot_off + WHICH * HALF + PIECE * 4096 + wave_off),
                                     16, 0, 0);
}

// DMA schedule: piece number (0..15 = half * 4 + piece) issued after MFMA i of k-step ks, or -1.  The pieces issued in
// k-step 3 (right after the barrier that frees the slot) belong to K-tile t+2, those in k-steps 0..2 of the NEXT iteration
// are the rest of the same K-tile; front-loaded so that the last piece has two k-steps to land before vmcnt(0).
#ifndef W4_EXP
#define W4_EXP 0          // diagnostic builds: 1 no vmcnt wait, 2 no DMA in the loop, 4 no barrier, 8 no fragment reads, 16 no epilogue at all,
                          // 32 epilogue without its global stores, 64 one K-tile per item, 128 no drain behind the epilogue (wrong results)
#endif
constexpr int N3 = 8;     // pieces issued in k-step 3
__host__ __device__ constexpr int piece_at(int ks, int i) {
    if (ks == 3) return i >= 8 ? i - 8 : -1;
    if (ks == 0) return i >= 8 ? N3 + i - 8 : -1;
    return -1;
}

template <int EPI, int FLAGS>
__global__ __launch_bounds__(NTH) void gemm256w4_kernel(Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // persistent item assignment: XCD x (= block id % 8) owns a contiguous range of items (as gemm256.hip)
    const int G = gridDim.x, bid = blockIdx.x;
    int first_item, stride, n_mine;
    if ((G & 7) == 0) {
        const int gx = G >> 3, x = bid & 7, j = bid >> 3;
        const int q = a.n_items >> 3, r = a.n_items & 7;
        const int beg = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = x < r ? q + 1 : q;
        first_item = beg + j;
        stride = gx;
        n_mine = j < cnt ? (cnt - j + gx - 1) / gx : 0;
    } else {
        const int x = bid & 7, jx = bid >> 3;
        int off = 0;
        for (int y = 0; y < x; ++y) off += (G + 7 - y) >> 3;
        first_item = off + jx;
        stride = 1;
        n_mine = 1;
    }
    if (n_mine == 0) return;
    if constexpr ((FLAGS & F_FP8) != 0) {              // alpha carries 1 / (scale_A scale_B), both kept on the device
        if (a.alpha_dev[0]) a.alpha *= *a.alpha_dev[0];
        if (a.alpha_dev[1]) a.alpha *= *a.alpha_dev[1];
    }

    const uint32_t smem_off = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t wave_off = smem_off + (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u;
    // staging: thread -> (row0 = tid >> 3 in piece 0, physical chunk tid & 7); the logical chunk is the same in all 4 pieces
    // because (row >> 1) & 7 does not change when the row grows by 32
    const int st_row0 = tid >> 3;
    const uint32_t st_colb = (uint32_t)(((tid & 7) ^ ((st_row0 >> 1) & 7)) * 16);
    const uint32_t toffA = (uint32_t)st_row0 * (uint32_t)a.lda * 2u + st_colb, toffB = (uint32_t)st_row0 * (uint32_t)a.ldb * 2u + st_colb;
#define ISSUE_PN(PN, slot) issue_piece<((PN) >> 2), ((PN) & 3)>(a, cB, st_row0, st_colb, toffA, toffB, (slot), wave_off)

    Cursor cB, cur;
    cur.j = 0; cur.kt = 0;
    decode(a, first_item, stride, cur);
    cB = cur;
    // prologue: all of K-tile 0, the first N3 pieces of K-tile 1 (what k-step 3 of "K-tile -1" would have issued)
    static_for<0, 16>([&](auto pn) { ISSUE_PN(decltype(pn)::value, 0u); });
    advance(a, first_item, stride, n_mine, cB);                     // cB -> K-tile 1
    static_for<0, N3>([&](auto pn) { ISSUE_PN(decltype(pn)::value, (uint32_t)SLOT); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N3) : "memory");
    SEG_BARRIER();

    // fragment addresses: row r = lane & 31 of a 32-row tile, k-group kg = lane >> 5; k-step s reads chunk (2 s + kg) ^ x
    const int r = lane & 31, kg = lane >> 5, x = (r >> 1) & 7;
    uint32_t la[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) la[s] = (uint32_t)(r * 128 + (((2 * s + kg) ^ x) << 4));
    const uint32_t offA = smem_off + wr * HALF, offB = smem_off + (2 + wc) * HALF;

    bf16x8 fa[2][4], fb[2][4];      // [k-step parity][32-row tile]
    // fp8: a fragment is the 32-byte operand of v_mfma_scale_f32_32x32x64_f8f6f4 = the two 16-byte reads that the bf16 path makes
    // for k-steps 2S and 2S + 1 (chunks 4S + kg and 4S + 2 + kg of the row).  Which 32 of a k-step's 64 k a lane group holds is
    // immaterial as long as A and B agree (scripts/mb/mfma_fp8_probe.hip), so nothing but the MFMA changes.  One such MFMA takes
    // the cycles of two bf16 32x32x16 at four times their k: the K-tile's 2048 MFMA cycles now cover 128 k instead of 64.
    typedef __attribute__((ext_vector_type(8))) int v8i_t;
    [[maybe_unused]] v8i_t xa[2][4], xb[2][4];      // [k-step parity][32-row tile]
    auto frag8 = [](const bf16x8& lo, const bf16x8& hi) {
        const int4_t l = __builtin_bit_cast(int4_t, lo), h = __builtin_bit_cast(int4_t, hi);
        return v8i_t{l[0], l[1], l[2], l[3], h[0], h[1], h[2], h[3]};
    };
#define RD_FRAGS8(buf, S, KS8, i)                                                                                                   \
    do {                                                                                                                            \
        if ((i) < 4) xa[buf][(i) & 3] = frag8(lds_b128<((i) & 3) * 4096>(la[2 * (KS8)] + offA + (S)), lds_b128<((i) & 3) * 4096>(la[2 * (KS8) + 1] + offA + (S))); \
        else xb[buf][(i) & 3] = frag8(lds_b128<((i) & 3) * 4096>(la[2 * (KS8)] + offB + (S)), lds_b128<((i) & 3) * 4096>(la[2 * (KS8) + 1] + offB + (S)));         \
    } while (0)
#define RD_FRAGS(buf, S, KS, i)                                                              \
    do {                                                                                       \
        if ((i) < 4) fa[buf][(i) & 3] = lds_b128<((i) & 3) * 4096>(la[KS] + offA + (S));       \
        else fb[buf][(i) & 3] = lds_b128<((i) & 3) * 4096>(la[KS] + offB + (S));               \
    } while (0)
    // k-step 0 of K-tile 0
    if constexpr ((FLAGS & F_FP8) != 0) {
        static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; RD_FRAGS8(0, 0u, 0, i); });
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[0][2]), "+v"(xa[0][3]), "+v"(xb[0][0]), "+v"(xb[0][1]), "+v"(xb[0][2]), "+v"(xb[0][3]));
    } else {
        static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; RD_FRAGS(0, 0u, 0, i); });
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]));
    }
    __builtin_amdgcn_sched_barrier(0);

    [[maybe_unused]] float q8_scale = 1.f, q8_amax = 0.f;
    if constexpr ((FLAGS & F_Q8) != 0) q8_scale = a.c8_site[1];
    // F_Q8_BF8: the bias gradient colsum(C) from the f32 epilogue values, per lane its 8 columns, carried ACROSS the items of this
    // workgroup while they stay in the same tile column (they do whenever tiles_n divides the item stride: 4096 / 256 = 16 | 32) and
    // added to memory once per change of column: one atomic per column and workgroup instead of one per column and 128-row wave tile
    // (11 680 tiles x 2 waves on 4096 addresses serialised at the memory side: +1.0 ms on a 1.3 ms launch)
    [[maybe_unused]] float cs[8];
    [[maybe_unused]] int64_t cs_n0 = -1;
    if constexpr ((FLAGS & F_Q8_BF8) != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) cs[i] = 0.f;
    }
    [[maybe_unused]] auto cs_flush = [&]() {
        if (cs_n0 >= 0 && a.colsum != nullptr) {     // lanes with equal lane & 15 hold the same 8 columns: fold the 4 of them, lanes 0..15 add
            int lane_f = lane;
            asm volatile("" : "+v"(lane_f));
            const int64_t ncol = cs_n0 + wc * 128 + (lane_f & 15) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = cs[i];
                v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                if (lane_f < 16 && ncol < a.N) atomicAdd(a.colsum + ncol + i, v);
                cs[i] = 0.f;
            }
        }
    };
    int t = 0;
    for (int j = 0; j < n_mine; ++j) {
        f32x16 acc[16];                      // [ni * 4 + mi]
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

        for (int kt = 0; kt < ((W4_EXP & 64) ? 1 : a.nkt); ++kt, ++t) {
            const uint32_t sc = (t & 1) * SLOT, sn = SLOT - sc;
            // per-lane address pieces go through an opaque asm each K-tile so that the derived addresses are formed at their use
            asm volatile("" : "+v"(la[0]), "+v"(la[1]), "+v"(la[2]), "+v"(la[3]));
            if constexpr ((FLAGS & F_FP8) != 0) {
                // fp8: two k-steps of 64 k per K-tile, 16 scaled MFMAs each; fill(i) behind MFMA i loads one whole fragment (i < 8) or
                // issues one DMA piece (i >= 8), as in the bf16 schedule with its k-steps 0..2 folded into one:
                //   K0 (buffer 0): reads k-step 1 of this slot; the rest (pieces 8-15) of K-tile t+1  -> wait; vmcnt(0); BARRIER
                //   K1 (buffer 1): reads k-step 0 of K-tile t+1; pieces 0-7 of K-tile t+2 into the slot the barrier freed
                constexpr int AFMT = (FLAGS & F_BF8A) != 0 ? 1 : 0;      // format of the activation / gradient operand: 0 e4m3, 1 e5m2
                // the prefetch position after K0 (K-tile t+2) is worked out HERE, ahead of the K-tile's MFMAs: the branches of advance()
                // between K0 and K1 ended K0's basic block, and hipcc sank its 16 MFMAs below them - behind all of K0's fragment reads and
                // DMA issues instead of between them (round 4, seen in the listing)
                Cursor cN = cB;
                advance(a, first_item, stride, n_mine, cN);
                auto kstep8 = [&](auto ks_tag, auto&& fill) {
                    constexpr int cb = decltype(ks_tag)::value & 1;
                    constexpr bool BURST = decltype(ks_tag)::value == 1 && (FLAGS & F_BURST1) != 0;      // the k-step's reads / DMA issues ahead of its MFMAs
                    if constexpr (BURST) {
                        static_for<0, 16>([&](auto ic) { fill(ic); });
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    static_for<0, 16>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, ni = i >> 2, mi = i & 3;
                        acc[ni * 4 + mi] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xb[cb][ni], xa[cb][mi], acc[ni * 4 + mi], 0, AFMT, 0,
                                                                                           0x7F7F7F7F, 0, 0x7F7F7F7F);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (!BURST) fill(ic);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                };
#define WAIT_FRAGS8(buf)                                                                                                    \
    do {                                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                     : "+v"(xa[buf][0]), "+v"(xa[buf][1]), "+v"(xa[buf][2]), "+v"(xa[buf][3]), "+v"(xb[buf][0]),            \
                       "+v"(xb[buf][1]), "+v"(xb[buf][2]), "+v"(xb[buf][3]));                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
    } while (0)
#define FILL8(KSP, buf, S, RKS8, slot)                                                        \
    [&](auto ic) {                                                                            \
        constexpr int i = decltype(ic)::value;                                                \
        if constexpr (i < 8) RD_FRAGS8(buf, S, RKS8, i);                                      \
        if constexpr (piece_at(KSP, i) >= 0) ISSUE_PN(piece_at(KSP, i), slot);                \
    }
                kstep8(std::integral_constant<int, 0>{}, FILL8(0, 1, sc, 1, sn));
                cB = cN;                                              // cB -> K-tile t+2
                WAIT_FRAGS8(1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of K-tile t+1 have landed
                SEG_BARRIER();                                        // slot sc is free, slot sn is complete
                kstep8(std::integral_constant<int, 1>{}, FILL8(3, 0, sn, 0, sc));
                WAIT_FRAGS8(0);
                continue;
            }
            // k-step KS: 16 MFMAs on buffer KS & 1; fill(i) after MFMA i
            auto kstep = [&](auto ks_tag, auto&& fill) {
                constexpr int KS = decltype(ks_tag)::value, cb = KS & 1;
                constexpr bool BURST = ((W4_BURST >> KS) & 1) != 0;      // (experiment) the k-step's reads / DMA issues in one burst ahead of its MFMAs
                if constexpr (BURST) {
                    static_for<0, 16>([&](auto ic) { fill(ic); });
                    __builtin_amdgcn_sched_barrier(0);
                }
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, ni = i >> 2, mi = i & 3;
                    acc[ni * 4 + mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cb][ni], fa[cb][mi], acc[ni * 4 + mi], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!BURST) fill(ic);
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
#define WAIT_FRAGS(buf)                                                                                                     \
    do {                                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                     : "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fa[buf][2]), "+v"(fa[buf][3]), "+v"(fb[buf][0]),            \
                       "+v"(fb[buf][1]), "+v"(fb[buf][2]), "+v"(fb[buf][3]));                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
    } while (0)
#define FILL(KS, buf, S, RKS, slot)                                                          \
    [&](auto ic) {                                                                            \
        constexpr int i = decltype(ic)::value;                                                \
        if constexpr (i < 8 && !(W4_EXP & 8)) RD_FRAGS(buf, S, RKS, i);                        \
        if constexpr (piece_at(KS, i) >= 0 && !(W4_EXP & 2)) ISSUE_PN(piece_at(KS, i), slot);  \
    }
            // k-steps 0..2 read the next k-step of this slot and issue the rest of K-tile t+1 (into the other slot)
            kstep(std::integral_constant<int, 0>{}, FILL(0, 1, sc, 1, sn));
            WAIT_FRAGS(1);
            kstep(std::integral_constant<int, 1>{}, FILL(1, 0, sc, 2, sn));
            WAIT_FRAGS(0);
            kstep(std::integral_constant<int, 2>{}, FILL(2, 1, sc, 3, sn));
            advance(a, first_item, stride, n_mine, cB);           // cB -> K-tile t+2
            WAIT_FRAGS(1);
            if constexpr (!(W4_EXP & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of K-tile t+1 have landed
            if constexpr (!(W4_EXP & 4)) SEG_BARRIER();           // slot sc is free, slot sn is complete
            // k-step 3 reads k-step 0 of K-tile t+1 and starts K-tile t+2 into the slot just freed
            kstep(std::integral_constant<int, 3>{}, FILL(3, 0, sn, 0, sc));
            WAIT_FRAGS(0);
        }

        // ---- epilogue of this item
        {
            const int64_t m0 = cur.m0, n0 = cur.n0;
            const bool interior = m0 + BM <= a.M && n0 + BN <= a.N;
            float* stg = reinterpret_cast<float*>(smem + 2 * SLOT) + wave * (STG_WAVE / 4);
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            // write side: this lane's accumulator row m = 32 mi + (lane & 31); 16-row half hh = (lane & 31) >> 4
            const int wrow = lane_e & 15, whalf = (lane_e & 31) >> 4, wk = lane_e >> 5;
            // read side: every global access of a pass is 16 bytes per lane with 16 consecutive lanes on ONE row (256 contiguous bytes,
            // 4 rows per instruction): lane -> row 4 i + (lane >> 4) of the pass's 16 (i = 0..3), columns 8 (lane & 15) .. + 7 - fixed for the
            // lane.  (Until round 3 a lane owned 32 columns of one row and a store instruction wrote 64 separate 16-byte pieces 64 bytes
            // apart: the stores alone were 8 us of a 28 us K = 1024 item, scripts/gemm_w4_items.py with the -DW4_EXP=32 build.)
            const int rq = lane_e >> 4, rp = lane_e & 15;
            const int64_t ncol = n0 + wc * 128 + rp * 8;
            // F_Q8: the bf16 C is optional - a consumer chain that is fp8 end to end (next product AND the weight-gradient product) reads
            // only the fp8 copy, and 2 of the epilogue's 5 bytes per element stay unwritten (config E: the 4d-wide activation / its gradient)
            const bool store_c = (FLAGS & F_Q8) == 0 || a.C != nullptr;
            if constexpr ((FLAGS & F_Q8_BF8) != 0) {
                if (n0 != cs_n0) {
                    cs_flush();
                    cs_n0 = n0;
                }
            }
            auto tiles = [&](auto check_tag) {
                constexpr bool CHECK = decltype(check_tag)::value;
                const bool col_ok = !CHECK || ncol < a.N;                      // N % 32 == 0 on this path (dispatcher)
                f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
                if (a.bias) { b0 = load4(a.bias + (col_ok ? ncol : 0)); b1 = load4(a.bias + (col_ok ? ncol + 4 : 0)); }
                // derivative epilogues read the saved pre-activation (8 columns of 4 rows per lane and pass = 4 x 16 bytes): loaded one
                // pass AHEAD, so the latency sits under the previous pass's LDS transpose and stores instead of in front of every use
                // (config E, the 4d-wide dX product: 2.53 ms per launch against 1.53 for the forward twin that only stores)
                // (the residual of the plain epilogues likewise)
                constexpr bool NEED_AUX = epi_needs_aux<EPI>;
                constexpr bool PRE = NEED_AUX || (FLAGS & F_RES) != 0;
                [[maybe_unused]] bf16x8 xpre[2][4];
                auto load_pre = [&](int pass, bf16x8* dst) {
                    const bf16_t* pb = NEED_AUX ? a.aux : a.residual;
                    const int64_t pld = NEED_AUX ? a.ldaux : a.ldr;
#pragma unroll
                    for (int h8 = 0; h8 < 4; ++h8) {
                        const int64_t mm = m0 + wr * 128 + (pass >> 1) * 32 + (pass & 1) * 16 + 4 * h8 + rq;
                        const bool okp = !CHECK || (col_ok && mm < a.M);
                        dst[h8] = *reinterpret_cast<const bf16x8*>(pb + (okp ? mm * pld + ncol : 0));
                    }
                };
                if constexpr (PRE) load_pre(0, xpre[0]);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        if constexpr (PRE) {
                            if (2 * mi + hh + 1 < 8) load_pre(2 * mi + hh + 1, xpre[(2 * mi + hh + 1) & 1]);
                        }
                        if (whalf == hh) {
#pragma unroll
                            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const int c4 = ni * 8 + g * 2 + wk;                      // 16-byte chunk index of n = 32 ni + 8 g + 4 wk
                                    const f32x16& av = acc[ni * 4 + mi];
                                    *reinterpret_cast<f32x4*>(stg + wrow * 128 + ((c4 ^ wrow) << 2)) =
                                        f32x4{av[4 * g], av[4 * g + 1], av[4 * g + 2], av[4 * g + 3]};
                                }
                        }
                        WAVE_LDS_FENCE();
                        f32x4 v[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {                                  // row 4 q + rq, chunks 2 rp and 2 rp + 1
                            const int row = 4 * q + rq;
                            v[2 * q] = *reinterpret_cast<const f32x4*>(stg + row * 128 + (((2 * rp) ^ row) << 2));
                            v[2 * q + 1] = *reinterpret_cast<const f32x4*>(stg + row * 128 + (((2 * rp + 1) ^ row) << 2));
                        }
                        WAVE_LDS_FENCE();
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[2 * q] = v[2 * q] * a.alpha + b0;
                            v[2 * q + 1] = v[2 * q + 1] * a.alpha + b1;
                        }
#pragma unroll
                        for (int h8 = 0; h8 < 4; ++h8) {                              // 4 rows x (8 columns = 16 bytes of bf16)
                            f32x4 v0 = v[2 * h8], v1 = v[2 * h8 + 1];
                            const int64_t m = m0 + wr * 128 + mi * 32 + hh * 16 + 4 * h8 + rq;
                            const bool ok = !CHECK || (col_ok && m < a.M);
                            const int64_t off = ok ? m * a.ldc + ncol : 0;
                            if constexpr (epi_saves_deriv<EPI>) {      // the derivative where the pre-activation went (LECCR_EPI_MULAUX reads it back)
                                f32x4 d0, d1;
#pragma unroll
                                for (int e = 0; e < 4; ++e) { { const ActD t_ = apply_epi_d<EPI>(v0[e]); v0[e] = t_.a; d0[e] = t_.d; } { const ActD t_ = apply_epi_d<EPI>(v1[e]); v1[e] = t_.a; d1[e] = t_.d; } }
                                bf16x8 pv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) { pv[e] = (bf16_t)d0[e]; pv[4 + e] = (bf16_t)d1[e]; }
                                if (ok) *reinterpret_cast<bf16x8*>(a.aux_out + off) = pv;
                            } else if constexpr ((FLAGS & F_AUXOUT) != 0) {
                                bf16x8 pv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) { pv[e] = (bf16_t)v0[e]; pv[4 + e] = (bf16_t)v1[e]; }
                                if (ok) *reinterpret_cast<bf16x8*>(a.aux_out + off) = pv;
                            }
                            if constexpr (EPI != LECCR_EPI_NONE && !epi_saves_deriv<EPI>) {
                                f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
                                if constexpr (NEED_AUX) {
                                    const bf16x8 xv = xpre[(2 * mi + hh) & 1][h8];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { x0[e] = (float)xv[e]; x1[e] = (float)xv[4 + e]; }
                                }
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v0[e] = apply_epi<EPI>(v0[e], x0[e]); v1[e] = apply_epi<EPI>(v1[e], x1[e]); }
                            }
                            if constexpr ((FLAGS & F_RES) != 0) {
                                bf16x8 rv;
                                if constexpr (!NEED_AUX) rv = xpre[(2 * mi + hh) & 1][h8];
                                else rv = *reinterpret_cast<const bf16x8*>(a.residual + (ok ? m * a.ldr + ncol : 0));
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v0[e] += (float)rv[e]; v1[e] += (float)rv[4 + e]; }
                            }
                            bf16x8 ov;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { ov[e] = (bf16_t)v0[e]; ov[4 + e] = (bf16_t)v1[e]; }
                            if constexpr ((W4_EXP & 32) != 0) { asm volatile("" ::"v"(ov)); } else
                            if (ok && store_c) *reinterpret_cast<bf16x8*>(a.C + off) = ov;
                            if constexpr ((FLAGS & F_Q8_BF8) != 0) {
                                if (ok) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { cs[e] += v0[e]; cs[4 + e] += v1[e]; }
                                }
                            }
                            if constexpr ((FLAGS & F_Q8) != 0) {
                                constexpr float FMAX = (FLAGS & F_Q8_BF8) != 0 ? 57344.f : 448.f;
                                float q[8];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (ok) q8_amax = fmaxf(q8_amax, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
                                    q[e] = fminf(fmaxf(v0[e] * q8_scale, -FMAX), FMAX);
                                    q[4 + e] = fminf(fmaxf(v1[e] * q8_scale, -FMAX), FMAX);
                                }
                                int w0 = 0, w1 = 0;
                                if constexpr ((FLAGS & F_Q8_BF8) != 0) {
                                    w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], w0, true);
                                    w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(q[6], q[7], w1, true);
                                } else {
                                    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
                                    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
                                }
                                if (ok) *reinterpret_cast<int2*>(a.c8 + m * a.ldc8 + ncol) = int2{w0, w1};
                            }
                        }
                    }
                }
            };
            if constexpr ((W4_EXP & 16) != 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(acc[i]));
            } else {
                if (interior) tiles(std::false_type{});
                else tiles(std::true_type{});
            }
            // every ordinary load of the epilogue retired, visibly to hipcc's waitcnt pass (no vmcnt(0) sneaks into the K loop)
            if constexpr (!(W4_EXP & 128)) __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j + 1 < n_mine) {
            cur.j = j + 1;
            cur.kt = 0;
            decode(a, first_item, stride, cur);
        }
    }
    if constexpr ((FLAGS & F_Q8_BF8) != 0) cs_flush();
    if constexpr ((FLAGS & F_Q8) != 0) {               // one conditional atomic per wave and launch (see fp8.hip: block_amax)
        float m = q8_amax;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        if (lane == 0 && m > __builtin_nontemporal_load(a.c8_site))
            atomicMax(reinterpret_cast<unsigned int*>(a.c8_site), __float_as_uint(m));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // clamped tail prefetches still target this block's LDS
}

template <int EPI, int FLAGS>
static void launch_inst(const Args& a, int grid, hipStream_t st) {
    auto kfn = gemm256w4_kernel<EPI, FLAGS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NTH), LDS_BYTES, st, a);
}

}  // namespace g4

static int w4_flags(const leccr_gemm_params& p) { return (p.residual ? g4::F_RES : 0) | (p.aux_out ? g4::F_AUXOUT : 0); }

// fp8 operands (LECCR_FP8: e4m3 x e4m3; LECCR_FP8_BF8A: A e5m2 x B e4m3): whole 128-element K-tiles, 16-byte aligned rows, and the
// epilogues a forward (e4m3 activations) or a dX (e5m2 gradients) product needs
bool gemm256w4_fp8_legal(const leccr_gemm_params& p) {
    if (p.ab_dtype != LECCR_FP8 && p.ab_dtype != LECCR_FP8_BF8A) return false;
    if (p.trans_a != 0 || p.trans_b != 0 || p.c_dtype != LECCR_BF16 || p.accumulate) return false;
    if (p.M < 256 || p.N < 256 || (p.N & 31) || (p.ldc & 7) || (p.ldr & 7) || (p.ldaux & 7) || (p.K & 127) || (p.lda & 15) || (p.ldb & 15)) return false;
    if ((((uintptr_t)p.A | (uintptr_t)p.B) & 15) || p.lda >= (1 << 23) || p.ldb >= (1 << 23)) return false;
    const int f = w4_flags(p);
    if (p.ab_dtype == LECCR_FP8)
        return (p.epilogue == LECCR_EPI_NONE && (f == 0 || f == g4::F_RES)) ||
               ((p.epilogue == LECCR_EPI_QUICKGELU || p.epilogue == LECCR_EPI_GELU || p.epilogue == LECCR_EPI_QUICKGELU_D || p.epilogue == LECCR_EPI_GELU_D) &&
                f == g4::F_AUXOUT);
    return (p.epilogue == LECCR_EPI_NONE && (f == 0 || f == g4::F_RES)) ||
           ((p.epilogue == LECCR_EPI_DQUICKGELU || p.epilogue == LECCR_EPI_DGELU || p.epilogue == LECCR_EPI_MULAUX) && f == 0);
}

// the fused fp8 copy of C: activation epilogues of a forward product (e4m3 copy) or derivative epilogues of a dX product (e5m2)
bool gemm256w4_c8_legal(const leccr_gemm_params& p) {
    if (p.colsum && !(p.c8 && p.ab_dtype == LECCR_FP8_BF8A)) return false;       // fused column sums: the dX + derivative + c8 form only
    if (!p.C && !p.c8) return false;                                              // C may be omitted only beside its fp8 copy
    if (!p.c8) return true;
    if (!p.c8_site || (p.ldc8 & 7) || ((uintptr_t)p.c8 & 7) || p.ldc8 < p.N) return false;
    if (p.ab_dtype == LECCR_FP8)
        return (p.epilogue == LECCR_EPI_QUICKGELU || p.epilogue == LECCR_EPI_GELU || p.epilogue == LECCR_EPI_QUICKGELU_D || p.epilogue == LECCR_EPI_GELU_D) &&
               p.c8_fmt == 0;
    if (p.ab_dtype == LECCR_FP8_BF8A)
        return (p.epilogue == LECCR_EPI_DQUICKGELU || p.epilogue == LECCR_EPI_DGELU || p.epilogue == LECCR_EPI_MULAUX) && p.c8_fmt == 1;
    return false;
}

bool gemm256w4_legal(const leccr_gemm_params& p) {
    if (p.trans_a != 0 || p.trans_b != 0 || p.c_dtype != LECCR_BF16 || p.ab_dtype != LECCR_BF16 || p.accumulate) return false;
    if (p.M < 256 || p.N < 256 || (p.N & 31) || (p.ldc & 7) || (p.ldr & 7) || (p.ldaux & 7) || (p.K % g4::BK)) return false;
    if (p.lda >= (1 << 22) || p.ldb >= (1 << 22)) return false;      // 24-bit row * (ld * 2) in the staging addresses
    const int f = w4_flags(p);
    switch (p.epilogue) {
        case LECCR_EPI_NONE: return f == 0 || f == g4::F_RES;
        case LECCR_EPI_QUICKGELU:
        case LECCR_EPI_GELU:
        case LECCR_EPI_QUICKGELU_D:
        case LECCR_EPI_GELU_D: return f == g4::F_AUXOUT;
        case LECCR_EPI_DQUICKGELU:
        case LECCR_EPI_DGELU:
        case LECCR_EPI_MULAUX: return f == 0;
    }
    return false;
}

namespace g4 {
template <int EPI, int FLAGS>
static void launch_fp8(const Args& a, int grid, hipStream_t st, bool burst1) {
    if (burst1) launch_inst<EPI, FLAGS | F_BURST1>(a, grid, st);
    else launch_inst<EPI, FLAGS>(a, grid, st);
}
}  // namespace g4

int gemm256w4_launch(const leccr_gemm_params& p, hipStream_t st) {
    using namespace g4;
    Args a{};
    a.A = (const bf16_t*)p.A; a.B = (const bf16_t*)p.B; a.C = (bf16_t*)p.C;
    a.M = p.M; a.N = p.N; a.K = p.K; a.lda = p.lda; a.ldb = p.ldb; a.ldc = p.ldc;
    a.bias = p.bias; a.aux = (const bf16_t*)p.aux; a.ldaux = p.ldaux; a.aux_out = (bf16_t*)p.aux_out;
    a.residual = (const bf16_t*)p.residual; a.ldr = p.ldr; a.alpha = p.alpha;
    a.tiles_m = (int)cdiv(p.M, BM); a.tiles_n = (int)cdiv(p.N, BN);
    a.nkt = (int)(p.K / BK);
    a.n_items = a.tiles_m * a.tiles_n;
    const int grid = a.n_items < 256 ? a.n_items : 256;
    const int f = w4_flags(p);
    a.alpha_dev[0] = p.alpha_dev[0]; a.alpha_dev[1] = p.alpha_dev[1];
    if (p.ab_dtype == LECCR_FP8 || p.ab_dtype == LECCR_FP8_BF8A) {
        static const int burst_env = [] { const char* e = getenv("LECCR_FP8_BURST1"); return e ? atoi(e) : -1; }();      // 0 never, 1 always
        const bool b1 = burst_env < 0 ? p.K <= 1536 : burst_env != 0;          // F_BURST1: the short reductions
        // the bf16 problem with half the columns: byte-identical staging (see F_FP8)
        a.K = p.K / 2; a.lda = p.lda / 2; a.ldb = p.ldb / 2;
        a.nkt = (int)(a.K / BK);
        a.c8 = (uint8_t*)p.c8; a.ldc8 = p.ldc8; a.c8_site = p.c8_site; a.colsum = p.colsum;
        if (p.c8) {     // the C written is also the next product's fp8 operand: activation output (e4m3) / its gradient (e5m2)
            if (p.ab_dtype == LECCR_FP8 && p.epilogue == LECCR_EPI_QUICKGELU) launch_fp8<LECCR_EPI_QUICKGELU, F_AUXOUT | F_FP8 | F_Q8>(a, grid, st, b1);
            else if (p.ab_dtype == LECCR_FP8 && p.epilogue == LECCR_EPI_QUICKGELU_D) launch_fp8<LECCR_EPI_QUICKGELU_D, F_AUXOUT | F_FP8 | F_Q8>(a, grid, st, b1);
            else if (p.ab_dtype == LECCR_FP8 && p.epilogue == LECCR_EPI_GELU_D) launch_fp8<LECCR_EPI_GELU_D, F_AUXOUT | F_FP8 | F_Q8>(a, grid, st, b1);
            else if (p.ab_dtype == LECCR_FP8) launch_fp8<LECCR_EPI_GELU, F_AUXOUT | F_FP8 | F_Q8>(a, grid, st, b1);
            else if (p.epilogue == LECCR_EPI_MULAUX) launch_fp8<LECCR_EPI_MULAUX, F_FP8 | F_BF8A | F_Q8 | F_Q8_BF8>(a, grid, st, b1);
            else if (p.epilogue == LECCR_EPI_DQUICKGELU) launch_fp8<LECCR_EPI_DQUICKGELU, F_FP8 | F_BF8A | F_Q8 | F_Q8_BF8>(a, grid, st, b1);
            else launch_fp8<LECCR_EPI_DGELU, F_FP8 | F_BF8A | F_Q8 | F_Q8_BF8>(a, grid, st, b1);
            return check_launch("gemm256w4 fp8 q8");
        }
        if (p.ab_dtype == LECCR_FP8) {
            if (p.epilogue == LECCR_EPI_NONE && f == F_RES) launch_fp8<LECCR_EPI_NONE, F_RES | F_FP8>(a, grid, st, b1);
            else if (p.epilogue == LECCR_EPI_NONE) launch_fp8<LECCR_EPI_NONE, F_FP8>(a, grid, st, b1);
            else if (p.epilogue == LECCR_EPI_QUICKGELU) launch_fp8<LECCR_EPI_QUICKGELU, F_AUXOUT | F_FP8>(a, grid, st, b1);
            else if (p.epilogue == LECCR_EPI_QUICKGELU_D) launch_fp8<LECCR_EPI_QUICKGELU_D, F_AUXOUT | F_FP8>(a, grid, st, b1);
            else if (p.epilogue == LECCR_EPI_GELU_D) launch_fp8<LECCR_EPI_GELU_D, F_AUXOUT | F_FP8>(a, grid, st, b1);
            else launch_fp8<LECCR_EPI_GELU, F_AUXOUT | F_FP8>(a, grid, st, b1);
        } else {
            if (p.epilogue == LECCR_EPI_NONE && f == F_RES) launch_fp8<LECCR_EPI_NONE, F_RES | F_FP8 | F_BF8A>(a, grid, st, b1);
            else if (p.epilogue == LECCR_EPI_NONE) launch_fp8<LECCR_EPI_NONE, F_FP8 | F_BF8A>(a, grid, st, b1);
            else if (p.epilogue == LECCR_EPI_DQUICKGELU) launch_fp8<LECCR_EPI_DQUICKGELU, F_FP8 | F_BF8A>(a, grid, st, b1);
            else if (p.epilogue == LECCR_EPI_MULAUX) launch_fp8<LECCR_EPI_MULAUX, F_FP8 | F_BF8A>(a, grid, st, b1);
            else launch_fp8<LECCR_EPI_DGELU, F_FP8 | F_BF8A>(a, grid, st, b1);
        }
        return check_launch("gemm256w4 fp8");
    }
    switch (p.epilogue) {
        case LECCR_EPI_NONE:
            if (f == F_RES) launch_inst<LECCR_EPI_NONE, F_RES>(a, grid, st);
            else launch_inst<LECCR_EPI_NONE, 0>(a, grid, st);
            break;
        case LECCR_EPI_QUICKGELU: launch_inst<LECCR_EPI_QUICKGELU, F_AUXOUT>(a, grid, st); break;
        case LECCR_EPI_GELU: launch_inst<LECCR_EPI_GELU, F_AUXOUT>(a, grid, st); break;
        case LECCR_EPI_DQUICKGELU: launch_inst<LECCR_EPI_DQUICKGELU, 0>(a, grid, st); break;
        case LECCR_EPI_DGELU: launch_inst<LECCR_EPI_DGELU, 0>(a, grid, st); break;
        case LECCR_EPI_QUICKGELU_D: launch_inst<LECCR_EPI_QUICKGELU_D, F_AUXOUT>(a, grid, st); break;
        case LECCR_EPI_GELU_D: launch_inst<LECCR_EPI_GELU_D, F_AUXOUT>(a, grid, st); break;
        case LECCR_EPI_MULAUX: launch_inst<LECCR_EPI_MULAUX, 0>(a, grid, st); break;
        default: set_error("gemm256w4: bad epilogue %d", p.epilogue); return LECCR_EINVAL;
    }
    return check_launch("gemm256w4");
}

}  // namespace leccr
