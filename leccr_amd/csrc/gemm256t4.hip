// gemm256t4.hip - the weight-gradient product C[M,N] (f32) += alpha * A[K,M]^T B[K,N] (bf16 operands, TN) on a 256 x 256 x 64 tile
// run by FOUR waves, one per SIMD (round 3).
//
// Why.  gemm256.hip's TN instantiation (8 waves, wave tile 128 x 64, LOAD / COMPUTE segments alternating between the two waves of a
// SIMD, 8 barriers per K-tile) is the largest single kernel of the train step (33 ms of 137) and the one whose schedule has the most
// slack (profiles/r02_gemm256_pmc_step_shapes.txt: matrix pipe 0.49 busy, 39 % of wave-cycles waiting): a TN fragment is two
// transposing LDS reads whose results form one operand, and with 128 accumulator + 96 fragment registers per wave the 8-wave kernel
// cannot software-pipeline them without spilling.  With four waves a wave owns 512 registers:
//   * wave tile 128 x 128 = 4 x 4 accumulators of v_mfma_f32_32x32x16_bf16 (256 registers), fragments of the NEXT k-step (32
//     registers) read while the current k-step's 16 MFMAs run - every transposing read and every DMA piece sits between two MFMAs,
//     ONE barrier per K-tile (the schedule of gemm256w4.hip);
//   * half the LDS fragment bytes per MFMA cycle of the 8-wave kernel (a fragment feeds four 32-cycle MFMAs instead of 4-8 16-cycle ones);
//   * the MFMAs are issued A-role = A^T rows (m), B-role = B columns (n), so an accumulator register holds 32 consecutive n of two rows:
//     the split-K partial sums go to memory as global_atomic_add_f32 of 2 x 128 contiguous bytes STRAIGHT from the accumulators - no
//     LDS transpose in the epilogue (the 8-wave kernel's lane holds 4 n of one m and needs one);
//   * the bias gradient asum[m] += alpha sum_k A[k, m] is summed from the A fragments already in registers (v_dot2c_f32_bf16 against a
//     (1, 1) literal: 8 instructions per k-step), two of the wave's four m-tiles per wave column, the tiles_n items of a tile row taking
//     turns K-tile by K-tile as in gemm256.hip.  (Two more 32 x 32 MFMAs per k-step with an all-ones operand, as the 8-wave kernel does it
//     with 16 x 16 ones, made hipcc shuttle their 32 accumulator registers between VGPRs and AGPRs around every k-step.)
//
// LDS image of an operand K-tile: [64 k][256 columns] bf16, 512-byte rows, 16-byte chunks; physical chunk = logical chunk ^ ((k & 3) << 2):
// a ds_read_b64_tr_b16's 32-lane half takes 4 consecutive k x 64 bytes, and the XOR puts the four rows in the four 64-byte bank groups of
// the 256-byte bank window (plain rows would be 4-way conflicts: the row stride is a multiple of 256 bytes).  The swizzle is applied to the
// SOURCE address of the LDS-DMA (global_load_lds_dwordx4 writes lane-linear).
//
// Replaces the backward of every nn.Linear on the path (clip/model.py:177-191, HF BertModel dense layers via models/xvlm.py:238,
// models/attention.py:27-54): dW = dY^T X, db = colsum(dY).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace leccr {
namespace t4 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int OPT = 32768;                        // bytes of one operand K-tile (64 k x 256 columns)
constexpr int SLOT = 2 * OPT, NTH = 256;
constexpr int LDS_BYTES = 2 * SLOT;               // 131072: no epilogue staging
constexpr int F_ASUM = 1;
constexpr int F_ASUM3 = 2;       // with F_ASUM: A is a 3-segment split copy read as [3 rows, M] (below): asum counts its hi and lo segments once each
#ifndef T4_BURST
#define T4_BURST 0            // (experiment) bit ks: k-step ks of the bf16 kernel issues its fragment reads and DMA pieces in one burst ahead of its 16 MFMAs
#endif

struct Args {
    const bf16_t* A;
    const bf16_t* B;
    float* C;
    int64_t M, N, K, lda, ldb, ldc;
    float alpha;
    int tiles_m, tiles_n, splits, nkt, ktiles, n_items;        // nkt: K-tiles per split (the last one may have fewer)
    float* asum;
};

typedef __attribute__((ext_vector_type(4))) int int4_t;
typedef __attribute__((ext_vector_type(2))) int int2_t;

// one operand fragment of v_mfma_f32_32x32x16_bf16 from a [k][column] tile: lane (column r = lane & 31, k group h = lane >> 5) gets
// k = 8 h + 0..7 of its column: the 4 rows at IMM and the 4 rows below them (IMM + 4 * 512)
template <int IMM>
__device__ __forceinline__ bf16x8 lds_tr_pair(uint32_t addr) {
    int2_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(IMM));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(IMM + 2048));
    int4_t v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

// acc + the sum of the 8 bf16 of a fragment (v_dot2c_f32_bf16 with a packed (1, 1) literal): the bias gradient's share of one k-step
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float colsum8(const bf16x8& f, float acc) {
    const bf16x2_t one2 = {(bf16_t)1.0f, (bf16_t)1.0f};
#pragma unroll
    for (int w = 0; w < 4; ++w) acc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{f[2 * w], f[2 * w + 1]}, one2, acc, false);
    return acc;
}

// F_ASUM3 - the fp32 path's weight-gradient product (split.hip): A is the G-form copy [rows, 3 M] (hi | lo | hi) of an f32 dY read as [3 rows, M], so
// reduction index k = 3 r + s carries segment s of row r and the bias gradient colsum(dY) is the sum over k of the hi and lo segments ONCE each
// (s = 0, 1; the plain sum would count hi twice).  A fragment holds k = kbase + 0..7 of its column (element j = row j of the transposing reads:
// cdna_hip_programming.md T10), taken in as four pairs (k, k + 1) with k even; a pair's weights depend on k mod 3 only:
//   k mod 3 = 0: segments (0, 1) -> (1, 1)     1: segments (1, 2) -> (1, 0)     2: segments (2, 0) -> (0, 1)
// `ph` = kbase mod 3; pair w starts at k = kbase + 2 w: phases ph, ph + 2, ph + 1, ph.
__device__ __forceinline__ uint32_t asum3_weight(int ph) { return ph == 0 ? 0x3F803F80u : ph == 1 ? 0x00003F80u : 0x3F800000u; }   // bf16 pairs, element 0 in the low half
__device__ __forceinline__ float colsum8_x3(const bf16x8& f, uint32_t w0, uint32_t w1, uint32_t w2, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{f[0], f[1]}, __builtin_bit_cast(bf16x2_t, w0), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{f[2], f[3]}, __builtin_bit_cast(bf16x2_t, w2), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{f[4], f[5]}, __builtin_bit_cast(bf16x2_t, w1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{f[6], f[7]}, __builtin_bit_cast(bf16x2_t, w0), acc, false);
    return acc;
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

struct Cursor {                // position of one K-tile of this block's stream (wave-uniform)
    int j, kt, nkt;            // item ordinal, K-tile within the item, K-tiles of the item (the last split may be shorter)
    int kph;                   // (first K-tile of the item) mod 3: with 64 = 1 (mod 3), K-tile kt0 + kt starts at k = kt0 + kt (mod 3)  (F_ASUM3)
    int64_t m0, n0;
    const bf16_t* pa;          // A + k0 * lda + m0
    const bf16_t* pb;          // B + k0 * ldb + n0
};

__device__ __forceinline__ void decode(const Args& a, int first_item, int stride, Cursor& c) {
    const int item = first_item + c.j * stride;
    const int tiles = a.tiles_m * a.tiles_n;
    const int split = item / tiles;
    const int tile = item - split * tiles;
    const int tm = tile / a.tiles_n;
    c.m0 = (int64_t)tm * BM;
    c.n0 = (int64_t)(tile - tm * a.tiles_n) * BN;
    const int kt0 = split * a.nkt, left = a.ktiles - kt0;
    c.kph = kt0 % 3;
    c.nkt = left < a.nkt ? left : a.nkt;
    const int64_t k0 = (int64_t)(kt0 + c.kt) * BK;
    c.pa = a.A + k0 * a.lda + c.m0;
    c.pb = a.B + k0 * a.ldb + c.n0;
}

// next K-tile of the stream; at the end it stays on the last one (harmless re-reads keep the vmcnt counts uniform)
__device__ __forceinline__ void advance(const Args& a, int first_item, int stride, int n_mine, Cursor& c) {
    if (c.kt + 1 < c.nkt) {
        ++c.kt;
        c.pa += (int64_t)BK * a.lda;
        c.pb += (int64_t)BK * a.ldb;
    } else if (c.j + 1 < n_mine) {
        ++c.j;
        c.kt = 0;
        decode(a, first_item, stride, c);
    }
}

// One DMA piece = 256 threads x 16 bytes = 8 k-rows of one operand tile.  PN 0..7: A rows 8 PN .., PN 8..15: B rows 8 (PN - 8) ..
// Thread -> (row r8 = tid >> 5 of the 8, physical chunk tid & 31); the logical chunk (the swizzle depends on k & 3 = r8 & 3 only) is the same
// for every piece: SGPR base + 32-bit byte offset.  No tails: the launcher takes this kernel for M, N multiples of 256 and K a multiple of
// 64 only (every Linear of the ViT-B / ViT-L / BERT geometries at the training batch sizes; the rest stays on gemm256.hip's TN kernel).
template <int PN>
__device__ __forceinline__ void issue_piece(const Args& a, const Cursor& c, uint32_t toffA, uint32_t toffB, uint32_t lds_slot_off,
                                            uint32_t wave_off) {
    constexpr bool isA = PN < 8;
    constexpr int P = PN & 7;
    const int64_t ld = isA ? a.lda : a.ldb;
    const char* ub = reinterpret_cast<const char*>(isA ? c.pa : c.pb) + (int64_t)(8 * P) * ld * 2;        // wave-uniform
    __builtin_amdgcn_global_load_lds(GLB_PTR(ub + (isA ? toffA : toffB)),
                                     (__attribute__((address_space(3))) void*)(size_t)(lds_slot_off + (isA ? 0 : OPT) + P * 4096 + wave_off),
                                     16, 0, 0);
}

// pieces issued behind MFMA i of k-step ks (0..15 = the 16 pieces of the K-tile being prefetched; -1: none).  The first pieces of K-tile
// t+2 go out in k-step 3, right behind the barrier that frees their slot, the rest in k-steps 0 (and 1) of the next K-tile.
// T4_SCHED 0: as gemm256w4.hip - eight in consecutive gaps of k-step 3, eight in k-step 0 (the last piece then has two k-steps to land);
// T4_SCHED 1: spread - one piece every third gap over k-steps 3, 0 and 1 (a piece costs the issuing wave ~60-80 cycles against a 32-cycle
//             MFMA gap: back to back they starve the matrix pipe for a whole burst; the last piece keeps one k-step + the barrier to land)
#ifndef T4_SCHED
#define T4_SCHED 1
#endif
constexpr int N3 = T4_SCHED == 1 ? 6 : 8;        // pieces issued in k-step 3 (the prologue issues as many of K-tile 1)
__host__ __device__ constexpr int piece_at(int ks, int i) {
    if constexpr (T4_SCHED == 1) {
        if (i % 3 != 2 && !(ks == 3 && i == 0)) return -1;       // gaps 2, 5, 8, 11, 14 (+ gap 0 of k-step 3)
        const int n = i / 3;                                     // 0..4
        if (ks == 3) return i == 0 ? 0 : 1 + n;                  // 0, 1..5
        if (ks == 0) return 6 + n;                               // 6..10
        if (ks == 1) return 11 + n;                              // 11..15
        return -1;
    } else {
        if (ks == 3) return i >= 8 ? i - 8 : -1;
        if (ks == 0) return i >= 8 ? N3 + i - 8 : -1;
        return -1;
    }
}

template <int FLAGS>
__global__ __launch_bounds__(NTH) void gemm256t4_kernel(Args a) {
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

    const uint32_t smem_off = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t wave_off = smem_off + (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u;
    // staging: thread -> row r8 of a piece's 8, physical chunk tid & 31, logical chunk lc
    const int st_r8 = tid >> 5, st_lc = (tid & 31) ^ ((st_r8 & 3) << 2);
    const uint32_t toffA = ((uint32_t)st_r8 * (uint32_t)a.lda + (uint32_t)st_lc * 8u) * 2u;
    const uint32_t toffB = ((uint32_t)st_r8 * (uint32_t)a.ldb + (uint32_t)st_lc * 8u) * 2u;
#define ISSUE_PN(PN, slot) issue_piece<(PN)>(a, cB, toffA, toffB, (slot), wave_off)

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

    // fragment addresses.  Lane 16 g + 4 q + p of a transposing read addresses row (k) 8 h + q (h = g >> 1), columns 16 (g & 1) + 4 p .. + 3
    // of a 32-column tile; tile ti of the wave's 128 columns starts at column 128 w + 32 ti: logical chunk 16 w + 4 ti + 2 (g & 1) + (p >> 1),
    // physical = logical ^ (q << 2) = 16 w + 4 (ti ^ q) + .. - one address register per tile (the XOR is not an immediate)
    uint32_t la[4], lb[4];
    {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = g >> 1;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const uint32_t rowb = (uint32_t)((8 * h + q) * 512), inb = (uint32_t)(8 * (p & 1));
            la[ti] = rowb + (uint32_t)((16 * wr + 4 * (ti ^ q) + 2 * (g & 1) + (p >> 1)) << 4) + inb;
            lb[ti] = rowb + (uint32_t)((16 * wc + 4 * (ti ^ q) + 2 * (g & 1) + (p >> 1)) << 4) + inb + (uint32_t)OPT;
        }
    }
    bf16x8 fa[2][4], fb[2][4];      // [k-step parity][32-column tile]
    // fragment i (0..3: A tile i, 4..7: B tile i - 4) of k-step KS (rows 16 KS ..) of the slot at byte S
#define RD_FRAG(buf, S, KS, i)                                                                 \
    do {                                                                                         \
        if ((i) < 4) fa[buf][(i) & 3] = lds_tr_pair<(KS) * 8192>(la[(i) & 3] + smem_off + (S));  \
        else fb[buf][(i) & 3] = lds_tr_pair<(KS) * 8192>(lb[(i) & 3] + smem_off + (S));          \
    } while (0)
    static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; RD_FRAG(0, 0u, 0, i); });
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]));
    __builtin_amdgcn_sched_barrier(0);

    // bias gradient: column sums of the A panel, for the two m-tiles 2 wc, 2 wc + 1 of this wave's four (its partner wave column takes the
    // others), in the K-tiles whose turn it is for this item (tile column c takes K-tiles kt = c mod tiles_n: see gemm256.hip)
    [[maybe_unused]] float cs[2] = {0.f, 0.f};
    [[maybe_unused]] const bool cs_on = (FLAGS & F_ASUM) != 0 && a.asum != nullptr;

    int t = 0;
    for (int j = 0; j < n_mine; ++j) {
        f32x16 acc[16];                      // [mi * 4 + ni]
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
        if constexpr ((FLAGS & F_ASUM) != 0) { cs[0] = 0.f; cs[1] = 0.f; }
        [[maybe_unused]] int cs_ctr = (int)(cur.n0 >> 8);      // K-tiles until this item's next turn (BN = 256)
        [[maybe_unused]] int kph = cur.kph;                    // (this K-tile's first k) mod 3  (F_ASUM3)

        for (int kt = 0; kt < cur.nkt; ++kt, ++t) {
            const uint32_t sc = (t & 1) * SLOT, sn = SLOT - sc;
            [[maybe_unused]] bool do_cs = false;
            if constexpr ((FLAGS & F_ASUM) != 0) {
                do_cs = cs_on && cs_ctr == 0;
                cs_ctr = cs_ctr == 0 ? a.tiles_n - 1 : cs_ctr - 1;
            }
            // per-lane address pieces go through an opaque asm each K-tile so that the derived addresses are formed at their use
            asm volatile("" : "+v"(la[0]), "+v"(la[1]), "+v"(la[2]), "+v"(la[3]), "+v"(lb[0]), "+v"(lb[1]), "+v"(lb[2]), "+v"(lb[3]));
            // k-step KS: 16 MFMAs on buffer KS & 1; fill(i) behind MFMA i
            auto kstep = [&](auto ks_tag, auto&& fill) {
                constexpr int KS = decltype(ks_tag)::value, cb = KS & 1;
                constexpr bool BURST = ((T4_BURST >> KS) & 1) != 0;      // the k-step's reads / DMA issues in one burst ahead of its MFMAs
                if constexpr (BURST) {
                    static_for<0, 16>([&](auto ic) { fill(ic); });
                    __builtin_amdgcn_sched_barrier(0);
                }
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, mi = i >> 2, ni = i & 3;
                    acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][mi], fb[cb][ni], acc[mi * 4 + ni], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!BURST) fill(ic);
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr ((FLAGS & F_ASUM3) != 0) {
                    if (do_cs) {           // this lane's fragment starts at k = 64 (kt0 + kt) + 16 KS + 8 h: kph + KS + 2 h (mod 3)
                        int ph = kph + KS + ((lane >> 5) << 1);
                        ph = ph >= 6 ? ph - 6 : ph >= 3 ? ph - 3 : ph;
                        const int ph1 = ph == 2 ? 0 : ph + 1, ph2 = ph1 == 2 ? 0 : ph1 + 1;
                        const uint32_t w0 = asum3_weight(ph), w1 = asum3_weight(ph1), w2 = asum3_weight(ph2);
                        // (a fragment's four dot2 start from 0 and join the running sum by one v_add_f32: the dot2 accumulate does not round to
                        //  nearest, and carried through a 36928-row column of one sign its error reached 5e-6 of the sum - the fp32 path's bias
                        //  gradients are held to the split's 2^-17 per term, not to a drift)
                        if (wc == 0) { cs[0] += colsum8_x3(fa[cb][0], w0, w1, w2, 0.f); cs[1] += colsum8_x3(fa[cb][1], w0, w1, w2, 0.f); }
                        else { cs[0] += colsum8_x3(fa[cb][2], w0, w1, w2, 0.f); cs[1] += colsum8_x3(fa[cb][3], w0, w1, w2, 0.f); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if constexpr ((FLAGS & F_ASUM) != 0) {
                    if (do_cs) {           // (the fragments just used: their registers are rewritten only by the NEXT k-step's fill)
                        if (wc == 0) { cs[0] = colsum8(fa[cb][0], cs[0]); cs[1] = colsum8(fa[cb][1], cs[1]); }
                        else { cs[0] = colsum8(fa[cb][2], cs[0]); cs[1] = colsum8(fa[cb][3], cs[1]); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
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
        if constexpr (i < 8) RD_FRAG(buf, S, RKS, i);                                         \
        if constexpr (piece_at(KS, i) >= 0) ISSUE_PN(piece_at(KS, i), slot);                  \
    }
            // k-steps 0..2 read the next k-step of this slot and issue the rest of K-tile t+1 (into the other slot)
            kstep(std::integral_constant<int, 0>{}, FILL(0, 1, sc, 1, sn));
            WAIT_FRAGS(1);
            kstep(std::integral_constant<int, 1>{}, FILL(1, 0, sc, 2, sn));
            WAIT_FRAGS(0);
            kstep(std::integral_constant<int, 2>{}, FILL(2, 1, sc, 3, sn));
            advance(a, first_item, stride, n_mine, cB);           // cB -> K-tile t+2
            WAIT_FRAGS(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of K-tile t+1 have landed
            SEG_BARRIER();                                        // slot sc is free, slot sn is complete
            // k-step 3 reads k-step 0 of K-tile t+1 and starts K-tile t+2 into the slot just freed
            kstep(std::integral_constant<int, 3>{}, FILL(3, 0, sn, 0, sc));
            WAIT_FRAGS(0);
            if constexpr ((FLAGS & F_ASUM3) != 0) kph = kph == 2 ? 0 : kph + 1;
        }

        // ---- epilogue of this item: f32 atomics straight from the accumulators.  Register i of acc[mi * 4 + ni], lane (r, h):
        // C[m = m0 + 128 wr + 32 mi + (i & 3) + 8 (i >> 2) + 4 h][n = n0 + 128 wc + 32 ni + r] - 2 x 128 contiguous bytes per instruction
        {
            const int64_t m0 = cur.m0 + wr * 128, n0 = cur.n0 + wc * 128;
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int r = lane_e & 31, h = lane_e >> 5;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    float* cp = a.C + (m0 + 32 * mi + 4 * h) * a.ldc + n0 + 32 * ni + r;
#pragma unroll
#ifdef T4_EXP_NOATOMIC
                    for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(acc[mi * 4 + ni][i]));       // diagnostic build: no epilogue atomics (wrong results)
#else
                    for (int i = 0; i < 16; ++i) atomicAdd(cp + (int64_t)((i & 3) + 8 * (i >> 2)) * a.ldc, acc[mi * 4 + ni][i] * a.alpha);
#endif
                }
            if constexpr ((FLAGS & F_ASUM) != 0) {
                // cs[u]: sum over this lane's k half (8 h ..) of column m = m0 + 32 (2 wc + u) + r: add the two halves, one atomic per column
                if (cs_on) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(cs[u]), __float_as_uint(cs[u]), false, false);
                        const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                        if (h == 0) atomicAdd(a.asum + m0 + 32 * (2 * wc + u) + r, tot * a.alpha);
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);       // (keeps hipcc's conservative vmcnt(0) out of the K loop, as in gemm256.hip)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j + 1 < n_mine) {
            cur.j = j + 1;
            cur.kt = 0;
            decode(a, first_item, stride, cur);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail prefetches still target this block's LDS
}

// ---------------------------------------------------------------------------------------------------- fp8 operands (BASELINE configs[4])
// The same kernel with OCP fp8 operands: A = dY in e5m2 [K, M], B = X in e4m3 [K, N] (the copies the forward / dX products of the Linear
// already use, row-major as they are - no transposed copies), v_mfma_scale_f32_32x32x64_f8f6f4 with block scales 1.0; the per-tensor
// inverse scales are device scalars folded into alpha at kernel start.  A K-tile is 128 k: an operand tile [128 k][256 bytes] has the
// bytes of the bf16 tile, 16 DMA pieces and 64 transposing reads per K-tile as before, and 32 scaled MFMAs (64 cycles each) in place of
// 64 bf16 ones - twice the k per K-tile in the same matrix-pipe time.
//   * ds_read_b64_tr_b8 (probed: scripts/mb/ds_tr8_probe.hip): per 16 lanes an 8 row x 16 byte block; lane 2 q + p supplies the address of
//     row q, bytes 8 p .. 8 p + 7; lane i receives byte column i of the 8 rows.  A fragment = 4 such reads = 32 k of one column; which k
//     a lane half holds is immaterial as long as A and B agree (here k = 64 ks + 32 h + 8 j + e for read j, element e).
//   * LDS image: 256-byte rows; the 8 rows of a read would share banks (row stride = the bank window), so the 32-byte group index of a
//     chunk is XORed with k & 7: physical 16-byte chunk = logical ^ ((k & 7) << 1).
//   * the bias gradient asum[m] += alpha / scale_A * sum_k A8[k, m] (round 4: the column sums of the e5m2 copy - the values the product itself
//     multiplies - in place of a pass over the bf16 tensor) is summed on the VALU from the A fragments already in registers: an e5m2 byte is the
//     high byte of an f16, so a fragment word gives two packed-f16 pairs (v_and / v_perm) and v_dot2_f32_f16 against (1, 1) adds each pair
//     into an f32 - 32 instructions per fragment, two of the wave's four m-tiles per wave column, spread behind the k-step's MFMAs; the sums of a
//     tile row are taken by the items of its first tile column over their whole K range (one K loop per case, chosen per item).  (On the matrix pipe - the fragment as the first operand of a 16 x 16 x 128 MFMA against a 0 / 1
//     selector - the 4-register accumulators do not fit beside the product's 256: hipcc moves acc[0] out of the AGPRs and back around every
//     k-step; as inline assembly on VGPR accumulators with one straight-line K-tile body per case it spilled 269 registers.)
typedef __attribute__((ext_vector_type(8))) int v8i_t;

template <int IMM>
__device__ __forceinline__ v8i_t lds_tr8_quad(uint32_t addr) {       // 4 x 8 rows: IMM, + 8, + 16, + 24 rows (256 bytes each)
    int2_t r0, r1, r2, r3;
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r0) : "v"(addr), "n"(IMM));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r1) : "v"(addr), "n"(IMM + 2048));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r2) : "v"(addr), "n"(IMM + 4096));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(r3) : "v"(addr), "n"(IMM + 6144));
    return v8i_t{r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
}

typedef _Float16 __attribute__((ext_vector_type(2))) half2_t;

// acc += the 8 e5m2 values in words w0, w0 + 1 of a fragment: byte b as an f16 is b << 8
__device__ __forceinline__ float colsum8_bf8(const v8i_t& f, int w0, float acc) {
    const half2_t one2 = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int w = w0; w < w0 + 2; ++w) {
        const uint32_t x = (uint32_t)f[w];
        const uint32_t odd = x & 0xFF00FF00u;                               // bytes 1, 3
        const uint32_t even = __builtin_amdgcn_perm(0u, x, 0x020C000Cu);     // bytes 0, 2 moved up: (b2 << 24) | (b0 << 8)
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, odd), one2, acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, even), one2, acc, false);
    }
    return acc;
}

struct Args8 {
    const uint8_t* A;
    const uint8_t* B;
    float* C;
    int64_t M, N, K, lda, ldb, ldc;
    float alpha;
    const float* alpha_dev[2];
    int tiles_m, tiles_n, splits, nkt, ktiles, n_items;
    float* asum;
};
constexpr int BK8 = 128;
#ifndef T4_FP8_SCHED
#define T4_FP8_SCHED 1        // bit ks: k-step ks issues its fragment reads and DMA pieces in one burst ahead of its 16 MFMAs instead of between them
#endif

struct Cursor8 {
    int j, kt, nkt;
    int64_t m0, n0;
    const uint8_t* pa;
    const uint8_t* pb;
};

__device__ __forceinline__ void decode8(const Args8& a, int first_item, int stride, Cursor8& c) {
    const int item = first_item + c.j * stride;
    const int tiles = a.tiles_m * a.tiles_n;
    const int split = item / tiles;
    const int tile = item - split * tiles;
    const int tm = tile / a.tiles_n;
    c.m0 = (int64_t)tm * BM;
    c.n0 = (int64_t)(tile - tm * a.tiles_n) * BN;
    const int kt0 = split * a.nkt, left = a.ktiles - kt0;
    c.nkt = left < a.nkt ? left : a.nkt;
    const int64_t k0 = (int64_t)(kt0 + c.kt) * BK8;
    c.pa = a.A + k0 * a.lda + c.m0;
    c.pb = a.B + k0 * a.ldb + c.n0;
}

__device__ __forceinline__ void advance8(const Args8& a, int first_item, int stride, int n_mine, Cursor8& c) {
    if (c.kt + 1 < c.nkt) {
        ++c.kt;
        c.pa += (int64_t)BK8 * a.lda;
        c.pb += (int64_t)BK8 * a.ldb;
    } else if (c.j + 1 < n_mine) {
        ++c.j;
        c.kt = 0;
        decode8(a, first_item, stride, c);
    }
}

// One DMA piece = 256 threads x 16 bytes = 16 k-rows (256 bytes each) of one operand tile; PN 0..7: A rows 16 PN .., 8..15: B.
template <int PN>
__device__ __forceinline__ void issue_piece8(const Args8& a, const Cursor8& c, uint32_t toffA, uint32_t toffB, uint32_t lds_slot_off,
                                             uint32_t wave_off) {
    constexpr bool isA = PN < 8;
    constexpr int P = PN & 7;
    const int64_t ld = isA ? a.lda : a.ldb;
    const char* ub = reinterpret_cast<const char*>(isA ? c.pa : c.pb) + (int64_t)(16 * P) * ld;        // wave-uniform
    __builtin_amdgcn_global_load_lds(GLB_PTR(ub + (isA ? toffA : toffB)),
                                     (__attribute__((address_space(3))) void*)(size_t)(lds_slot_off + (isA ? 0 : OPT) + P * 4096 + wave_off),
                                     16, 0, 0);
}

// pieces behind MFMA i of k-step ks (two k-steps of 16 scaled MFMAs per K-tile: k-step 1 follows the barrier that frees the slot of K-tile
// t+2, k-step 0 of the next K-tile takes the rest): every second gap (a scaled MFMA gap is 64 cycles)
__host__ __device__ constexpr int piece8_at(int ks, int i) {
    if ((i & 1) == 0) return -1;
    return ks == 1 ? i / 2 : 8 + i / 2;
}

template <int FLAGS>
__global__ __launch_bounds__(NTH) void gemm256t4_fp8_kernel(Args8 a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
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
    if (a.alpha_dev[0]) a.alpha *= *a.alpha_dev[0];          // 1 / (scale_A scale_B), both kept on the device
    [[maybe_unused]] const float alpha_cs = a.alpha;          // the A sums carry A's scale only
    if (a.alpha_dev[1]) a.alpha *= *a.alpha_dev[1];

    const uint32_t smem_off = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t wave_off = smem_off + (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u;
    // staging: thread -> row r16 = tid >> 4 of a piece's 16, physical chunk tid & 15, logical chunk = physical ^ ((row & 7) << 1)
    const int st_r = tid >> 4, st_lc = (tid & 15) ^ ((st_r & 7) << 1);
    const uint32_t toffA = (uint32_t)st_r * (uint32_t)a.lda + (uint32_t)st_lc * 16u;
    const uint32_t toffB = (uint32_t)st_r * (uint32_t)a.ldb + (uint32_t)st_lc * 16u;
#define ISSUE8(PN, slot) issue_piece8<(PN)>(a, cB, toffA, toffB, (slot), wave_off)

    Cursor8 cB, cur;
    cur.j = 0; cur.kt = 0;
    decode8(a, first_item, stride, cur);
    cB = cur;
    static_for<0, 16>([&](auto pn) { ISSUE8(decltype(pn)::value, 0u); });
    advance8(a, first_item, stride, n_mine, cB);                    // cB -> K-tile 1
    static_for<0, 8>([&](auto pn) { ISSUE8(decltype(pn)::value, (uint32_t)SLOT); });
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    SEG_BARRIER();

    // fragment addresses: lane 16 g + 2 q + p of a transposing read addresses row 32 h + q (h = g >> 1), bytes 16 (g & 1) + 8 p .. of a
    // 32-column tile; tile ti of the wave's 128 columns is 32-byte group 4 w + ti of the 256-byte row; physical group = logical ^ (row & 7)
    uint32_t la[4], lb[4];
    {
        const int g = lane >> 4, i = lane & 15, q = i >> 1, pp = i & 1, h = g >> 1;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const uint32_t rowb = (uint32_t)((32 * h + q) * 256), inb = (uint32_t)(16 * (g & 1) + 8 * pp);
            la[ti] = smem_off + rowb + (uint32_t)(((4 * wr + ti) ^ q) << 5) + inb;
            lb[ti] = smem_off + rowb + (uint32_t)(((4 * wc + ti) ^ q) << 5) + inb + (uint32_t)OPT;
        }
    }
    v8i_t fa[2][4], fb[2][4];
#define RD_FRAG8(buf, S, KS, i)                                                          \
    do {                                                                                   \
        if ((i) < 4) fa[buf][(i) & 3] = lds_tr8_quad<(KS) * 16384>(la[(i) & 3] + (S));     \
        else fb[buf][(i) & 3] = lds_tr8_quad<(KS) * 16384>(lb[(i) & 3] + (S));             \
    } while (0)
#define WAIT_FRAGS8(buf)                                                                                                    \
    do {                                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                     : "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fa[buf][2]), "+v"(fa[buf][3]), "+v"(fb[buf][0]),            \
                       "+v"(fb[buf][1]), "+v"(fb[buf][2]), "+v"(fb[buf][3]));                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
    } while (0)
    static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; RD_FRAG8(0, 0u, 0, i); });
    WAIT_FRAGS8(0);

    // bias gradient (F_ASUM): the second operand of the row-sum MFMA - lane 16 g + n is column n, k group g: column 0 = 1.0 (e2m1: 0x2) in
    // the even groups, column 1 in the odd ones, everything else 0
    // bias gradient (F_ASUM): cs[u] = this lane's share (its k half) of the sum of column 32 (2 wc + u) + r of the wave's A rows
    [[maybe_unused]] float cs[2] = {0.f, 0.f};
    [[maybe_unused]] const bool cs_on = (FLAGS & F_ASUM) != 0 && a.asum != nullptr;

    int t = 0;
    for (int j = 0; j < n_mine; ++j) {
        f32x16 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
        if constexpr ((FLAGS & F_ASUM) != 0) { cs[0] = 0.f; cs[1] = 0.f; }
        // The K loop exists once per sum mode (0: no sums, 1 / 2: m-tiles (0, 1) / (2, 3) of the A fragments), chosen per ITEM: a branch
        // inside the loop body splits its basic block and hipcc then sinks all 32 MFMAs of a K-tile below the fragment reads (seen in the
        // listing).  The sums of a tile row's A panel are taken by its first tile column's items over their whole K range: the VALU work
        // rides under the MFMAs.
        auto kloop = [&](auto cs_tag) __attribute__((always_inline)) {
            constexpr int CS = decltype(cs_tag)::value;
            for (int kt = 0; kt < cur.nkt; ++kt, ++t) {
                const uint32_t sc = (t & 1) * SLOT, sn = SLOT - sc;
                asm volatile("" : "+v"(la[0]), "+v"(la[1]), "+v"(la[2]), "+v"(la[3]), "+v"(lb[0]), "+v"(lb[1]), "+v"(lb[2]), "+v"(lb[3]));
                // the prefetch position after this K-tile's first k-step (K-tile t+2) is worked out HERE: the branches of advance8() between
                // the two k-steps ended the first one's basic block, and hipcc sank its 16 MFMAs below them - behind all of its fragment
                // reads and DMA issues instead of between them (round 4, seen in the listing: the matrix pipe idled through that burst)
                Cursor8 cN = cB;
                advance8(a, first_item, stride, n_mine, cN);
                auto kstep = [&](auto ks_tag, auto&& fill) {
                    constexpr int cb = decltype(ks_tag)::value & 1;
                    constexpr bool BURST = ((T4_FP8_SCHED >> decltype(ks_tag)::value) & 1) != 0;      // the k-step's reads / DMA issues ahead of its MFMAs
                    if constexpr (BURST) {
                        static_for<0, 16>([&](auto ic) { fill(ic); });
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    static_for<0, 16>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, mi = i >> 2, ni = i & 3;
                        // first operand = dY^T rows (e5m2: format 1), second = X columns (e4m3: format 0); block scales 2^0
                        acc[mi * 4 + ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[cb][mi], fb[cb][ni], acc[mi * 4 + ni], 1, 0, 0,
                                                                                           0x7F7F7F7F, 0, 0x7F7F7F7F);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (!BURST) fill(ic);
                        // (the sums of the fragments in use: a share behind each of the k-step's last eight MFMAs)
                        if constexpr (CS != 0 && i >= 8) cs[(i >> 2) & 1] = colsum8_bf8(fa[cb][2 * (CS - 1) + ((i >> 2) & 1)], 2 * (i & 3), cs[(i >> 2) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                };
#define FILL8T(KSP, buf, S, RKS, slot)                                                        \
    [&](auto ic) {                                                                            \
        constexpr int i = decltype(ic)::value;                                                \
        if constexpr (i < 8) RD_FRAG8(buf, S, RKS, i);                                        \
        if constexpr (piece8_at(KSP, i) >= 0) ISSUE8(piece8_at(KSP, i), slot);                \
    }
                // k-step 0 (buffer 0): reads k-step 1 of this slot; the rest (pieces 8-15) of K-tile t+1  -> wait; vmcnt(0); BARRIER
                kstep(std::integral_constant<int, 0>{}, FILL8T(0, 1, sc, 1, sn));
                cB = cN;                                              // cB -> K-tile t+2
                WAIT_FRAGS8(1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SEG_BARRIER();
                // k-step 1 (buffer 1): reads k-step 0 of K-tile t+1; pieces 0-7 of K-tile t+2 into the slot the barrier freed
                kstep(std::integral_constant<int, 1>{}, FILL8T(1, 0, sn, 0, sc));
                WAIT_FRAGS8(0);
            }
        };
        [[maybe_unused]] const bool cs_item = cs_on && cur.n0 == 0;
        if constexpr ((FLAGS & F_ASUM) != 0) {
            if (!cs_item) kloop(std::integral_constant<int, 0>{});
            else if (wc == 0) kloop(std::integral_constant<int, 1>{});
            else kloop(std::integral_constant<int, 2>{});
        } else {
            kloop(std::integral_constant<int, 0>{});
        }
        {
            const int64_t m0 = cur.m0 + wr * 128, n0 = cur.n0 + wc * 128;
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int r = lane_e & 31, h = lane_e >> 5;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    float* cp = a.C + (m0 + 32 * mi + 4 * h) * a.ldc + n0 + 32 * ni + r;
#pragma unroll
                    for (int i = 0; i < 16; ++i) atomicAdd(cp + (int64_t)((i & 3) + 8 * (i >> 2)) * a.ldc, acc[mi * 4 + ni][i] * a.alpha);
                }
            if constexpr ((FLAGS & F_ASUM) != 0) {
                // cs[u]: sum over this lane's k half of column m = m0 + 32 (2 wc + u) + r: add the two halves, one atomic per column
                if (cs_item) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(cs[u]), __float_as_uint(cs[u]), false, false);
                        const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                        if (h == 0) atomicAdd(a.asum + m0 + 32 * (2 * wc + u) + r, tot * alpha_cs);
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j + 1 < n_mine) {
            cur.j = j + 1;
            cur.kt = 0;
            decode8(a, first_item, stride, cur);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int FLAGS>
static void launch_inst(const Args& a, int grid, hipStream_t st) {
    auto kfn = gemm256t4_kernel<FLAGS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NTH), LDS_BYTES, st, a);
}

}  // namespace t4

// fp8 weight-gradient products: A = dY e5m2 [K, M], B = X e4m3 [K, N] (ab_dtype LECCR_FP8_BF8A, trans_a = trans_b = 1), f32 C in place
bool gemm256t4_fp8_legal(const leccr_gemm_params& p) {
    if (p.trans_a != 1 || p.trans_b != 1 || p.ab_dtype != LECCR_FP8_BF8A || p.c_dtype != LECCR_F32 || !p.accumulate) return false;
    if (p.epilogue != LECCR_EPI_NONE || p.residual || p.aux || p.aux_out || p.bias || p.colsum || p.c8) return false;
    if (p.M < 256 || p.N < 256 || (p.M & 255) || (p.N & 255) || (p.K & 127) || (p.lda & 15) || (p.ldb & 15)) return false;
    if ((((uintptr_t)p.A | (uintptr_t)p.B) & 15) || p.lda >= (1 << 23) || p.ldb >= (1 << 23)) return false;
    return p.K >= 2048;
}

int gemm256t4_fp8_launch(const leccr_gemm_params& p, int cap, hipStream_t st) {
    using namespace t4;
    Args8 a{};
    a.A = (const uint8_t*)p.A; a.B = (const uint8_t*)p.B; a.C = (float*)p.C;
    a.M = p.M; a.N = p.N; a.K = p.K; a.lda = p.lda; a.ldb = p.ldb; a.ldc = p.ldc;
    a.alpha = p.alpha; a.alpha_dev[0] = p.alpha_dev[0]; a.alpha_dev[1] = p.alpha_dev[1]; a.asum = p.asum;
    a.tiles_m = (int)cdiv(p.M, BM); a.tiles_n = (int)cdiv(p.N, BN);
    const int64_t ktiles = p.K / BK8;
    const int tiles = a.tiles_m * a.tiles_n;
    int splits = tiles >= cap ? 1 : cap / tiles;
    if (splits > ktiles / 4) splits = (int)(ktiles / 4);
    if (splits < 1) splits = 1;
    a.nkt = (int)cdiv(ktiles, splits);
    splits = (int)cdiv(ktiles, a.nkt);
    a.splits = splits;
    a.ktiles = (int)ktiles;
    a.n_items = tiles * splits;
    const int grid = a.n_items < cap ? a.n_items : cap;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm256t4_fp8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm256t4_fp8_kernel<F_ASUM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    if (a.asum) hipLaunchKernelGGL(gemm256t4_fp8_kernel<F_ASUM>, dim3(grid), dim3(NTH), LDS_BYTES, st, a);
    else hipLaunchKernelGGL(gemm256t4_fp8_kernel<0>, dim3(grid), dim3(NTH), LDS_BYTES, st, a);
    return check_launch("gemm256t4 fp8");
}

// bf16 TN weight-gradient products this kernel takes: f32 C accumulated in place, no epilogue, whole 8-column chunks, rows 16-byte aligned
bool gemm256t4_legal(const leccr_gemm_params& p) {
    if (p.trans_a != 1 || p.trans_b != 1 || p.ab_dtype != LECCR_BF16 || p.c_dtype != LECCR_F32 || !p.accumulate) return false;
    if (p.epilogue != LECCR_EPI_NONE || p.residual || p.aux || p.aux_out || p.bias || p.colsum) return false;
    if (p.M < 256 || p.N < 256 || (p.M & 255) || (p.N & 255) || (p.K & 63) || (p.lda & 7) || (p.ldb & 7)) return false;      // no tails
    if ((((uintptr_t)p.A | (uintptr_t)p.B) & 15) || p.lda >= (1 << 22) || p.ldb >= (1 << 22)) return false;
    return p.K >= 4096;                // deep reductions only (split-K over up to 256 workgroups)
}

int gemm256t4_launch(const leccr_gemm_params& p, int cap, hipStream_t st) {
    using namespace t4;
    Args a{};
    a.A = (const bf16_t*)p.A; a.B = (const bf16_t*)p.B; a.C = (float*)p.C;
    a.M = p.M; a.N = p.N; a.K = p.K; a.lda = p.lda; a.ldb = p.ldb; a.ldc = p.ldc;
    a.alpha = p.alpha; a.asum = p.asum;
    a.tiles_m = (int)cdiv(p.M, BM); a.tiles_n = (int)cdiv(p.N, BN);
    const int64_t ktiles = cdiv(p.K, BK);
    const int tiles = a.tiles_m * a.tiles_n;
    // one item per CU in ONE round: tiles * splits <= cap (a second, partial round would double the time)
    int splits = tiles >= cap ? 1 : cap / tiles;
    if (splits > ktiles / 8) splits = (int)(ktiles / 8);
    if (splits < 1) splits = 1;
    a.nkt = (int)cdiv(ktiles, splits);
    splits = (int)cdiv(ktiles, a.nkt);        // no empty last split
    a.splits = splits;
    a.ktiles = (int)ktiles;
    a.n_items = tiles * splits;
    const int grid = a.n_items < cap ? a.n_items : cap;
    if (a.asum && p.asum_x3) launch_inst<F_ASUM | F_ASUM3>(a, grid, st);
    else if (a.asum) launch_inst<F_ASUM>(a, grid, st);
    else launch_inst<0>(a, grid, st);
    return check_launch("gemm256t4");
}

}  // namespace leccr
