// gemm256n4.hip - the 256x256x64 bf16 NT GEMM tile (forward / dX products, bf16 out) on FOUR waves with v_mfma_f32_16x16x32_bf16,
// just-in-time operand fragments and the finished output tile PARKED in registers while the next item's K loop runs: round 5's answer to
// "a GEMM main loop at the vendor's rate" (clip/model.py:177-191 is 70 % of the step's kernel time).
//
// What the measurements behind it say (profiles/r05_gemm_n4_v1_diagnostics.txt: a first version of this file with a four-stage ring of
// 64-byte rows, and diagnostic builds that switch its parts off one at a time; M = 147712 rows, qkv / fc1 shapes):
//   * MFMAs + fragment reads + barriers alone run at 1500-1700 TFLOP/s in this structure (the 8-wave kernel of gemm256.hip: 850-1000 on
//     the same shapes, everything included);
//   * issuing the LDS-DMA pieces costs ~4 % - what they fetch costs 25-40 %: a piece of 16 HALF cache lines (64-byte rows) is 12-14 %
//     slower than a piece of 8 whole lines, so the LDS image keeps 128-byte rows (one K-tile = one row) as in gemm256w4.hip;
//   * the epilogue is 26-28 % of a K = 768 launch and ALL of it is the global stores (LDS transposes, conversions, v_permlane swaps: ~3 %):
//     a CU drains stores at ~16 GB/s (128 KB tile = 8 us) and the vmcnt counter is in order, so the first "my DMA pieces have landed" wait
//     of the next item sits out every store of the tile.  Desynchronising the workgroups changes nothing (the limit is per CU).  Stores of
//     64-byte row segments (what a register-only epilogue can build from the MFMA layout) are slower still; 256 contiguous bytes per 16
//     lanes (the LDS-transposed layout), issued a few per K-tile UNDER the next item's K loop, hide a third to a half of the cost.
//
// Structure
//   * 256 threads = 4 waves = 2 (M) x 2 (N); wave tile 128 x 128 = 8 x 8 accumulators of v_mfma_f32_16x16x32_bf16 (256 AGPRs), issued with
//     swapped roles (the B rows are the A operand) so that a lane ends with 4 consecutive n of one row:
//     register r of accumulator (mi, ni) in lane l = C[m = 16 mi + (l & 15)][n = 16 ni + 4 (l >> 4) + r].
//   * LDS: 2 slots x {A0, A1, B0, B1} half-tiles of 16 KiB (128 rows x 64 k = 128-byte rows), physical 16-byte chunk = logical chunk ^
//     ((row >> 1) & 7) applied to the SOURCE address of the LDS-DMA pieces (8 rows x 128 bytes = 8 whole cache lines per wave instruction);
//     a fragment ds_read_b128 (row = l & 15, chunk 4 ks + (l >> 4)) is conflict-free under it.  + 4 x 8 KiB of transpose staging.
//   * K-tile = two stages (k-steps of 32) of 64 MFMAs = 8 groups (mi) x 8 (ni).  Fragments are read JUST IN TIME - 48 registers instead of
//     128: the A fragment of group g + 2 is read behind the first MFMA of group g (4 rotating slots), the B fragment ni of the NEXT stage
//     overwrites fb[ni] right behind its last use (MFMA (7, ni)); counted lgkmcnt waits in front of the consumers.
//   * ONE barrier per K-tile, in front of group 6 of the second stage: behind it the fragment reads belong to the next K-tile (other slot)
//     and this K-tile's slot takes K-tile t + 2: 3 DMA pieces right behind the barrier, 13 spread over the first 80 MFMAs of the next
//     K-tile - every piece has >= a quarter of a K-tile to land before the vmcnt wait in front of the next barrier.
//   * Epilogue: per 16-row strip a wave-private LDS transpose (f32) so that 16 consecutive lanes hold 256 contiguous bytes of one row;
//     bias / activation (+ saved pre-activation) / derivative / residual as gemm256.hip; the 32 resulting 16-byte vectors per lane stay in
//     registers ("parked", 128 VGPRs) and leave as 2-3 stores per K-tile behind the DMA pieces of the NEXT item's first HK K-tiles (those
//     K-tile bodies are unrolled: the parked registers need static indices), vmcnt(stores of this K-tile) in front of the barrier.
//   * Persistent items, XCD-contiguous, the K-tile stream continuous across items; an item's first stage starts its accumulators from
//     the instruction's zero operand.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace leccr {
#ifndef N4_EXP
#define N4_EXP 0              // diagnostic builds (wrong results): 1 no vmcnt wait, 2 no DMA in the loop, 4 no barrier, 16 no epilogue at all, 128 no global stores
#endif
#ifndef N4_PIN
#define N4_PIN 0
#endif
#ifndef N4_NT
#define N4_NT 1               // the parked tile leaves by non-temporal stores (0: default policy): fc1 745 -> 680 us, out 205 -> 193 (scripts/r5_n4_e.sh)
#endif
#ifndef N4_PARK
#define N4_PARK 32            // vectors of the finished tile (of 32 per lane) that stay in registers and leave under the next item's K loop; the
                              // others are stored by the epilogue itself
#endif
#ifndef N4_LATE
#define N4_LATE 3             // DMA pieces of K-tile t + 2 issued right behind the barrier (MFMAs 113 .. 127 of the K-tile body)
#endif
#ifndef N4_DMA_STEP
#define N4_DMA_STEP 6         // the other 16 - N4_LATE pieces: one behind every N4_DMA_STEP-th MFMA of the next K-tile body, from MFMA 2
#endif
#ifndef N4_LATE_STEP
#define N4_LATE_STEP 5
#endif
namespace g5 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 16384, SLOT = 4 * HALF, NTH = 256;
constexpr int STG_WAVE = 16 * 128 * 4;                      // 16 rows x 128 f32 per wave
constexpr int LDS_BYTES = 2 * SLOT + 4 * STG_WAVE;          // 131072 + 32768 = all of the CU's LDS
constexpr int F_RES = 2, F_AUXOUT = 4;
constexpr int HK = 12;                                      // K-tile bodies that carry the parked tile's stores (K >= 768)

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
};

typedef __attribute__((ext_vector_type(4))) int int4_t;

template <int IMM>
__device__ __forceinline__ bf16x8 lds_b128(uint32_t addr) {
    int4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return __builtin_bit_cast(bf16x8, v);
}
template <int IMM>
__device__ __forceinline__ f32x4 lds_f128(uint32_t addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return v;
}
__device__ __forceinline__ void lds_w128(uint32_t addr, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

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

struct Cur {                   // position of the DMA stream: one K-tile of this block's item list (wave-uniform)
    int j, kt, tm, tn;         // item index in the block's list, K-tile, the item's tile coordinates
    const char* p[4];          // A0, A1, B0, B1 half-tile sources: row 8 wave of the half, this K-tile's k0
};

// the block's items are first_item + j * stride in row-major tile order: (tm, tn) advances by (dm, dn) with a carry - no division, no branch
struct Walk { int dm, dn; };

__device__ __forceinline__ void seat(const Args& a, int wave, Cur& c) {
    const int64_t m0 = (int64_t)c.tm * BM, n0 = (int64_t)c.tn * BN;
    c.p[0] = reinterpret_cast<const char*>(a.A + (m0 + 8 * wave) * a.lda);
    c.p[1] = reinterpret_cast<const char*>(a.A + (m0 + 128 + 8 * wave) * a.lda);
    c.p[2] = reinterpret_cast<const char*>(a.B + (n0 + 8 * wave) * a.ldb);
    c.p[3] = reinterpret_cast<const char*>(a.B + (n0 + 128 + 8 * wave) * a.ldb);
}

__device__ __forceinline__ void next_tile(const Args& a, const Walk& w, int& tm, int& tn) {
    int n = tn + w.dn;
    const int carry = n >= a.tiles_n ? 1 : 0;
    tn = n - (carry ? a.tiles_n : 0);
    tm = tm + w.dm + carry;
}

// Branch-free (selects only): a branch inside a K-tile body ends its basic block, and hipcc then sinks MFMAs below the fragment reads and
// DMA issues they were interleaved with (profiles/r04_fp8_kstep_schedules.txt).  Past the end of the stream the last K-tile is fetched
// again (never read; keeps the vmcnt arithmetic uniform).
__device__ __forceinline__ void advance(const Args& a, const Walk& w, int n_mine, int wave, Cur& c) {
    const bool in_item = c.kt + 1 < a.nkt;
    const bool more = c.j + 1 < n_mine;
    Cur nx = c;
    next_tile(a, w, nx.tm, nx.tn);
    seat(a, wave, nx);
    const bool hop = !in_item && more;
#pragma unroll
    for (int h = 0; h < 4; ++h) c.p[h] = hop ? nx.p[h] : (in_item ? c.p[h] + 2 * BK : c.p[h]);
    c.tm = hop ? nx.tm : c.tm;
    c.tn = hop ? nx.tn : c.tn;
    c.j += hop ? 1 : 0;
    c.kt = in_item ? c.kt + 1 : (more ? 0 : c.kt);
}

// DMA piece n (0..15 = half * 4 + quarter) of a K-tile; MFMA index m (0..127) of the K-tile body behind which it is issued:
// pieces 0 .. N4_LATE - 1 of K-tile t + 2 behind the barrier of body t (m = 113 + ..), the rest during body t + 1
__host__ __device__ constexpr int late_piece_at(int m) {
    return (m >= 113 && (m - 113) % N4_LATE_STEP == 0 && (m - 113) / N4_LATE_STEP < N4_LATE) ? (m - 113) / N4_LATE_STEP : -1;
}
__host__ __device__ constexpr int early_piece_at(int m) {
    return (m >= 2 && (m - 2) % N4_DMA_STEP == 0 && (m - 2) / N4_DMA_STEP < 16 - N4_LATE) ? N4_LATE + (m - 2) / N4_DMA_STEP : -1;
}
constexpr int LAST_EARLY = 2 + (15 - N4_LATE) * N4_DMA_STEP;          // MFMA index of the last early piece
constexpr int STORE_FIRST = 84, STORE_STEP = 7;                        // head bodies: store s of the body behind MFMA 84 + 7 s (s < 4)
static_assert(LAST_EARLY < STORE_FIRST && STORE_FIRST + 3 * STORE_STEP < 112 && 113 + (N4_LATE - 1) * N4_LATE_STEP < 128, "schedule");
// the 32 stores of a parked tile over the HK head bodies
constexpr int NPARK = N4_PARK;
__host__ __device__ constexpr int store_begin(int kt) { return kt * (NPARK / HK) + (kt < NPARK % HK ? kt : NPARK % HK); }
__host__ __device__ constexpr int store_count(int kt) { return NPARK / HK + (kt < NPARK % HK ? 1 : 0); }
static_assert(store_begin(HK) == NPARK && NPARK / HK + 1 <= 4 && NPARK % 4 == 0 && NPARK <= 32, "stores per body");

template <int EPI, int FLAGS>
__global__ __launch_bounds__(NTH) void gemm256n4_kernel(Args a) {
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
    // staging: lane -> row 8 wave + (lane >> 3) of a 32-row piece, physical chunk lane & 7; the logical chunk is the same in all four
    // pieces of a half because (row >> 1) & 7 does not change when the row grows by 32
    const uint32_t st_lc = (uint32_t)((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7));
    const uint32_t voffA = (uint32_t)(lane >> 3) * (uint32_t)a.lda * 2u + st_lc * 16u;
    const uint32_t voffB = (uint32_t)(lane >> 3) * (uint32_t)a.ldb * 2u + st_lc * 16u;
    const uint32_t strideA32 = (uint32_t)a.lda * 64u, strideB32 = (uint32_t)a.ldb * 64u;        // 32 rows, bytes
    const uint32_t wave_lds = smem_off + (uint32_t)wave * 1024u;
#define ISSUE_PIECE(N, c, slot_off)                                                                                                    \
    do {                                                                                                                               \
        const char* src_ = ((c).p[(N) >> 2] + (uint64_t)((N) & 3) * (((N) >> 2) < 2 ? strideA32 : strideB32)) + (((N) >> 2) < 2 ? voffA : voffB); \
        __builtin_amdgcn_global_load_lds(GLB_PTR(src_),                                                                                \
                                         (__attribute__((address_space(3))) void*)(size_t)(wave_lds + (slot_off) + ((N) >> 2) * HALF + ((N) & 3) * 4096), \
                                         16, 0, 0);                                                                                    \
    } while (0)

    Walk walk;
    walk.dm = stride / a.tiles_n;
    walk.dn = stride - walk.dm * a.tiles_n;
    Cur cD;                          // the DMA stream
    cD.j = 0; cD.kt = 0;
    cD.tm = first_item / a.tiles_n;
    cD.tn = first_item - cD.tm * a.tiles_n;
    int etm = cD.tm, etn = cD.tn;    // the tile of the item whose MFMAs are running
    seat(a, wave, cD);
    // prologue: all of K-tile 0, the first N4_LATE pieces of K-tile 1
    static_for<0, 16>([&](auto pc) { ISSUE_PIECE(decltype(pc)::value, cD, 0u); });
    advance(a, walk, n_mine, wave, cD);
    static_for<0, N4_LATE>([&](auto pc) { ISSUE_PIECE(decltype(pc)::value, cD, (uint32_t)SLOT); });
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N4_LATE) : "memory");
    SEG_BARRIER();

    // fragment addresses: row r = lane & 15 of a 16-row tile, logical chunk 4 ks + (lane >> 4)
    const int fr = lane & 15, fx = (fr >> 1) & 7;
    uint32_t laA[2], laB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const uint32_t o = (uint32_t)fr * 128u + (uint32_t)(((4 * ks + (lane >> 4)) ^ fx) << 4);
        laA[ks] = smem_off + (uint32_t)wr * HALF + o;
        laB[ks] = smem_off + (uint32_t)(2 + wc) * HALF + o;
    }

    bf16x8 fa[4], fb[8];            // A fragments: rotating slots (group g uses slot g & 3); B fragments of the running stage
    // the order of the steady state: fa(0), fa(1), fb[0..7] of the first stage
    fa[0] = lds_b128<0>(laA[0]);
    fa[1] = lds_b128<2048>(laA[0]);
    static_for<0, 8>([&](auto nc) { constexpr int ni = decltype(nc)::value; fb[ni] = lds_b128<ni * 2048>(laB[0]); });

    f32x4 acc[64];                   // [mi * 8 + ni]
    bf16x8 parked[NPARK];            // the finished tile of the previous item, its LAST NPARK vectors: vector k = row 16 (k >> 2) + 4 (k & 3) + (lane >> 4),
                                     // columns 8 (lane & 15) .. + 7; parked[k - (32 - NPARK)]
#pragma unroll
    for (int k = 0; k < NPARK; ++k) parked[k] = bf16x8{};
    // this lane's address of the next parked vector to leave (before the first epilogue: the first item's own tile, see the item loop)
    char* dst_ptr = reinterpret_cast<char*>(a.C + ((int64_t)etm * BM + wr * 128 + (lane >> 4) + 4 * (32 - NPARK)) * a.ldc + (int64_t)etn * BN + wc * 128 + (lane & 15) * 8);
    const int64_t row4_bytes = a.ldc * 8;                // 4 rows of C
    int t = 0;                       // K-tile counter of the whole stream (slot = t & 1)

#define LGKM_WAIT(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
    // a wait that the compiler must treat as a definition of the fragments it guards (an MFMA is register-only: "memory" does not order it)
#define LGKM_X_ ((N4_EXP & 8) ? 15 : 0)      // diagnostic 8: no fragment wait ever blocks (wrong results): what the exposed LDS latency costs
#define LGKM_WAIT_A(N, slot) do { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fa[slot]) : "n"((N) | LGKM_X_)); __builtin_amdgcn_sched_barrier(0); } while (0)
#define LGKM_WAIT_AB(N, slot, ni) do { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[slot]), "+v"(fb[ni]) : "n"((N) | LGKM_X_)); __builtin_amdgcn_sched_barrier(0); } while (0)
#define LGKM_WAIT_B(N, ni) do { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fb[ni]) : "n"((N) | LGKM_X_)); __builtin_amdgcn_sched_barrier(0); } while (0)

    // One K-tile: stage E (k-step 0) and stage O (k-step 1) of 64 MFMAs.  FIRST: the accumulators start from zero.  SB / SC: the parked
    // vectors [SB, SB + SC) leave behind MFMAs 84, 91, 98, 105 (SC <= 4).
    auto body = [&](auto first_tag, auto sb_tag, auto sc_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int SB = decltype(sb_tag)::value, SC = decltype(sc_tag)::value;
        const uint32_t sc = (uint32_t)(t & 1) * SLOT, sn = SLOT - sc;
        // read bases: this slot's k-step 0 / 1, the other slot's k-step 0
        uint32_t rA0 = laA[0] + sc, rB1 = laB[1] + sc, rA1 = laA[1] + sc, rAn = laA[0] + sn, rBn = laB[0] + sn;
        asm volatile("" : "+v"(rA0), "+v"(rA1), "+v"(rB1), "+v"(rAn), "+v"(rBn));
        // register files pinned at every K-tile boundary: the accumulators in the accumulator file, the parked tile in VGPRs.  Left to itself
        // (460 of 512 registers live) hipcc's allocator migrates accumulators through spare VGPRs inside the head bodies: v_accvgpr_write /
        // read quadruples with s_nop 7 in front of MFMAs
#if N4_PIN
        if constexpr (!FIRST) {
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("" : "+a"(acc[i]));
        }
#pragma unroll
        for (int k = SB; k < NPARK; ++k) asm volatile("" : "+v"(parked[k]));
#endif
        static_for<0, 128>([&](auto mc) {
            constexpr int m = decltype(mc)::value, S = m >> 6, g = (m >> 3) & 7, ni = m & 7;
            // ---- in front of MFMA m
            if constexpr (m == 112) {              // the barrier of the K-tile: every read of this slot has returned, K-tile t + 1 has landed
                LGKM_WAIT(0);
                if constexpr (!(N4_EXP & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SC) : "memory");
                if constexpr (!(N4_EXP & 4)) SEG_BARRIER();
            }
            if constexpr (g == 0) {
                if constexpr (ni == 0) LGKM_WAIT_AB(7, 0, 0);
                else LGKM_WAIT_B(8 - ni, ni);
            } else if constexpr (ni == 0 && m != 112) {
                LGKM_WAIT_A(1, g & 3);
            } else if constexpr (m == 112) {
                asm volatile("" : "+v"(fa[g & 3]));
                __builtin_amdgcn_sched_barrier(0);
            }
            // asm with the accumulator constrained to the accumulator file: with the builtin (operand class "either file") hipcc's allocator, 460 of
            // 512 registers live, parks accumulators in whatever VGPRs the leaving stores free and shuttles them back through v_accvgpr_write /
            // read quadruples with s_nop 7 around the MFMAs of the head bodies (seen in the listing: 524 writes, 131 s_nop 7)
            if constexpr (FIRST && S == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[g * 8 + ni]) : "v"(fb[ni]), "v"(fa[g & 3]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[g * 8 + ni]) : "v"(fb[ni]), "v"(fa[g & 3]));
            __builtin_amdgcn_sched_barrier(0);
            // ---- behind MFMA m
            if constexpr (ni == 0) {               // the A fragment of group g + 2
                constexpr int g2 = (g + 2) & 7;
                if constexpr (g < 6) fa[(g + 2) & 3] = lds_b128<g2 * 2048>(S == 0 ? rA0 : rA1);
                else fa[(g + 2) & 3] = lds_b128<g2 * 2048>(S == 0 ? rA1 : rAn);
            }
            if constexpr (g == 7) fb[ni] = lds_b128<ni * 2048>(S == 0 ? rB1 : rBn);          // the next stage's B fragment ni, behind its last use
            if constexpr (early_piece_at(m) >= 0 && !(N4_EXP & 2)) ISSUE_PIECE(early_piece_at(m), cD, sn);
            if constexpr (m == LAST_EARLY + 1) advance(a, walk, n_mine, wave, cD);
            if constexpr (late_piece_at(m) >= 0 && !(N4_EXP & 2)) ISSUE_PIECE(late_piece_at(m), cD, sc);
            if constexpr (SC > 0 && m >= STORE_FIRST && (m - STORE_FIRST) % STORE_STEP == 0 && (m - STORE_FIRST) / STORE_STEP < SC) {
                constexpr int k = SB + (m - STORE_FIRST) / STORE_STEP;
                if constexpr ((N4_EXP & 128) != 0) asm volatile("" ::"v"(parked[k]));
                else if constexpr (N4_NT != 0) __builtin_nontemporal_store(parked[k], reinterpret_cast<bf16x8*>(dst_ptr));
                else *reinterpret_cast<bf16x8*>(dst_ptr) = parked[k];
                dst_ptr += row4_bytes;              // vector k + 1: 4 rows further (the strips are contiguous: 16 = 4 x 4)
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        ++t;
    };
    constexpr std::integral_constant<int, 0> I0{};

    for (int j = 0; j < n_mine; ++j) {
        // the previous item's tile leaves under this item's first HK K-tiles.  (One path for every item - an if / else around the head put
        // phi copies of the 256 accumulators into VGPRs: for the first item the parked registers hold zeros and go to this item's OWN tile,
        // which the same lanes overwrite with the results later: same thread, same address, program order.)
        static_for<0, HK>([&](auto kc) {
            constexpr int kt = decltype(kc)::value;
            body(std::integral_constant<bool, kt == 0>{}, std::integral_constant<int, store_begin(kt)>{}, std::integral_constant<int, store_count(kt)>{});
        });
        for (int kt = HK; kt < a.nkt; ++kt) body(std::false_type{}, I0, I0);
        // (the MFMAs are asm: the compiler pads nothing behind them; the last accumulator was written 16 cycles ago, its first reader is far)
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");

        // ---- epilogue of this item: transpose, arithmetic, park
        if constexpr ((N4_EXP & 16) != 0) {
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("" ::"v"(acc[i]));
        } else {
            const int64_t m0 = (int64_t)etm * BM, n0 = (int64_t)etn * BN;
            next_tile(a, walk, etm, etn);
            const uint32_t stg = smem_off + 2 * SLOT + (uint32_t)wave * STG_WAVE;
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            // write side: accumulator row 16 mi + (lane & 15), 16-byte chunk 4 ni + (lane >> 4) of the strip's 128 columns
            const int wrow = lane_e & 15, wq = lane_e >> 4;
            const uint32_t wbase = stg + (uint32_t)wrow * 512u;
            // read side: 16 consecutive lanes on ONE row (256 contiguous bytes of C, 4 rows per store instruction):
            // lane -> row 4 q + (lane >> 4) of the strip (q = 0..3), columns 8 (lane & 15) .. + 7
            const int rq = lane_e >> 4, rp = lane_e & 15;
            const int64_t ncol = n0 + wc * 128 + rp * 8;
            const int64_t mrow = m0 + wr * 128 + rq;
            dst_ptr = reinterpret_cast<char*>(a.C + (mrow + 4 * (32 - NPARK)) * a.ldc + ncol);
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (a.bias) { b0 = load4(a.bias + ncol); b1 = load4(a.bias + ncol + 4); }
            constexpr bool NEED_AUX = EPI == LECCR_EPI_DQUICKGELU || EPI == LECCR_EPI_DGELU;
            constexpr bool PRE = NEED_AUX || (FLAGS & F_RES) != 0;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                // the saved pre-activation (derivative epilogues) / the residual of this strip: under the transposes
                [[maybe_unused]] bf16x8 xpre[4];
                if constexpr (PRE) {
                    const bf16_t* pb = NEED_AUX ? a.aux : a.residual;
                    const int64_t pld = NEED_AUX ? a.ldaux : a.ldr;
#pragma unroll
                    for (int h8 = 0; h8 < 4; ++h8) xpre[h8] = *reinterpret_cast<const bf16x8*>(pb + (mrow + mi * 16 + 4 * h8) * pld + ncol);
                }
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) lds_w128(wbase + (uint32_t)(((4 * ni + wq) ^ wrow) << 4), acc[mi * 8 + ni]);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {                               // two rounds of two rows: 16 registers of transposed values at a time
                    f32x4 v[4];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {                              // row 4 (2 hh + q) + rq, chunks 2 rp and 2 rp + 1
                        const int row = 4 * (2 * hh + q) + rq;
                        v[2 * q] = lds_f128<0>(stg + (uint32_t)row * 512u + (uint32_t)(((2 * rp) ^ row) << 4));
                        v[2 * q + 1] = lds_f128<0>(stg + (uint32_t)row * 512u + (uint32_t)(((2 * rp + 1) ^ row) << 4));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int h8 = 2 * hh + q;
                        f32x4 v0 = v[2 * q] * a.alpha + b0, v1 = v[2 * q + 1] * a.alpha + b1;
                        const int64_t m = mrow + mi * 16 + 4 * h8;
                        if constexpr ((FLAGS & F_AUXOUT) != 0) {               // the saved pre-activation is stored at once (a second tile would not fit the registers)
                            bf16x8 pv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { pv[e] = (bf16_t)v0[e]; pv[4 + e] = (bf16_t)v1[e]; }
                            if constexpr ((N4_EXP & 128) != 0) asm volatile("" ::"v"(pv));
                            else *reinterpret_cast<bf16x8*>(a.aux_out + m * a.ldc + ncol) = pv;
                        }
                        if constexpr (EPI != LECCR_EPI_NONE) {
                            f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
                            if constexpr (NEED_AUX) {
                                const bf16x8 xv = xpre[h8];
#pragma unroll
                                for (int e = 0; e < 4; ++e) { x0[e] = (float)xv[e]; x1[e] = (float)xv[4 + e]; }
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v0[e] = apply_epi<EPI>(v0[e], x0[e]); v1[e] = apply_epi<EPI>(v1[e], x1[e]); }
                        }
                        if constexpr ((FLAGS & F_RES) != 0) {
                            bf16x8 rv;
                            if constexpr (!NEED_AUX) rv = xpre[h8];
                            else rv = *reinterpret_cast<const bf16x8*>(a.residual + m * a.ldr + ncol);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v0[e] += (float)rv[e]; v1[e] += (float)rv[4 + e]; }
                        }
                        bf16x8 ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { ov[e] = (bf16_t)v0[e]; ov[4 + e] = (bf16_t)v1[e]; }
                        if (mi * 4 + h8 < 32 - NPARK) {              // the first vectors leave at once
                            if constexpr ((N4_EXP & 128) != 0) asm volatile("" ::"v"(ov));
                            else *reinterpret_cast<bf16x8*>(a.C + m * a.ldc + ncol) = ov;
                        } else {
                            parked[mi * 4 + h8 - (32 - NPARK)] = ov;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // the last item's tile
    if constexpr (!(N4_EXP & (16 | 128))) {
#pragma unroll
        for (int k = 0; k < NPARK; ++k) *reinterpret_cast<bf16x8*>(dst_ptr + (int64_t)k * row4_bytes) = parked[k];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-fetched tail K-tiles still target this block's LDS
}

template <int EPI, int FLAGS>
static void launch_inst(const Args& a, int grid, hipStream_t st) {
    auto kfn = gemm256n4_kernel<EPI, FLAGS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NTH), LDS_BYTES, st, a);
}

}  // namespace g5

static int n4_flags(const leccr_gemm_params& p) { return (p.residual ? g5::F_RES : 0) | (p.aux_out ? g5::F_AUXOUT : 0); }

// whole tiles only, K >= 768 (the step's shapes: M = 256 x tokens, N = 768 .. 4096, K = 768 .. 4096): everything else stays on the 8-wave kernel
bool gemm256n4_legal(const leccr_gemm_params& p) {
    if (p.trans_a != 0 || p.trans_b != 0 || p.c_dtype != LECCR_BF16 || p.ab_dtype != LECCR_BF16 || p.accumulate || p.c3 || p.c8 || p.colsum) return false;
    if (p.M < 256 || p.N < 256 || (p.M & 255) || (p.N & 255) || (p.K & 63) || p.K < g5::HK * g5::BK) return false;
    if ((p.ldc & 7) || (p.ldr & 7) || (p.ldaux & 7) || (p.lda & 7) || (p.ldb & 7)) return false;
    if ((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15)) return false;
    if (p.lda >= (1 << 22) || p.ldb >= (1 << 22)) return false;      // 32-bit (row * ld * 2) in the staging offsets
    const int f = n4_flags(p);
    switch (p.epilogue) {
        case LECCR_EPI_NONE: return f == 0 || f == g5::F_RES;
        case LECCR_EPI_QUICKGELU:
        case LECCR_EPI_GELU: return f == g5::F_AUXOUT;
        case LECCR_EPI_DQUICKGELU:
        case LECCR_EPI_DGELU: return f == 0;
    }
    return false;
}

int gemm256n4_launch(const leccr_gemm_params& p, int cap, hipStream_t st) {
    using namespace g5;
    Args a{};
    a.A = (const bf16_t*)p.A; a.B = (const bf16_t*)p.B; a.C = (bf16_t*)p.C;
    a.M = p.M; a.N = p.N; a.K = p.K; a.lda = p.lda; a.ldb = p.ldb; a.ldc = p.ldc;
    a.bias = p.bias; a.aux = (const bf16_t*)p.aux; a.ldaux = p.ldaux; a.aux_out = (bf16_t*)p.aux_out;
    a.residual = (const bf16_t*)p.residual; a.ldr = p.ldr; a.alpha = p.alpha;
    a.tiles_m = (int)(p.M / BM); a.tiles_n = (int)(p.N / BN);
    a.nkt = (int)(p.K / BK);
    a.n_items = a.tiles_m * a.tiles_n;
    const int grid = a.n_items < cap ? a.n_items : cap;
    const int f = n4_flags(p);
#ifdef N4_QUICK            // one instantiation: compile-time of the schedule experiments
    launch_inst<LECCR_EPI_NONE, 0>(a, grid, st);
    return check_launch("gemm256n4");
#else
    switch (p.epilogue) {
        case LECCR_EPI_NONE:
            if (f == F_RES) launch_inst<LECCR_EPI_NONE, F_RES>(a, grid, st);
            else launch_inst<LECCR_EPI_NONE, 0>(a, grid, st);
            break;
        case LECCR_EPI_QUICKGELU: launch_inst<LECCR_EPI_QUICKGELU, F_AUXOUT>(a, grid, st); break;
        case LECCR_EPI_GELU: launch_inst<LECCR_EPI_GELU, F_AUXOUT>(a, grid, st); break;
        case LECCR_EPI_DQUICKGELU: launch_inst<LECCR_EPI_DQUICKGELU, 0>(a, grid, st); break;
        case LECCR_EPI_DGELU: launch_inst<LECCR_EPI_DGELU, 0>(a, grid, st); break;
        default: set_error("gemm256n4: bad epilogue %d", p.epilogue); return LECCR_EINVAL;
    }
    return check_launch("gemm256n4");
#endif
}

}  // namespace leccr
