// gemm256.hip - the large-shape bf16 GEMM of the LECCR step: 256x256x64 block tile, 8 waves, persistent, ping-pong.
//
// Why (measured on MI355X, random data): the 128x128 kernel in gemm.hip reaches ~940 TFLOP/s at 4096^3 but only
// ~600 on the step's dominant shapes (M = 147712 tokens, K = 768) and 350-550 on the weight-gradient products: at
// 64 FLOP per L2->LDS byte and one prologue + epilogue per 12 K-tiles it is fill/drain- and L2-traffic bound.
// This kernel doubles the tile (128 FLOP/B), keeps a continuous operand stream across output tiles (persistent
// blocks, the next tile's first K-tiles are in flight during the epilogue) and schedules the two waves that share a
// SIMD in opposite phases.
//
// Structure
//   * 512 threads = 8 waves = 2 (M) x 4 (N); wave tile 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16
//     (128 accumulator VGPRs), issued with swapped roles so a lane ends with 4 consecutive N of one row.
//   * LDS 128 KiB = 2 slots x {A0, A1, B0, B1} half-tiles of 16 KiB (128 rows x 64 k), filled by
//     global_load_lds_dwordx4 with the XOR swizzle applied to the SOURCE address (gemm.hip's images).
//   * A K-tile is 4 quadrant steps ("quads") of 16 MFMAs per wave.  Two schedules:
//     NT (forward / dX, TRANS = 0) - software pipelined: every LDS fragment read and every DMA piece is issued between
//       two MFMAs of the same wave (one per MFMA at most), a fragment register is reloaded right after the last MFMA
//       that needs its old contents has issued, the 8 pieces of K-tile t+2 go out as soon as slot t is free, and there
//       are two barriers per K-tile ("slot t free", "K-tile t+1 visible").  See the comment at the loop.
//     TN (weight gradients, TRANS = 1) - LOAD (ds_reads + a share of the prefetch) and COMPUTE (16 MFMAs) segments;
//       waves 0-3 (M row 0) and 4-7 (M row 1) pair up on the 4 SIMDs and run ONE SEGMENT APART, every segment ends in a
//       raw s_barrier; rolling prefetch TWO K-tiles ahead with one counted `s_waitcnt vmcnt(8)` per K-tile.  (The
//       pipelined loop needs 96 fragment registers; with the transposed reads' register pairs it spills.)
//     In-kernel stamps: the alternating schedule runs ~4200 cycles per K-tile (LOAD segments 2x the COMPUTE segments),
//     the pipelined one ~3800 at 4096^3, against the 2048-cycle MFMA floor.
//   * The K-tile stream runs across the block's output tiles ("items"), so the epilogue of one item overlaps the
//     partner wave's last COMPUTE and the loads of the next item.
//
// TRANS = 0: C = A[M,K] B[N,K]^T (forward and dX).   TRANS = 1: C += A[K,M]^T B[K,N] (weight gradients, split-K with
// f32 atomics).  Same epilogues as gemm.hip.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// The output tiles of the forward / dX products leave by NON-TEMPORAL stores (round 5): a launch writes 0.2 - 0.9 GB that nothing re-reads before
// it has left every cache, and written with the default policy the stream evicts the weight panels and A rows the K loops are re-reading from
// the 4 MiB L2 of each XCD (scripts/r5_n4_e.sh: the 4-wave kernel's fc1 launch 745 -> 680 us, out 205 -> 193).  G256_NT=0 builds the old stores.
#ifndef G256_NT
#define G256_NT 1
#endif
// (a compile-time choice on purpose: behind a runtime switch - `if (a.nt) nontemporal else plain` - hipcc merges the two stores into ONE plain
// store and the policy is lost without a word: scripts/r5_nt_ab2.sh, qkv 603 - 630 us with the switch, 570 - 579 without it)
#if G256_NT
#define G256_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define G256_STORE(ptr, val) (*(ptr) = (val))
#endif

namespace leccr {
namespace g256 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 16384;            // bytes per half-tile
constexpr int SLOT = 4 * HALF;         // A0 A1 B0 B1
constexpr int NTHREADS = 512;
constexpr int LDS_BYTES = 2 * SLOT + 8 * (16 * 64 * 4);     // operand slots + per-wave epilogue staging (16 rows x 64 f32): 160 KiB

struct Args {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    int64_t M, N, K, lda, ldb, ldc;
    const float* bias;
    const void* aux;
    int64_t ldaux;
    void* aux_out;
    const void* residual;
    int64_t ldr;
    float alpha;
    int accumulate;
    int tiles_m, tiles_n, splits, nkt;   // nkt = K-tiles per item (uniform; reduction tail reads zeros)
    int band_rows, col_group;            // NT item order (decode): tile rows per XCD band, tile columns per group; col_group 0 = row-major
    int n_items;
    unsigned long long* stamps;          // diagnostic builds only (LECCR_G256_STAMPS): per-segment cycle sums
    float* asum;                         // TRANS = 1 with F_ASUM: asum[m] += alpha * sum_k A[k, m] (the bias gradient), or null
    bf16_t* c3;                          // f32 C only: the (hi, lo) split copy of C, [M, 3 N] in form c3_form (split.hip), or null; C may then be null
    int c3_form;
};

__device__ __attribute__((aligned(256))) unsigned char g_zero_page256[512];

__device__ __forceinline__ int tn_swz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Position of one K-tile of this block's stream.  Everything is wave-uniform (SGPRs).  The four half-tile source
// pointers are advanced incrementally (one 64-bit add per half per K-tile); integer divisions happen once per item.
struct Cursor {
    int j, kt;                 // item ordinal within this block's list, K-tile within the item
    int split;
    int64_t m0, n0, k0;        // k0 = first reduction index of this K-tile
    const bf16_t* p[4];        // source of half A0, A1, B0, B1 at this K-tile (row 0 / column 0 of the half)
    bool in[4];                // the half is interior in M/N (no row/column tail)
};

template <int TRANS>
__device__ __forceinline__ void decode(const Args& a, int first_item, int stride, Cursor& c) {
    const int item = first_item + c.j * stride;
    const int tiles = a.tiles_m * a.tiles_n;
    c.split = item / tiles;
    const int tile = item - c.split * tiles;
    int tm, tn;
    if (TRANS == 0 && a.col_group > 0) {
        // Column-grouped order (forward / dX with many tile columns).  The tile rows are cut into 8 bands (one per XCD, as the
        // XCD-contiguous item ranges are); inside a band the tiles are walked one GROUP of `col_group` tile columns at a time,
        // row by row.  The XCD's 32 workgroups then share col_group weight panels (<= 1.6 MB: they stay in its 4 MiB L2 for the
        // whole group) and stream the band's activation panels past them, once per group.  In plain row-major order a round of 32
        // workgroups touches EVERY weight panel (3.5 - 4.7 MB for N = 2304 / 3072 at K = 768) next to its own rows and the
        // streaming output, so each XCD re-fetches the whole weight matrix from the fabric every round: measured 1.33 / 2.16 GB
        // read per launch against 0.23 GB algorithmic (profiles/r02_gemm256_traffic_by_shape.txt).  Opt-in: see gemm256_launch.
        const int per_band = a.band_rows * a.tiles_n;
        int band = tile / per_band;
        const int nb = (a.tiles_m + a.band_rows - 1) / a.band_rows;
        if (band > nb - 1) band = nb - 1;
        const int within = tile - band * per_band;
        const int rows_b = min(a.band_rows, a.tiles_m - band * a.band_rows);
        const int full = a.tiles_n / a.col_group;                      // full-width groups
        int g = within / (rows_b * a.col_group);
        if (g > full) g = full;
        const int cw = g < full ? a.col_group : a.tiles_n - full * a.col_group;
        const int wg = within - g * rows_b * a.col_group;
        const int row = wg / cw;
        tm = band * a.band_rows + row;
        tn = g * a.col_group + (wg - row * cw);
    } else {
        tm = tile / a.tiles_n;
        tn = tile - tm * a.tiles_n;
    }
    c.m0 = (int64_t)tm * BM;
    c.n0 = (int64_t)tn * BN;
    c.k0 = (int64_t)c.split * a.nkt * BK + (int64_t)c.kt * BK;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const bool isA = w < 2;
        const int64_t r0 = (isA ? c.m0 : c.n0) + (w & 1) * 128;
        const int64_t ld = isA ? a.lda : a.ldb;
        const bf16_t* base = isA ? a.A : a.B;
        c.in[w] = r0 + 128 <= (isA ? a.M : a.N);
        c.p[w] = TRANS == 0 ? base + r0 * ld + c.k0 : base + c.k0 * ld + r0;
    }
}

// next K-tile of the stream; at the end it stays on the last one (harmless reloads keep the vmcnt counts uniform).
// FIRST/LAST select which half pointers this cursor actually uses (A prefetch: 0-1, B prefetch: 2-3).
template <int TRANS, int FIRST, int LAST>
__device__ __forceinline__ void advance(const Args& a, int first_item, int stride, int n_mine, Cursor& c) {
    if (c.kt + 1 < a.nkt) {
        ++c.kt;
        c.k0 += BK;
#pragma unroll
        for (int w = FIRST; w <= LAST; ++w) c.p[w] += TRANS == 0 ? (int64_t)BK : (int64_t)BK * (w < 2 ? a.lda : a.ldb);
    } else if (c.j + 1 < n_mine) {
        ++c.j;
        c.kt = 0;
        decode<TRANS>(a, first_item, stride, c);
    }
}

// Per-thread, loop-invariant parts of the staging addresses (byte offsets inside a half-tile's source window).
struct StageOff {
    uint32_t a[2], b[2];     // byte offset of this thread's 16-byte chunk i = 0,1 for the A / B operand
    int row[2];              // NT: row inside the half (for the M/N tail clamp);  TN: k inside the K-tile
    int col8[2];             // element column of the chunk (NT: k offset; TN: column inside the half)
};

template <int TRANS>
__device__ __forceinline__ StageOff make_stage_off(const Args& a) {
    StageOff o;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = i * NTHREADS + (int)threadIdx.x;
        if constexpr (TRANS == 0) {
            const int row = s >> 3, pc = s & 7, lc = pc ^ (row & 7);
            o.row[i] = row;
            o.col8[i] = lc * 8;
            o.a[i] = (uint32_t)(row * a.lda + lc * 8) * 2u;
            o.b[i] = (uint32_t)(row * a.ldb + lc * 8) * 2u;
        } else {
            const int k = s >> 4, pc = s & 15;
            const int lc = (((pc >> 1) ^ tn_swz(k)) << 1) | (pc & 1);
            o.row[i] = k;
            o.col8[i] = lc * 8;
            o.a[i] = (uint32_t)(k * a.lda + lc * 8) * 2u;
            o.b[i] = (uint32_t)(k * a.ldb + lc * 8) * 2u;
        }
    }
    return o;
}

// Issue one half-tile (2 x 16 B per thread).  which: 0 = A rows/cols 0-127, 1 = A 128-255, 2 = B 0-127, 3 = B 128-255.
// Interior halves (the common case): uniform 64-bit base (SGPR pair) + 32-bit per-thread byte offset, nothing else.
template <int TRANS, int WHICH, int I0 = 0, int I1 = 2>
__device__ __forceinline__ void issue_half(const Args& a, const StageOff& so, const Cursor& c, uint32_t lds_slot_off,
                                           uint32_t wave_off) {
    constexpr bool isA = WHICH < 2;
    const bool interior = c.in[WHICH] && (TRANS == 0 || c.k0 + BK <= a.K);
    const char* ub = reinterpret_cast<const char*>(c.p[WHICH]);
    const uint32_t dst0 = lds_slot_off + WHICH * HALF + wave_off;       // uniform; the hardware adds lane*16
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const void* src;
        if (interior) {
            src = ub + (isA ? so.a[i] : so.b[i]);
        } else {
            const bf16_t* base = isA ? a.A : a.B;
            const int64_t ld = isA ? a.lda : a.ldb;
            const int64_t rmax = isA ? a.M : a.N;
            const int64_t r0 = (isA ? c.m0 : c.n0) + (WHICH & 1) * 128;
            if constexpr (TRANS == 0) {
                int64_t r = r0 + so.row[i];
                r = r < rmax ? r : rmax - 1;
                src = base + r * ld + c.k0 + so.col8[i];
            } else {
                const int64_t kk = c.k0 + so.row[i], col = r0 + so.col8[i];
                src = (kk < a.K && col < rmax) ? (const void*)(base + kk * ld + col) : (const void*)g_zero_page256;
            }
        }
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), (__attribute__((address_space(3))) void*)(size_t)(dst0 + i * NTHREADS * 16),
                                         16, 0, 0);
    }
}

// ---- LDS fragment reads through inline asm.
// hipcc cannot prove that a ds_read does not alias an in-flight global_load_lds, so a C++ LDS load anywhere in the loop
// makes SIInsertWaitcnts put `s_waitcnt vmcnt(0)` at the loop head - draining the prefetch every K-tile (seen in the
// .s).  Reads issued from asm are invisible to that pass; their completion is the explicit WAIT_LDS() before the
// segment barrier, and SEG_BARRIER's sched_barrier keeps the MFMAs behind it.
typedef __attribute__((ext_vector_type(4))) int int4_t;
typedef __attribute__((ext_vector_type(2))) int int2_t;

template <int IMM>
__device__ __forceinline__ bf16x8 lds_b128(uint32_t addr) {
    int4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return __builtin_bit_cast(bf16x8, v);
}

template <int IMM>
__device__ __forceinline__ bf16x8 lds_tr16x2(uint32_t addr) {      // rows k..k+3 (offset IMM) and k+4..k+7 (IMM + 1024)
    int2_t lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(IMM));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(IMM + 1024));
    int4_t v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}

// Per-lane address pieces.  NT: va[ks] = row*128 + ((4ks + h) ^ (row & 7))*16 with row = lane & 15 (fragment rows are
// multiples of 16 apart, so row & 7 == lane & 7 and the fragment row is a pure immediate: + row0*128).
// TN: base = k*256 + 8*(i & 3), k = 8h + (i >> 2); x = tn_swz(k)*32; a fragment whose 16 columns start at 16*cb is read
// at ((32*cb) ^ x) + base (+1024: k+4, +8192: k-step 1).
struct LaneAddr {
    uint32_t nt[2];
    uint32_t tbase, tx;
};

__device__ __forceinline__ LaneAddr make_lane_addr(int lane) {
    LaneAddr la;
    const int r = lane & 15, h = lane >> 4;
    la.nt[0] = r * 128 + (((0 + h) ^ (r & 7)) << 4);
    la.nt[1] = r * 128 + (((4 + h) ^ (r & 7)) << 4);
    const int k = 8 * h + (r >> 2);
    la.tbase = k * 256 + 8 * (r & 3);
    la.tx = tn_swz(k) * 32;
    return la;
}

// fragment (k-step KS) of the 16 rows/columns starting at ROW0 (compile time) of the half-tile at LDS byte `hoff`
// (uniform) [+ runtime, wave-uniform extra rows folded in by the caller]
template <int TRANS, int ROW0, int KS>
__device__ __forceinline__ bf16x8 rd(const LaneAddr& la, uint32_t nt_base0, uint32_t nt_base1, uint32_t t_base,
                                     uint32_t t_x) {
    if constexpr (TRANS == 0) {
        return lds_b128<ROW0 * 128>(KS == 0 ? nt_base0 : nt_base1);
    } else {
        const uint32_t addr = ((uint32_t)((ROW0 >> 4) * 32) ^ t_x) + t_base;
        return lds_tr16x2<KS * 8192>(addr);
    }
}

#define SEG_BARRIER()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)
#ifdef LECCR_G256_STAMPS
#define STAMP(i)                                                            \
    do {                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();       \
        stamp_sum[i] += now_ - stamp_last;                                  \
        stamp_last = now_;                                                  \
        __builtin_amdgcn_sched_barrier(0);                                  \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
#define WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// Lanes of one wave exchange data through LDS without a barrier (LDS operations of a wave execute in order); for the
// compiler these are unsynchronised accesses, so without a fence it may reuse an earlier read of the same address or
// sink the read into the writers' exec-masked block (seen in the .s: wrong results).  A compiler-only memory barrier.
#define WAVE_LDS_FENCE() asm volatile("" ::: "memory")

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
#define ON(n, stmt) if constexpr (i == (n)) { stmt; }


// FLAGS (compile-time epilogue shape, so the per-item epilogue carries no runtime branches):
constexpr int F_ATOMIC = 1, F_RES = 2, F_AUXOUT = 4, F_ACCUM = 8, F_ASUM = 16;

template <int TRANS, typename TC, int EPI, int FLAGS>
__global__ __launch_bounds__(NTHREADS, 2) void gemm256_kernel(Args a) {
    constexpr bool ATOMIC = (FLAGS & F_ATOMIC) != 0;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in an SGPR: everything derived from it is scalar
    const int wr = wave >> 2, wc = wave & 3;          // wr doubles as the ping-pong group

    // persistent item assignment: XCD x (= block id % 8, a speed-only assumption) owns a contiguous range of items
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
        // G == n_items (< 256, not a multiple of 8): exactly one item per block, still XCD-contiguous:
        // XCD x owns blocks {bid % 8 == x} = (G + 7 - x) / 8 of them, and the item range that follows XCD x-1's
        const int x = bid & 7, jx = bid >> 3;
        int off = 0;
        for (int y = 0; y < x; ++y) off += (G + 7 - y) >> 3;
        first_item = off + jx;
        stride = 1;
        n_mine = 1;
    }
    if (n_mine == 0) return;                           // whole block exits together: no barrier imbalance

    const StageOff so = make_stage_off<TRANS>(a);
    const uint32_t smem_off = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // wave-uniform LDS destination of this wave's DMA pieces (M0): must be an SGPR or hipcc emits a waterfall loop
    const uint32_t wave_off = smem_off + (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x & ~63) * 16u;
    Cursor cB;                                         // prefetch position (see the schedule below)
    Cursor cur;
    cur.j = 0; cur.kt = 0;
    decode<TRANS>(a, first_item, stride, cur);
    cB = cur;

    if constexpr (TRANS == 1) {
        // v2 schedule (see the K loop below): K-tiles 0 and 1 up front, group 1 one segment behind group 0
        advance<TRANS, 0, 3>(a, first_item, stride, n_mine, cB);
        // prologue: K-tiles 0 and 1 (8 + 8 pieces); vmcnt(8) retires tile 0, publish
        issue_half<TRANS, 0>(a, so, cur, 0, wave_off);
        issue_half<TRANS, 1>(a, so, cur, 0, wave_off);
        issue_half<TRANS, 2>(a, so, cur, 0, wave_off);
        issue_half<TRANS, 3>(a, so, cur, 0, wave_off);
        issue_half<TRANS, 2>(a, so, cB, SLOT, wave_off);
        issue_half<TRANS, 3>(a, so, cB, SLOT, wave_off);
        issue_half<TRANS, 0>(a, so, cB, SLOT, wave_off);
        issue_half<TRANS, 1>(a, so, cB, SLOT, wave_off);
        advance<TRANS, 0, 3>(a, first_item, stride, n_mine, cB);        // cB -> t+2
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        SEG_BARRIER();
        if (wr == 1) SEG_BARRIER();                        // group 1 runs one segment behind group 0
    } else {
        // prologue: K-tiles 0 and 1 (8 + 8 pieces); vmcnt(8) retires K-tile 0
        issue_half<TRANS, 0>(a, so, cB, 0, wave_off);
        issue_half<TRANS, 1>(a, so, cB, 0, wave_off);
        issue_half<TRANS, 2>(a, so, cB, 0, wave_off);
        issue_half<TRANS, 3>(a, so, cB, 0, wave_off);
        advance<TRANS, 0, 3>(a, first_item, stride, n_mine, cB);        // cB -> K-tile 1
        issue_half<TRANS, 0>(a, so, cB, SLOT, wave_off);
        issue_half<TRANS, 1>(a, so, cB, SLOT, wave_off);
        issue_half<TRANS, 2>(a, so, cB, SLOT, wave_off);
        issue_half<TRANS, 3>(a, so, cB, SLOT, wave_off);
        advance<TRANS, 0, 3>(a, first_item, stride, n_mine, cB);        // cB -> K-tile 2
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        SEG_BARRIER();
    }

    LaneAddr la = make_lane_addr(lane);
    // wave-uniform pieces of the fragment addresses: this wave's A half, its B half and its 64-row window inside it
    const uint32_t offA = smem_off + wr * HALF;
    const uint32_t offB = smem_off + (2 + (wc >> 1)) * HALF;
    const uint32_t bwin_nt = (wc & 1) * 64 * 128;          // NT: + 64 rows of 128 B
    const uint32_t bxs = (wc & 1) * 4 * 32;                 // TN: the window is 4 column blocks of 32 B
#define bx (la.tx ^ bxs)

    // Fragment registers: A rows 0-63 / 64-127 of the wave's half, B rows 0-31 / 32-63 of its window.
    bf16x8 aLo[4][2], aHi[4][2], bL[2][2], bH[2][2];
#define RD_A(S, ROW0, KS) rd<TRANS, ROW0, KS>(la, la.nt[0] + offA + (S), la.nt[1] + offA + (S), la.tbase + offA + (S), la.tx)
#define RD_B(S, ROW0, KS) rd<TRANS, ROW0, KS>(la, la.nt[0] + offB + bwin_nt + (S), la.nt[1] + offB + bwin_nt + (S), la.tbase + offB + (S), bx)
#define TIE4(f) "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1])
#define TIE8(f) "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1])
    // first fragments of a K-tile outside the steady state: A rows 0-63 and the k-step-0 fragments of B rows 0-31
    auto first_frags = [&](uint32_t slot) {
        aLo[0][0] = RD_A(slot, 0, 0); aLo[0][1] = RD_A(slot, 0, 1); aLo[1][0] = RD_A(slot, 16, 0); aLo[1][1] = RD_A(slot, 16, 1);
        aLo[2][0] = RD_A(slot, 32, 0); aLo[2][1] = RD_A(slot, 32, 1); aLo[3][0] = RD_A(slot, 48, 0); aLo[3][1] = RD_A(slot, 48, 1);
        bL[0][0] = RD_B(slot, 0, 0); bL[1][0] = RD_B(slot, 16, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : TIE8(aLo), "+v"(bL[0][0]), "+v"(bL[1][0]));
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (TRANS == 0) first_frags(0);

    int t = 0;                                         // global K-tile counter of this block's stream (slot parity)
#ifdef LECCR_G256_STAMPS
    unsigned long long sk_ = 0, se_ = 0, sf_ = 0;
    const unsigned long long start_ = __builtin_amdgcn_s_memrealtime(), c0_ = __builtin_amdgcn_s_memtime();
    unsigned long long ts_ = start_;
#endif
    for (int j = 0; j < n_mine; ++j) {
        f32x4 acc[4][8];                               // [ni][mi]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};

        if constexpr (TRANS == 1) {
        // ---- v2 schedule (weight gradients, TN): LOAD / COMPUTE segments alternating between the two waves of a SIMD.
        // (The software-pipelined NT schedule below needs 96 fragment registers; the transposed reads return register
        //  PAIRS and the allocator spills ~80 registers inside the K loop with it - measured 3x slower.)
        bf16x8 af[4][2], b0v[2][2], b1v[2][2];
        // F_ASUM: column sums of A over the reduction index (= the bias gradient colsum(dY) when this product is dW = dY^T X),
        // on the matrix cores: one more MFMA per k-step with an all-ones fragment in place of the B fragment gives
        // D[n][m] = sum_k A[k][m] for every n.  The four waves of a wave row hold the same A fragments, so wave column wc
        // takes 16-row group mi = wc of each 64-row half: 4 extra MFMAs per K-tile and wave, inside the COMPUTE segments.  The
        // tiles_n items that share an A panel (same tile row and split) share the work: the item of tile column c sums the
        // K-tiles kt = c (mod tiles_n), so every workgroup carries the same 1/tiles_n of it (with the sums in tile column 0
        // alone those workgroups ran 6 % longer than the rest of the launch: +2 ms of GEMM time per step).  It replaces a
        // separate HBM pass over dY (leccr_colsum: 139 launches, 5.7 ms of a 143 ms step).
        [[maybe_unused]] f32x4 cs_lo = {0.f, 0.f, 0.f, 0.f}, cs_hi = {0.f, 0.f, 0.f, 0.f};
        [[maybe_unused]] const bool cs_on = (FLAGS & F_ASUM) != 0 && a.asum != nullptr;
        [[maybe_unused]] int cs_ctr = (int)(cur.n0 >> 8);      // K-tiles until this item's next turn (BN = 256)
        [[maybe_unused]] bool do_cs = false;
#define ASUM_STEP(CS)                                                                                                  \
        do {                                                                                                         \
            if constexpr ((FLAGS & F_ASUM) != 0) {                                                                   \
                if (do_cs) {                                                                                         \
                    const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f}; \
                    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                 \
                        if (wc == mi) {                                                                              \
                            CS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[mi][0], CS, 0, 0, 0);              \
                            CS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[mi][1], CS, 0, 0, 0);              \
                        }                                                                                            \
                }                                                                                                    \
            }                                                                                                        \
        } while (0)

#define MMA_QUAD(BF, NI0, MI0)                                                                                       \
        do {                                                                                                         \
            __builtin_amdgcn_s_setprio(1);                                                                           \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                         \
            _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                         \
                acc[NI0 + ni][MI0 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[ni][ks], af[mi][ks],            \
                                                                                  acc[NI0 + ni][MI0 + mi], 0, 0, 0); \
            __builtin_amdgcn_s_setprio(0);                                                                           \
        } while (0)

        for (int kt = 0; kt < a.nkt; ++kt, ++t) {
            if constexpr ((FLAGS & F_ASUM) != 0) {
                do_cs = cs_on && cs_ctr == 0;
                cs_ctr = cs_ctr == 0 ? a.tiles_n - 1 : cs_ctr - 1;
            }
            const uint32_t so_ = (t & 1) * SLOT;                 // this K-tile's slot; K-tile t+1 lives in SLOT - so_
            const uint32_t a0 = la.nt[0] + offA + so_, a1 = la.nt[1] + offA + so_, at = la.tbase + offA + so_;
            const uint32_t b0 = la.nt[0] + offB + so_ + bwin_nt, b1 = la.nt[1] + offB + so_ + bwin_nt;
            const uint32_t bt = la.tbase + offB + so_;
#define RD2_A(ROW0, KS) rd<TRANS, ROW0, KS>(la, a0, a1, at, la.tx)
#define RD2_B(ROW0, KS) rd<TRANS, ROW0, KS>(la, b0, b1, bt, bx)

            // ---- L1: A rows 0-63 + B rows 0-31 (12 reads, no DMA) | C(0,0)
            b0v[0][0] = RD2_B(0, 0); b0v[0][1] = RD2_B(0, 1); b0v[1][0] = RD2_B(16, 0); b0v[1][1] = RD2_B(16, 1);
            af[0][0] = RD2_A(0, 0); af[0][1] = RD2_A(0, 1); af[1][0] = RD2_A(16, 0); af[1][1] = RD2_A(16, 1);
            af[2][0] = RD2_A(32, 0); af[2][1] = RD2_A(32, 1); af[3][0] = RD2_A(48, 0); af[3][1] = RD2_A(48, 1);
            WAIT_LDS();
            SEG_BARRIER();
            MMA_QUAD(b0v, 0, 0);
            ASUM_STEP(cs_lo);                                    // af = A rows 0-63 of this wave row's half
            SEG_BARRIER();

            // ---- L2: B rows 32-63 (4 reads; last read of this wave column's B half) | C(0,1)
            b1v[0][0] = RD2_B(32, 0); b1v[0][1] = RD2_B(32, 1); b1v[1][0] = RD2_B(48, 0); b1v[1][1] = RD2_B(48, 1);
            WAIT_LDS();
            SEG_BARRIER();
            MMA_QUAD(b1v, 2, 0);
            SEG_BARRIER();

            // ---- L3: A rows 64-127 (8 reads; last read of this wave row's A half) + issue B(t+2) | C(1,1)
            af[0][0] = RD2_A(64, 0); af[0][1] = RD2_A(64, 1); af[1][0] = RD2_A(80, 0); af[1][1] = RD2_A(80, 1);
            af[2][0] = RD2_A(96, 0); af[2][1] = RD2_A(96, 1); af[3][0] = RD2_A(112, 0); af[3][1] = RD2_A(112, 1);
            issue_half<TRANS, 2>(a, so, cB, so_, wave_off);
            issue_half<TRANS, 3>(a, so, cB, so_, wave_off);
            WAIT_LDS();
            SEG_BARRIER();
            MMA_QUAD(b1v, 2, 4);
            ASUM_STEP(cs_hi);                                    // af = A rows 64-127
            SEG_BARRIER();

            // ---- L4: no reads (B rows 0-31 are still in registers); issue A(t+2); the K-tile's one counted wait | C(1,0)
            issue_half<TRANS, 0>(a, so, cB, so_, wave_off);
            issue_half<TRANS, 1>(a, so, cB, so_, wave_off);
            advance<TRANS, 0, 3>(a, first_item, stride, n_mine, cB);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // retires K-tile t+1 (8 loads); t+2 stays in flight
            SEG_BARRIER();
            MMA_QUAD(b0v, 0, 4);
            SEG_BARRIER();
        }
        if constexpr ((FLAGS & F_ASUM) != 0) {
            // lanes 0-15 hold D[n = 0..3][m = lane] (all n equal): one f32 atomic per A column and split
            if (cs_on && lane < 16) {
                const int64_t m = cur.m0 + wr * 128 + wc * 16 + lane;
                if (m < a.M) atomicAdd(a.asum + m, cs_lo[0] * a.alpha);
                if (m + 64 < a.M) atomicAdd(a.asum + m + 64, cs_hi[0] * a.alpha);
            }
        }
        } else {
        // One K-tile = four quads of 16 MFMAs (k-step 0 of the quad's 8 accumulators, then k-step 1); every LDS read and
        // DMA piece is issued BETWEEN MFMAs of the same wave, one per MFMA at most, from asm, in the order written
        // (sched_barrier pins each step).  A fragment register is reloaded right after the last MFMA that needs its
        // old contents has been ISSUED - there are no spare copies (A 64 + B 32 registers beside 128 accumulators):
        //   Q1 (bL, aLo): reads bL[.][1](t) x2, bH(t) x4, aHi(t) x4                  (bL[.][1] awaited before k-step 1)
        //   Q2 (bH, aLo): reads aHi(t) x4                                           -> wait; BARRIER: slot(t) is free
        //   Q3 (bH, aHi):                                             DMA A(t+2) x4 -> vmcnt(4); BARRIER: t+1 visible
        //   Q4 (bL, aHi): reads aLo(t+1) x8, then bL[.][0](t+1) x2    DMA B(t+2) x4 -> wait
        // All 8 pieces of K-tile t+2 go out right after its slot is freed: the last one then has ~3 quads to land before
        // the vmcnt wait (spread over all four quads the last pieces had < 1 quad and the wait stalled on HBM latency).
        // (v2 alternated LOAD and COMPUTE segments between the two waves of a SIMD, 8 barriers per K-tile: its LOAD
        //  segments - 12 ds_reads or 4 DMA pieces back to back - were 2x longer than the 256-cycle COMPUTE segments.)
        auto quad = [&](auto ni0_tag, auto mi0_tag, bf16x8 (&bf)[2][2], bf16x8 (&af)[4][2], auto&& fill) {
            constexpr int NI0 = decltype(ni0_tag)::value, MI0 = decltype(mi0_tag)::value;
            static_for<0, 16>([&](auto ic) {
                constexpr int idx = decltype(ic)::value, ks = idx >> 3, ni = (idx >> 2) & 1, mi = idx & 3;
                acc[NI0 + ni][MI0 + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni][ks], af[mi][ks], acc[NI0 + ni][MI0 + mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                fill(ic);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        using I0 = std::integral_constant<int, 0>;
        using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;

        for (int kt = 0; kt < a.nkt; ++kt, ++t) {
            const bool last_of_item = kt + 1 == a.nkt;
            const uint32_t sc = (t & 1) * SLOT, sn = SLOT - sc;        // this K-tile's slot, the next one's
            // per-lane address pieces pass through an opaque asm each K-tile: the ~30 fragment addresses derived from them
            // are then recomputed where used (1 VALU op) instead of being hoisted out of the loop and spilled
            asm volatile("" : "+v"(la.nt[0]), "+v"(la.nt[1]), "+v"(la.tbase), "+v"(la.tx));
            // ---- Q1
            quad(I0{}, I0{}, bL, aLo, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ON(0, bL[0][1] = RD_B(sc, 0, 1))
                ON(1, bL[1][1] = RD_B(sc, 16, 1))
                ON(2, bH[0][0] = RD_B(sc, 32, 0))
                ON(3, bH[0][1] = RD_B(sc, 32, 1))
                ON(4, bH[1][0] = RD_B(sc, 48, 0))
                ON(5, bH[1][1] = RD_B(sc, 48, 1))
                ON(6, aHi[0][0] = RD_A(sc, 64, 0))
                ON(7, aHi[0][1] = RD_A(sc, 64, 1); asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bL[0][1]), "+v"(bL[1][1])))
                ON(8, aHi[1][0] = RD_A(sc, 80, 0))
                ON(9, aHi[1][1] = RD_A(sc, 80, 1))
            });
            asm volatile("s_waitcnt lgkmcnt(4)" : TIE4(bH));
            __builtin_amdgcn_sched_barrier(0);
            // ---- Q2
            quad(I2{}, I0{}, bH, aLo, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ON(0, aHi[2][0] = RD_A(sc, 96, 0))
                ON(1, aHi[2][1] = RD_A(sc, 96, 1))
                ON(2, aHi[3][0] = RD_A(sc, 112, 0))
                ON(3, aHi[3][1] = RD_A(sc, 112, 1))
            });
            asm volatile("s_waitcnt lgkmcnt(0)" : TIE8(aHi));
            SEG_BARRIER();                                                  // every wave is done reading slot(t)
            // ---- Q3
            quad(I2{}, I4{}, bH, aHi, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ON(3, (issue_half<TRANS, 0, 0, 1>(a, so, cB, sc, wave_off)))
                ON(7, (issue_half<TRANS, 0, 1, 2>(a, so, cB, sc, wave_off)))
                ON(11, (issue_half<TRANS, 1, 0, 1>(a, so, cB, sc, wave_off)))
                ON(15, (issue_half<TRANS, 1, 1, 2>(a, so, cB, sc, wave_off)))
            });
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                // this wave's pieces of K-tile t+1 have landed
            SEG_BARRIER();                                                  // ... and everyone else's
            // ---- Q4
            if (!last_of_item) {
                quad(I0{}, I4{}, bL, aHi, [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    ON(0, aLo[0][0] = RD_A(sn, 0, 0))
                    ON(1, aLo[0][1] = RD_A(sn, 0, 1))
                    ON(2, aLo[1][0] = RD_A(sn, 16, 0))
                    ON(3, aLo[1][1] = RD_A(sn, 16, 1))
                    ON(4, aLo[2][0] = RD_A(sn, 32, 0))
                    ON(5, aLo[2][1] = RD_A(sn, 32, 1))
                    ON(6, aLo[3][0] = RD_A(sn, 48, 0))
                    ON(7, aLo[3][1] = RD_A(sn, 48, 1))
                    ON(8, bL[0][0] = RD_B(sn, 0, 0))
                    ON(9, bL[1][0] = RD_B(sn, 16, 0))
                    ON(10, (issue_half<TRANS, 2, 0, 1>(a, so, cB, sc, wave_off)))
                    ON(12, (issue_half<TRANS, 2, 1, 2>(a, so, cB, sc, wave_off)))
                    ON(14, (issue_half<TRANS, 3, 0, 1>(a, so, cB, sc, wave_off)))
                    ON(15, (issue_half<TRANS, 3, 1, 2>(a, so, cB, sc, wave_off)))
                });
                asm volatile("s_waitcnt lgkmcnt(0)" : TIE8(aLo), "+v"(bL[0][0]), "+v"(bL[1][0]));
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // the item's last K-tile: its epilogue needs the registers; the next item's first fragments are read after it
                quad(I0{}, I4{}, bL, aHi, [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    ON(3, (issue_half<TRANS, 2, 0, 1>(a, so, cB, sc, wave_off)))
                    ON(7, (issue_half<TRANS, 2, 1, 2>(a, so, cB, sc, wave_off)))
                    ON(11, (issue_half<TRANS, 3, 0, 1>(a, so, cB, sc, wave_off)))
                    ON(15, (issue_half<TRANS, 3, 1, 2>(a, so, cB, sc, wave_off)))
                });
            }
            advance<TRANS, 0, 3>(a, first_item, stride, n_mine, cB);        // cB -> K-tile t+3
        }

        }

        // ---- epilogue segment of this item (the partner wave is in its last COMPUTE / first LOAD meanwhile)
#ifdef LECCR_G256_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1_ = __builtin_amdgcn_s_memrealtime();
        sk_ += t1_ - ts_;
        __builtin_amdgcn_sched_barrier(0);
#endif
        {
            const int64_t m0 = cur.m0, n0 = cur.n0;
            const bool interior = m0 + BM <= a.M && n0 + BN <= a.N;
            const int64_t nb = n0 + wc * 64 + (lane >> 4) * 4;           // this lane's first column; tiles at +16*ni
            const bool has_bias = a.bias != nullptr && (!ATOMIC || cur.split == 0);
            // bias is consumed unconditionally right here: a load whose only uses sit behind the tail-tile branches would
            // stay "pending" in hipcc's scoreboard across the loop back edge and cost a vmcnt(0) in the K loop
            f32x4 bv[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                bv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_bias) {                      // uniform branch; tail columns read a clamped (valid) address
                    const int64_t nc = nb + ni * 16 < a.N ? nb + ni * 16 : a.N - 4;
                    bv[ni] = load4(a.bias + nc);
                }
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    acc[ni][mi] = acc[ni][mi] * a.alpha + bv[ni];
                    asm volatile("" : "+v"(acc[ni][mi]));      // here, not sunk into the store branches
                }
            }
            if constexpr (sizeof(TC) == 2 && !ATOMIC) {
                // bf16 output: transpose each 16 x 64 accumulator strip through a wave-private LDS buffer (8 rows at a
                // time) so that a lane owns 8 CONSECUTIVE columns of one row: every global access of the epilogue
                // (C, aux, aux_out, residual) is 16 bytes per lane and a wave instruction covers 8 full 128-byte rows,
                // instead of 8-byte pieces scattered over 16 rows (the store tail was issue-bound: 15-17 % of a
                // K = 768 item, in-kernel stamps).
                float* stg = reinterpret_cast<float*>(smem + 2 * SLOT) + wave * (16 * 64);
                // lane-derived epilogue addresses are re-derived per item from an opaque copy of the lane id: hoisted out
                // of the item loop they live across the K loop, get spilled, and their reloads cost a scratch round
                // trip each (with a vmcnt(0)) in every epilogue
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const int rrow = lane_e >> 3, rcol = (lane_e & 7) * 8;         // read side: rows rrow and rrow + 8, 8 columns
                const int64_t ncol = n0 + wc * 64 + rcol;
                const bool col_ok = interior || ncol < a.N;                    // N % 8 == 0 (dispatcher)
                // The epilogue operand that is READ (the saved pre-activation of a d-activation epilogue, else the
                // residual) is fetched for all 16 strips up front: loaded inside the strip loop each strip pays a full
                // memory round trip (16 serial latencies per item - 30-40 % of a K = 768 item).
                constexpr bool NEED_AUX = epi_needs_aux<EPI>;
                constexpr bool PRE_RES = (FLAGS & F_RES) != 0 && !NEED_AUX;
                // The staging buffer holds one whole 16 x 64 strip (rows of 16 chunks of 16 bytes, chunk ^= row): every lane
                // writes its 4 accumulators of the strip, then reads 8 columns of rows rrow and rrow + 8.  LDS operations of
                // a wave execute in order, so the strips are software-pipelined without any wait between them: write(mi+2) is
                // issued right behind read(mi+1), and the only waits are for the data of read(mi) while read(mi+1) is already
                // in flight (the serial write -> read -> store round trip per half strip was 3.5 us of a 24 us K = 768 item).
                const int wrow = lane_e & 15, wq = lane_e >> 4;
                float* wbase = stg + wrow * 64;
                const float* rbase0 = stg + rrow * 64;
                const float* rbase1 = stg + (rrow + 8) * 64;
                const int rc = (lane_e & 7) * 2;                                  // first of this lane's two chunks
                auto put = [&](int mi) {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(wbase + (((ni * 4 + wq) ^ wrow) << 2)) = acc[ni][mi];
                    WAVE_LDS_FENCE();
                };
                auto get = [&](f32x4 (&v)[4]) {
                    v[0] = *reinterpret_cast<const f32x4*>(rbase0 + ((rc ^ rrow) << 2));
                    v[1] = *reinterpret_cast<const f32x4*>(rbase0 + (((rc + 1) ^ rrow) << 2));
                    v[2] = *reinterpret_cast<const f32x4*>(rbase1 + ((rc ^ (rrow + 8)) << 2));
                    v[3] = *reinterpret_cast<const f32x4*>(rbase1 + (((rc + 1) ^ (rrow + 8)) << 2));
                    WAVE_LDS_FENCE();
                };
                // Two copies of the strip loop: interior tiles (the hot case) carry no per-lane bounds checks, so the
                // code is straight-line and hipcc's waits on the prefetched operand are counted (vmcnt(N)); with the
                // checks' branches in the way every strip waited vmcnt(0), i.e. for the previous strip's store as well.
                auto strips = [&](auto check_tag) {
                    constexpr bool CHECK = decltype(check_tag)::value;
                    bf16x8 pre[(NEED_AUX || PRE_RES) ? 8 : 1];       // 8 half strips per batch: two round trips per item
                    f32x4 vbuf[2][4];
                    const int64_t mrow = m0 + wr * 128 + rrow;       // this lane's row of strip 0, half 0
                    bf16_t* cptr = reinterpret_cast<bf16_t*>(a.C) + mrow * a.ldc + ncol;
                    bf16_t* optr = reinterpret_cast<bf16_t*>(a.aux_out) + mrow * a.ldc + ncol;
                    const int64_t step8 = 8 * a.ldc;
                    put(0);
                    get(vbuf[0]);
                    put(1);
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
                        if constexpr (NEED_AUX || PRE_RES) {
                            if ((mi & 3) == 0) {
                                const bf16_t* pbase = reinterpret_cast<const bf16_t*>(NEED_AUX ? a.aux : a.residual);
                                const int64_t pld = NEED_AUX ? a.ldaux : a.ldr;
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    const int64_t m = mrow + (mi + (i >> 1)) * 16 + (i & 1) * 8;
                                    const bool ok = !CHECK || (col_ok && m < a.M);
                                    pre[i] = *reinterpret_cast<const bf16x8*>(pbase + (ok ? m * pld + ncol : 0));
                                }
                            }
                        }
                        if (mi + 1 < 8) get(vbuf[(mi + 1) & 1]);
                        if (mi + 2 < 8) put(mi + 2);
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp) {
                            f32x4 v0 = vbuf[mi & 1][2 * hp], v1 = vbuf[mi & 1][2 * hp + 1];
                            const int64_t m = mrow + mi * 16 + hp * 8;
                            const bool ok = !CHECK || (col_ok && m < a.M);
                            if constexpr (epi_saves_deriv<EPI>) {
                                // the derivative goes where the pre-activation went: the backward epilogue (LECCR_EPI_MULAUX) only multiplies
                                f32x4 d0, d1;
#pragma unroll
                                for (int r = 0; r < 4; ++r) { { const ActD t_ = apply_epi_d<EPI>(v0[r]); v0[r] = t_.a; d0[r] = t_.d; } { const ActD t_ = apply_epi_d<EPI>(v1[r]); v1[r] = t_.a; d1[r] = t_.d; } }
                                bf16x8 pv;
#pragma unroll
                                for (int r = 0; r < 4; ++r) { pv[r] = (bf16_t)d0[r]; pv[4 + r] = (bf16_t)d1[r]; }
                                if (ok) G256_STORE(reinterpret_cast<bf16x8*>(optr), pv);
                                optr += step8;
                            } else if constexpr ((FLAGS & F_AUXOUT) != 0) {
                                bf16x8 pv;
#pragma unroll
                                for (int r = 0; r < 4; ++r) { pv[r] = (bf16_t)v0[r]; pv[4 + r] = (bf16_t)v1[r]; }
                                if (ok) G256_STORE(reinterpret_cast<bf16x8*>(optr), pv);
                                optr += step8;
                            }
                            if constexpr (EPI != LECCR_EPI_NONE && !epi_saves_deriv<EPI>) {
                                f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
                                if constexpr (NEED_AUX) {
                                    const bf16x8 xv = pre[(mi & 3) * 2 + hp];
#pragma unroll
                                    for (int r = 0; r < 4; ++r) { x0[r] = (float)xv[r]; x1[r] = (float)xv[4 + r]; }
                                }
#pragma unroll
                                for (int r = 0; r < 4; ++r) { v0[r] = apply_epi<EPI>(v0[r], x0[r]); v1[r] = apply_epi<EPI>(v1[r], x1[r]); }
                            }
                            if constexpr ((FLAGS & F_RES) != 0) {
                                bf16x8 rv;
                                if constexpr (PRE_RES) rv = pre[(mi & 3) * 2 + hp];
                                else rv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(a.residual) + (ok ? m * a.ldr + ncol : 0));
#pragma unroll
                                for (int r = 0; r < 4; ++r) { v0[r] += (float)rv[r]; v1[r] += (float)rv[4 + r]; }
                            }
                            bf16x8 ov;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { ov[r] = (bf16_t)v0[r]; ov[4 + r] = (bf16_t)v1[r]; }
#ifdef G256_EXP_NOSTORE
                            asm volatile("" ::"v"(ov));            // diagnostic build: the epilogue without its C stores (wrong results)
#else
                            if (ok) G256_STORE(reinterpret_cast<bf16x8*>(cptr), ov);
#endif
                            cptr += step8;
                        }
                    }
                };
                if (interior) strips(std::false_type{});
                else strips(std::true_type{});
            } else if constexpr (ATOMIC) {
                // split-K partial sums: the same wave-private LDS transpose, read back with the COLUMN on the lane, so one
                // global_atomic_add_f32 covers 256 contiguous bytes of one row.  (Straight from the accumulator layout an
                // instruction touches 16 rows x 4 scattered dwords: 4x the 64-byte atomic requests at the memory side,
                // whose ~1.3 TB/s is what bounds this tail - and a pending bias load made hipcc wait vmcnt(0), i.e. for
                // the previous atomics' acknowledgement, before every group.)
                float* stg = reinterpret_cast<float*>(smem + 2 * SLOT) + wave * (8 * 68);
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const int64_t ncol = n0 + wc * 64 + lane_e;
                float* cbase = reinterpret_cast<float*>(a.C) + ncol;
                auto strips = [&](auto check_tag) {
                    constexpr bool CHECK = decltype(check_tag)::value;
                    const bool col_ok = !CHECK || ncol < a.N;
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp) {
                            if (((lane & 15) >> 3) == hp) {
#pragma unroll
                                for (int ni = 0; ni < 4; ++ni)
                                    *reinterpret_cast<f32x4*>(stg + (lane & 7) * 68 + ni * 16 + (lane >> 4) * 4) = acc[ni][mi];
                            }
                            WAVE_LDS_FENCE();        // the reads below fetch what OTHER lanes of the wave just wrote
                            float v[8];
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] = stg[r * 68 + lane_e];
                            WAVE_LDS_FENCE();
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                const int64_t m = m0 + wr * 128 + mi * 16 + hp * 8 + r;      // wave-uniform
                                if (col_ok && (!CHECK || m < a.M)) atomicAdd(cbase + m * a.ldc, v[r]);
                            }
                        }
                    }
                };
                if (interior) strips(std::false_type{});
                else strips(std::true_type{});
            } else {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int64_t m = m0 + wr * 128 + mi * 16 + (lane & 15);
                if (!interior && m >= a.M) continue;
                TC* crow = reinterpret_cast<TC*>(a.C) + m * a.ldc + nb;
                const TC* rrow = reinterpret_cast<const TC*>(a.residual) + m * a.ldr + nb;
                const TC* xrow = reinterpret_cast<const TC*>(a.aux) + m * a.ldaux + nb;
                TC* orow = reinterpret_cast<TC*>(a.aux_out) + m * a.ldc + nb;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    if (!interior && nb + ni * 16 >= a.N) continue;
                    f32x4 v = acc[ni][mi];
                    {
                        if constexpr (epi_saves_deriv<EPI>) {
                            f32x4 dv;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const ActD t_ = apply_epi_d<EPI>(v[r]); v[r] = t_.a; dv[r] = t_.d; }
                            store4(orow + ni * 16, dv);
                        } else {
                            if constexpr ((FLAGS & F_AUXOUT) != 0) store4(orow + ni * 16, v);
                            if constexpr (EPI != LECCR_EPI_NONE) {
                                f32x4 ax = {0.f, 0.f, 0.f, 0.f};
                                if constexpr (epi_needs_aux<EPI>) ax = load4(xrow + ni * 16);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = apply_epi<EPI>(v[r], ax[r]);
                            }
                        }
                        if constexpr ((FLAGS & F_RES) != 0) v += load4(rrow + ni * 16);
                        if constexpr ((FLAGS & F_ACCUM) != 0) v += load4(crow + ni * 16);
                        if constexpr (TRANS == 0 && sizeof(TC) == 4) {
                            if (a.c3) acc[ni][mi] = v;          // kept for the split copy below
                            if (a.C) store4(crow + ni * 16, v);
                        } else {
                            store4(crow + ni * 16, v);
                        }
                    }
                }
            }
            if constexpr (TRANS == 0 && sizeof(TC) == 4) {
                // c3: the (hi, lo) operand copy the fp32 path's next product reads (leccr_split3's [M, 3 N] layout), from the epilogue values
                // now in the accumulators.  Straight from the accumulator layout a lane would write three 8-byte pieces of one row (16 rows x
                // 32 bytes per wave store: measured +7 % on the launch); through the wave-private LDS strip of the bf16 epilogue a lane owns 8
                // consecutive columns of two rows: 16-byte stores, 8 lanes = one 128-byte run per segment.
                if (a.c3) {
                    float* stg = reinterpret_cast<float*>(smem + 2 * SLOT) + wave * (16 * 64);
                    int lane_e = lane;
                    asm volatile("" : "+v"(lane_e));
                    const int rrow = lane_e >> 3, rc = (lane_e & 7) * 2;
                    const int64_t ncol = n0 + wc * 64 + (lane_e & 7) * 8;
                    const bool col_ok = interior || ncol < a.N;
                    const int wrow = lane_e & 15, wq = lane_e >> 4;
                    float* wbase = stg + wrow * 64;
                    const float* rb0 = stg + rrow * 64;
                    const float* rb1 = stg + (rrow + 8) * 64;
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) *reinterpret_cast<f32x4*>(wbase + (((ni * 4 + wq) ^ wrow) << 2)) = acc[ni][mi];
                        WAVE_LDS_FENCE();
                        f32x4 vv[4];
                        vv[0] = *reinterpret_cast<const f32x4*>(rb0 + ((rc ^ rrow) << 2));
                        vv[1] = *reinterpret_cast<const f32x4*>(rb0 + (((rc + 1) ^ rrow) << 2));
                        vv[2] = *reinterpret_cast<const f32x4*>(rb1 + ((rc ^ (rrow + 8)) << 2));
                        vv[3] = *reinterpret_cast<const f32x4*>(rb1 + (((rc + 1) ^ (rrow + 8)) << 2));
                        WAVE_LDS_FENCE();
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp) {
                            const int64_t m = m0 + wr * 128 + mi * 16 + hp * 8 + rrow;
                            if (!(interior || (col_ok && m < a.M))) continue;
                            bf16x8 h8, l8;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                h8[r] = (bf16_t)vv[2 * hp][r];
                                l8[r] = (bf16_t)(vv[2 * hp][r] - (float)h8[r]);
                                h8[4 + r] = (bf16_t)vv[2 * hp + 1][r];
                                l8[4 + r] = (bf16_t)(vv[2 * hp + 1][r] - (float)h8[4 + r]);
                            }
                            bf16_t* d3 = a.c3 + m * 3 * a.N + ncol;
                            G256_STORE(reinterpret_cast<bf16x8*>(d3), h8);
                            G256_STORE(reinterpret_cast<bf16x8*>(d3 + a.N), a.c3_form == 0 ? h8 : l8);
                            G256_STORE(reinterpret_cast<bf16x8*>(d3 + 2 * a.N), a.c3_form == 0 ? l8 : h8);
                        }
                    }
                }
            }
            }
            // a compiler-visible wait: every ordinary load of the epilogue is retired for hipcc's scoreboard, so the K
            // loop carries no conservative vmcnt(0) (costs the in-flight A(t+2), once per item, long landed by now)
            __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0) expcnt(7) lgkmcnt(15)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TRANS == 1) SEG_BARRIER();
        }
#ifdef LECCR_G256_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t2_ = __builtin_amdgcn_s_memrealtime();
        se_ += t2_ - t1_;
        __builtin_amdgcn_sched_barrier(0);
#endif
        // move `cur` to the next item and fetch the first fragments of its first K-tile (already visible: Q3's barrier)
        if (j + 1 < n_mine) {
            cur.j = j + 1;
            cur.kt = 0;
            decode<TRANS>(a, first_item, stride, cur);
            if constexpr (TRANS == 0) first_frags((t & 1) * SLOT);
        }
#ifdef LECCR_G256_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        ts_ = __builtin_amdgcn_s_memrealtime();
        sf_ += ts_ - t2_;
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
#ifdef LECCR_G256_STAMPS
    if (a.stamps != nullptr && tid == 0) {
        unsigned long long* o = a.stamps + 32 + (size_t)blockIdx.x * 8;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = start_; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = sk_; o[3] = se_; o[4] = sf_; o[5] = (unsigned long long)n_mine;
        o[6] = xcc & 15; o[7] = __builtin_amdgcn_s_memtime() - c0_;
    }
#endif
    if constexpr (TRANS == 1) {
        if (wr == 0) SEG_BARRIER();                    // group 0 waits out group 1's trailing segment
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // clamped tail prefetches still target this block's LDS
}

}  // namespace g256

// ---------------------------------------------------------------------------------------------------- launch
// epilogue shapes compiled for the 256 kernel (everything else stays on gemm.hip's 128x128 kernel)
static int flags_of(const leccr_gemm_params& p) {
    using namespace g256;
    return (p.residual ? F_RES : 0) | (p.aux_out ? F_AUXOUT : 0) | (p.accumulate ? F_ACCUM : 0);
}

static bool shape_compiled(const leccr_gemm_params& p) {
    using namespace g256;
    const int f = flags_of(p);
    if (p.trans_a == 1) return p.c_dtype == LECCR_F32 && p.epilogue == LECCR_EPI_NONE && (f == F_ACCUM);
    // (f32 C: the same epilogue shapes - the reference-precision path runs its fp32 operands as bf16 hi / lo splits, split.hip)
    switch (p.epilogue) {
        case LECCR_EPI_NONE: return f == 0 || f == F_RES;
        case LECCR_EPI_QUICKGELU:
        case LECCR_EPI_GELU:
        case LECCR_EPI_QUICKGELU_D:
        case LECCR_EPI_GELU_D: return f == F_AUXOUT;
        case LECCR_EPI_DQUICKGELU:
        case LECCR_EPI_DGELU:
        case LECCR_EPI_MULAUX: return f == 0;
    }
    return false;
}

bool gemm256_legal(const leccr_gemm_params& p) {
    return p.M >= 256 && p.N >= 256 && (p.N & 7) == 0 && (p.ldc & 7) == 0 && (p.ldr & 7) == 0 && (p.ldaux & 7) == 0 &&
           p.lda < (1 << 24) && p.ldb < (1 << 24) && shape_compiled(p);
}

// Column sums of C (leccr_gemm_params.colsum) are NOT fused into this kernel: the pipelined NT loop sits at the 256-register
// limit and the extra epilogue state (8 accumulators + the cross-lane reduction) tipped it into ~135 spills inside the K loop
// (all NT GEMMs 2x slower); per-item global atomics were worse still (14,000 adders per address).  leccr_gemm runs the
// colsum pass over C instead.
bool gemm256_fuses_colsum(const leccr_gemm_params&) { return false; }
// The column sums of A over the reduction index (leccr_gemm_params.asum, trans_a = 1) ARE fused into the TN instantiations.
bool gemm256_fuses_asum(const leccr_gemm_params& p) { return p.trans_a == 1 && p.asum != nullptr; }

bool gemm256_wanted(const leccr_gemm_params& p) {
    const int64_t tm = cdiv(p.M, g256::BM), tn = cdiv(p.N, g256::BN);
    if (!gemm256_legal(p)) return false;
    if (p.trans_a == 0) return tm * tn >= 192 && p.K >= 128;        // enough tiles to fill 256 CUs
    return p.K >= 4096;                                                // weight gradients: deep reduction, split-K
}

template <int TRANS, typename TC, int EPI, int FLAGS>
static void launch_inst(const g256::Args& a, int grid, hipStream_t st) {
    using namespace g256;
    auto kfn = gemm256_kernel<TRANS, TC, EPI, FLAGS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NTHREADS), LDS_BYTES, st, a);
}

static unsigned long long* g_stamps = nullptr;
void gemm256_set_stamps(void* p) { g_stamps = (unsigned long long*)p; }

// Workgroups per launch (one per CU, static item lists).  A collective's kernel resident on some CUs leaves no room there for
// a workgroup that needs the CU's whole LDS and register file: a 256-workgroup launch would then run its last workgroups as a
// second round (2x the time).  While a gradient all-reduce overlaps the backward pass, leccr_amd.parallel lowers this to
// 256 - (CUs left to RCCL); always a multiple of 8 (XCD-contiguous item lists).
static int g_max_blocks = 256;
static int64_t g_max_blocks_launches = -1;       // >= 0: launches left before the limit lapses by itself
void gemm256_set_max_blocks(int n, int64_t launches) {
    g_max_blocks = n <= 0 || n > 256 ? 256 : (n < 8 ? 8 : n & ~7);
    g_max_blocks_launches = g_max_blocks == 256 ? -1 : launches;
}
int gemm256_max_blocks() { return g_max_blocks; }

int gemm256_launch(const leccr_gemm_params& p, hipStream_t st) {
    using namespace g256;
    const int cap = g_max_blocks;
    if (g_max_blocks_launches > 0 && --g_max_blocks_launches == 0) {       // this is the last limited launch
        g_max_blocks = 256;
        g_max_blocks_launches = -1;
    }
    Args a{};
    a.stamps = g_stamps;
    a.A = (const bf16_t*)p.A; a.B = (const bf16_t*)p.B; a.C = p.C;
    a.M = p.M; a.N = p.N; a.K = p.K; a.lda = p.lda; a.ldb = p.ldb; a.ldc = p.ldc;
    a.bias = p.bias; a.aux = p.aux; a.ldaux = p.ldaux; a.aux_out = p.aux_out; a.residual = p.residual; a.ldr = p.ldr;
    a.alpha = p.alpha; a.accumulate = p.accumulate;
    a.asum = p.trans_a == 1 ? p.asum : nullptr;
    a.c3 = (p.trans_a == 0 && p.c_dtype == LECCR_F32) ? (bf16_t*)p.c3 : nullptr;
    a.c3_form = p.c3_form;

    a.tiles_m = (int)cdiv(p.M, BM); a.tiles_n = (int)cdiv(p.N, BN);
    const int64_t ktiles = cdiv(p.K, BK);
    const int tiles = a.tiles_m * a.tiles_n;
    int splits = 1;
    if (p.trans_a == 1) {
        // one item per CU in ONE round: tiles * splits <= cap (a second, partial round would double the time)
        splits = tiles >= cap ? 1 : cap / tiles;
        if (splits > ktiles / 8) splits = (int)(ktiles / 8);
        if (splits < 1) splits = 1;
    }
    a.splits = splits;
    a.band_rows = (a.tiles_m + 7) / 8;
    a.col_group = 0;
    if (p.trans_a == 0) {
        // Opt-in (LECCR_G256_COLGROUP=c): walk the tile columns in groups of c inside each XCD band (decode()).  Measured on
        // MI355X (profiles/r02_gemm256_colgroup_sweep.txt): it cuts the fabric reads of the wide-N shapes as intended (qkv 1.35 ->
        // 0.81-0.95 GB, fc1 2.16 -> 1.04-1.25 GB per launch against 0.23 GB algorithmic) and makes them no faster (qkv 587 us
        // row-major vs 602-640 grouped, fc1 688 vs 684-738): the re-fetched weight panels are Infinity-Cache hits, and the K loop is
        // not waiting for them.  Row-major stays the default.
        static const int env_group = [] { const char* e = getenv("LECCR_G256_COLGROUP"); return e ? atoi(e) : 0; }();
        // Round 5, with non-temporal output stores: still no lever for the step's shapes (tiles_n <= 16: profiles/r05_gemm256_colgroup_again.txt),
        // but a very wide output (8192^3: 32 tile columns - the row-major walk streams all of B through each XCD's 4 MiB L2 once per tile row)
        // gains 5 % from groups of 8 (1245 -> 1307 - 1318 TFLOP/s on one box).  Automatic from 24 tile columns up; LECCR_G256_COLGROUP=-1: never.
        const int want = env_group != 0 ? env_group : (a.tiles_n >= 24 ? 8 : 0);
        const int cg = want > 0 && want < a.tiles_n ? want : 0;
        a.col_group = cg;
    }
    a.nkt = (int)cdiv(ktiles, splits);
    a.n_items = tiles * splits;
    const int grid = a.n_items < cap ? a.n_items : cap;              // < cap: one item per block (any count)
    const int f = flags_of(p);
    if (p.trans_a == 1) {
        if (a.asum) {
            if (splits > 1) launch_inst<1, float, LECCR_EPI_NONE, F_ATOMIC | F_ASUM>(a, grid, st);
            else launch_inst<1, float, LECCR_EPI_NONE, F_ACCUM | F_ASUM>(a, grid, st);
        } else if (splits > 1) launch_inst<1, float, LECCR_EPI_NONE, F_ATOMIC>(a, grid, st);
        else launch_inst<1, float, LECCR_EPI_NONE, F_ACCUM>(a, grid, st);
    } else if (p.c_dtype == LECCR_F32) {
        switch (p.epilogue) {
            case LECCR_EPI_NONE:
                if (f == F_RES) launch_inst<0, float, LECCR_EPI_NONE, F_RES>(a, grid, st);
                else launch_inst<0, float, LECCR_EPI_NONE, 0>(a, grid, st);
                break;
            case LECCR_EPI_QUICKGELU: launch_inst<0, float, LECCR_EPI_QUICKGELU, F_AUXOUT>(a, grid, st); break;
            case LECCR_EPI_GELU: launch_inst<0, float, LECCR_EPI_GELU, F_AUXOUT>(a, grid, st); break;
            case LECCR_EPI_DQUICKGELU: launch_inst<0, float, LECCR_EPI_DQUICKGELU, 0>(a, grid, st); break;
            case LECCR_EPI_DGELU: launch_inst<0, float, LECCR_EPI_DGELU, 0>(a, grid, st); break;
            case LECCR_EPI_QUICKGELU_D: launch_inst<0, float, LECCR_EPI_QUICKGELU_D, F_AUXOUT>(a, grid, st); break;
            case LECCR_EPI_GELU_D: launch_inst<0, float, LECCR_EPI_GELU_D, F_AUXOUT>(a, grid, st); break;
            case LECCR_EPI_MULAUX: launch_inst<0, float, LECCR_EPI_MULAUX, 0>(a, grid, st); break;
            default: set_error("gemm256: bad epilogue %d", p.epilogue); return LECCR_EINVAL;
        }
    } else {
        switch (p.epilogue) {
            case LECCR_EPI_NONE:
                if (f == F_RES) launch_inst<0, bf16_t, LECCR_EPI_NONE, F_RES>(a, grid, st);
                else launch_inst<0, bf16_t, LECCR_EPI_NONE, 0>(a, grid, st);
                break;
            case LECCR_EPI_QUICKGELU: launch_inst<0, bf16_t, LECCR_EPI_QUICKGELU, F_AUXOUT>(a, grid, st); break;
            case LECCR_EPI_GELU: launch_inst<0, bf16_t, LECCR_EPI_GELU, F_AUXOUT>(a, grid, st); break;
            case LECCR_EPI_DQUICKGELU: launch_inst<0, bf16_t, LECCR_EPI_DQUICKGELU, 0>(a, grid, st); break;
            case LECCR_EPI_DGELU: launch_inst<0, bf16_t, LECCR_EPI_DGELU, 0>(a, grid, st); break;
            case LECCR_EPI_QUICKGELU_D: launch_inst<0, bf16_t, LECCR_EPI_QUICKGELU_D, F_AUXOUT>(a, grid, st); break;
            case LECCR_EPI_GELU_D: launch_inst<0, bf16_t, LECCR_EPI_GELU_D, F_AUXOUT>(a, grid, st); break;
            case LECCR_EPI_MULAUX: launch_inst<0, bf16_t, LECCR_EPI_MULAUX, 0>(a, grid, st); break;
            default: set_error("gemm256: bad epilogue %d", p.epilogue); return LECCR_EINVAL;
        }
    }
    return check_launch("gemm256");
}

}  // namespace leccr
