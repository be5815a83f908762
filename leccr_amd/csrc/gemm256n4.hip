// gemm256n4.hip - the 256x256 bf16 NT GEMM tile (forward / dX products, bf16 out) on FOUR waves with v_mfma_f32_16x16x32_bf16 and a
// four-stage LDS ring: round 5's answer to "a GEMM main loop at the vendor's rate" (clip/model.py:177-191 is 70 % of the step's kernel time).
//
// What the earlier kernels left on the table (profiles/r04_gemm_vendor_kernels.txt, profiles/r01_ktile_mix_random_operands.txt):
//   * the vendor library runs this tile on 4 waves with 16x16 MFMAs; of the four instruction mixes measured on random operands the
//     4-wave 16x16x32 one is the fastest (1363 TF/s-equivalent against 1285 for 4-wave 32x32x16 and 1160-1250 for the 8-wave mixes): one
//     wave per SIMD owns the issue port and 512 registers, and the 16x16x32 shape holds a higher clock than 32x32x16 at equal cycles;
//   * both earlier kernels issue a K-tile's LDS-DMA pieces in bursts (8 pieces behind 8 consecutive MFMAs, four waves at once = twice
//     what the CU's vector-memory path takes) because with TWO 64 KiB slots a piece must land within the K-tile it is issued in.
//
// Structure
//   * 256 threads = 4 waves = 2 (M) x 2 (N); wave tile 128 x 128 = 8 x 8 accumulators of v_mfma_f32_16x16x32_bf16 (256 registers), issued
//     with swapped roles (the B rows are the A operand) so that a lane ends with 4 consecutive n of one row:
//     register r of accumulator (mi, ni) in lane l = C[m = 16 mi + (l & 15)][n = 16 ni + 4 (l >> 4) + r].
//   * LDS: a ring of FOUR stages of 32 KiB; a stage = one k-step (32 k) of the tile: A rows 0-255 then B rows 0-255, 64-byte rows.
//     Physical 16-byte chunk = logical chunk ^ g[(row >> 2) & 3], g = {0, 3, 2, 1}: every 16-lane group of a fragment ds_read_b128
//     (row = l & 15, logical chunk = l >> 4) then covers all 64 banks.  The swizzle is applied to the SOURCE address (LDS-DMA writes
//     lane-linear); a DMA piece = 16 rows x 64 bytes.
//   * Stage t: 64 MFMAs on fragment buffer t & 1; behind them the 16 fragment reads of stage t + 1 (into the other buffer) and the 8
//     DMA pieces of this wave for stage t + 4 (into the slot stage t was read from, one stage earlier) - one piece per 8 MFMAs, every
//     stage alike, so the vector-memory path sees a constant rate; at the stage's end vmcnt(16) (stages t + 3, t + 4 stay in flight:
//     every piece has two to three stages = 2000-3000 cycles to land) and ONE barrier.
//   * Persistent items, XCD-contiguous, the stage stream continuous across items (the ring never drains); an item's first stage
//     starts its accumulators from the instruction's zero operand (no 256-register clear per item).
//   * Epilogue: per 16-row strip a wave-private LDS transpose (16 rows x 128 f32, XOR-swizzled) so that 16 consecutive lanes store
//     256 contiguous bytes of one row; bias / activation (+ saved pre-activation) / derivative / residual as gemm256.hip.  Its LDS
//     accesses are asm (a C++ LDS access with LDS-DMA in flight makes hipcc drain vmcnt to 0).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace leccr {
#ifndef N4_EXP
#define N4_EXP 0              // diagnostic builds (wrong results): 1 no vmcnt wait, 2 no DMA in the loop, 4 no barrier, 8 no fragment reads, 16 no epilogue,
                              // 32 every DMA piece re-reads one 1 KiB block (issue cost without memory traffic), 64 pieces read 8 whole 128-byte lines
                              // instead of 16 half lines, 128 epilogue without its global stores, 256 epilogue stores without the transposes,
                              // 512 / 1024 (with 128): two stores per stage spread over the first N4_HEAD stages of the NEXT item instead (64-byte / 256-byte row
                              // segments): what the epilogue's stores cost when they drain under the K loop
#endif
#ifndef N4_DIRECT
#define N4_DIRECT 1           // epilogue without LDS: v_permlane16_swap pairs the 16-column accumulator tiles so that a lane holds 8 consecutive columns
#endif
#ifndef N4_RD_STEP
#define N4_RD_STEP 2          // fragment read f (0..15) goes behind MFMA f * N4_RD_STEP
#endif
#ifndef N4_DMA_STEP
#define N4_DMA_STEP 8         // DMA piece p (0..7) goes behind MFMA N4_DMA_FIRST + p * N4_DMA_STEP
#endif
#ifndef N4_DMA_FIRST
#define N4_DMA_FIRST 3
#endif
#ifndef N4_HEAD
#define N4_HEAD 16
#endif
#ifndef N4_PRIO
#define N4_PRIO 0
#endif
namespace g5 {

constexpr int BM = 256, BN = 256, BKS = 32;
constexpr int HALF = 16384, STAGE = 2 * HALF, NSTAGE = 4, NTH = 256;
constexpr int STG_WAVE = 16 * 128 * 4;                          // 16 rows x 128 f32 per wave
constexpr int LDS_BYTES = NSTAGE * STAGE + 4 * STG_WAVE;        // 131072 + 32768 = all of the CU's LDS
constexpr int F_RES = 2, F_AUXOUT = 4;

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
    int tiles_m, tiles_n, nks, n_items, col_group;
    int desync;                // > 0: blocks with fewer items than the longest list start up to this many 0.5-us units late (spreads the epilogues' store bursts)
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

struct Cur {                   // position of the DMA stream: one stage of this block's item list (wave-uniform)
    int j, ks;
    const char* pa;            // A + ((m0 + 64 wave) * lda + 32 ks) * 2
    const char* pb;            // B + ((n0 + 64 wave) * ldb + 32 ks) * 2
};

__device__ __forceinline__ void item_origin(const Args& a, int item, int64_t& m0, int64_t& n0) {
    int tm, tn;
    if (a.col_group > 0) {     // tile columns walked in groups of col_group inside the item order (weight panels stay L2-resident)
        const int per = a.tiles_m * a.col_group;
        const int g = item / per, rem = item - g * per;
        const int c0 = g * a.col_group;
        const int w = a.tiles_n - c0 < a.col_group ? a.tiles_n - c0 : a.col_group;
        tm = rem / w;
        tn = c0 + rem - tm * w;
    } else {
        tm = item / a.tiles_n;
        tn = item - tm * a.tiles_n;
    }
    m0 = (int64_t)tm * BM;
    n0 = (int64_t)tn * BN;
}

__device__ __forceinline__ void seat(const Args& a, int first_item, int stride, int wave, Cur& c) {
    int64_t m0, n0;
    item_origin(a, first_item + c.j * stride, m0, n0);
    c.pa = reinterpret_cast<const char*>(a.A + (m0 + 64 * wave) * a.lda);
    c.pb = reinterpret_cast<const char*>(a.B + (n0 + 64 * wave) * a.ldb);
}

__device__ __forceinline__ void advance(const Args& a, int first_item, int stride, int n_mine, int wave, Cur& c) {
    if (c.ks + 1 < a.nks) {
        ++c.ks;
        c.pa += (N4_EXP & 64) ? ((c.ks * 2 == a.nks) ? -(int64_t)(a.nks - 2) * 64 : 128) : 2 * BKS;
        c.pb += (N4_EXP & 64) ? ((c.ks * 2 == a.nks) ? -(int64_t)(a.nks - 2) * 64 : 128) : 2 * BKS;
    } else if (c.j + 1 < n_mine) {
        ++c.j;
        c.ks = 0;
        seat(a, first_item, stride, wave, c);
    }                           // past the end of the stream: the last stage is fetched again (never read; keeps the vmcnt arithmetic uniform)
}

// fragment read f (0..7: A row tile f, 8..15: B row tile f - 8) sits behind MFMA f * N4_RD_STEP; DMA piece p behind MFMA N4_DMA_FIRST + p * N4_DMA_STEP
__host__ __device__ constexpr int read_at(int i) { return (i % N4_RD_STEP == 0 && i / N4_RD_STEP < 16) ? i / N4_RD_STEP : -1; }
__host__ __device__ constexpr int piece_at(int i) {
    return (i >= N4_DMA_FIRST && (i - N4_DMA_FIRST) % N4_DMA_STEP == 0 && (i - N4_DMA_FIRST) / N4_DMA_STEP < 8) ? (i - N4_DMA_FIRST) / N4_DMA_STEP : -1;
}
static_assert(15 * N4_RD_STEP < 64 && N4_DMA_FIRST + 7 * N4_DMA_STEP < 64, "fillers must fit the stage's 64 MFMAs");

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
    if (a.desync > 0 && n_mine < (a.n_items + G - 1) / G) {
        const int d = (int)(((uint32_t)bid * 2654435761u >> 16) % (uint32_t)a.desync);
        for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(16);
    }

    const uint32_t smem_off = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // staging: lane -> row sr = lane >> 2 of a 16-row piece, physical chunk lane & 3 = logical chunk ^ g[(sr >> 2) & 3]
    const int sr = lane >> 2;
    const uint32_t slc = (uint32_t)((lane & 3) ^ ((0 - (sr >> 2)) & 3));
    uint32_t voffA[4], voffB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        voffA[j] = (uint32_t)(sr + 16 * j) * (uint32_t)a.lda * 2u + slc * 16u;
        voffB[j] = (uint32_t)(sr + 16 * j) * (uint32_t)a.ldb * 2u + slc * 16u;
        if constexpr ((N4_EXP & 64) != 0) {       // 8 rows x 128 bytes per piece (whole lines; the data is not the tile's)
            voffA[j] = (uint32_t)((lane >> 3) + 8 * j) * (uint32_t)a.lda * 2u + (uint32_t)(lane & 7) * 16u;
            voffB[j] = (uint32_t)((lane >> 3) + 8 * j) * (uint32_t)a.ldb * 2u + (uint32_t)(lane & 7) * 16u;
        }
    }
    const uint32_t wave_lds = smem_off + (uint32_t)wave * 4096u;          // this wave's rows 64 w .. 64 w + 63 of a half
    // piece p of a stage: p even -> A rows 16 (p >> 1), p odd -> B rows 16 (p >> 1) of this wave's 64
#define ISSUE_PIECE(P, c, slot_off)                                                                                                   \
    do {                                                                                                                              \
        const char* src_ = ((P) & 1) ? (c).pb + voffB[(P) >> 1] : (c).pa + voffA[(P) >> 1];                                           \
        if constexpr ((N4_EXP & 32) != 0) src_ = reinterpret_cast<const char*>(a.B) + lane * 16;                                      \
        __builtin_amdgcn_global_load_lds(GLB_PTR(src_),                                                                               \
                                         (__attribute__((address_space(3))) void*)(size_t)(wave_lds + (slot_off) + ((P) & 1) * HALF + ((P) >> 1) * 1024), \
                                         16, 0, 0);                                                                                   \
    } while (0)

    Cur cD;                          // the DMA stream, four stages ahead of the MFMAs
    cD.j = 0; cD.ks = 0;
    seat(a, first_item, stride, wave, cD);
    // prologue: stages 0..3
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) {
        static_for<0, 8>([&](auto pc) { ISSUE_PIECE(decltype(pc)::value, cD, (uint32_t)(s * STAGE)); });
        advance(a, first_item, stride, n_mine, wave, cD);
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // stages 0, 1 have landed (this wave's pieces)
    SEG_BARRIER();

    // fragment addresses: row r = lane & 15 of a 16-row tile, logical chunk lane >> 4
    const int fr = lane & 15;
    const uint32_t fpc = (uint32_t)((lane >> 4) ^ ((0 - (fr >> 2)) & 3));
    uint32_t laA = smem_off + (uint32_t)wr * 8192u + (uint32_t)fr * 64u + fpc * 16u;
    uint32_t laB = smem_off + (uint32_t)HALF + (uint32_t)wc * 8192u + (uint32_t)fr * 64u + fpc * 16u;

    bf16x8 fa[2][8], fb[2][8];      // [stage parity][16-row tile]
#define RD_FRAG(buf, f, ra, rb)                                              \
    do {                                                                     \
        if constexpr ((f) < 8) fa[buf][(f) & 7] = lds_b128<((f) & 7) * 1024>(ra); \
        else fb[buf][(f) & 7] = lds_b128<((f) & 7) * 1024>(rb);              \
    } while (0)
#define WAIT_FRAGS(buf)                                                                                                              \
    do {                                                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                                          \
                     : "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fa[buf][2]), "+v"(fa[buf][3]), "+v"(fa[buf][4]), "+v"(fa[buf][5]),   \
                       "+v"(fa[buf][6]), "+v"(fa[buf][7]), "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fb[buf][2]), "+v"(fb[buf][3]),   \
                       "+v"(fb[buf][4]), "+v"(fb[buf][5]), "+v"(fb[buf][6]), "+v"(fb[buf][7]));                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                           \
    } while (0)
    static_for<0, 16>([&](auto fc) { constexpr int f = decltype(fc)::value; RD_FRAG(0, f, laA, laB); });
    WAIT_FRAGS(0);
    SEG_BARRIER();                   // every wave has read stage 0: its slot may take stage 4

    f32x4 acc[64];                   // [mi * 8 + ni]
    int t = 0;                       // stage counter of the whole stream (ring slot = t & 3)
    if constexpr (N4_PRIO) __builtin_amdgcn_s_setprio(N4_PRIO);

    // one stage: 64 MFMAs on buffer P; FIRST: the accumulators start from zero
    int j = 0;                       // item index in this block's list
    [[maybe_unused]] char* dst_base = reinterpret_cast<char*>(a.C);      // (diagnostic 512 / 1024) this lane's first store address of the previous item
    auto stage = [&](auto ptag, auto first_tag, [[maybe_unused]] int ks) {
        constexpr int P = decltype(ptag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint32_t rd_off = (uint32_t)((t + 1) & 3) * STAGE, wr_off = (uint32_t)(t & 3) * STAGE;
        uint32_t ra = laA + rd_off, rb = laB + rd_off;
        asm volatile("" : "+v"(ra), "+v"(rb));
        static_for<0, 64>([&](auto ic) {
            constexpr int i = decltype(ic)::value, mi = i >> 3, ni = i & 7;
            if constexpr (FIRST) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[P][ni], fa[P][mi], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[P][ni], fa[P][mi], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (read_at(i) >= 0 && !(N4_EXP & 8)) RD_FRAG(P ^ 1, read_at(i), ra, rb);
            if constexpr (piece_at(i) >= 0 && !(N4_EXP & 2)) ISSUE_PIECE(piece_at(i), cD, wr_off);
            if constexpr ((N4_EXP & (512 | 1024)) != 0 && (i == 21 || i == 45)) {
                if (ks < N4_HEAD && j > 0) {
                    const int k = ks * (32 / N4_HEAD) + (i == 45);                     // store k of 32
                    char* d = (N4_EXP & 512) ? dst_base + ((int64_t)(16 * (k & 7)) * a.ldc + 32 * (k >> 3)) * 2
                                             : dst_base + ((int64_t)(16 * (k >> 2) + 4 * (k & 3)) * a.ldc) * 2;
                    *reinterpret_cast<bf16x8*>(d) = fa[P][i == 45];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        advance(a, first_item, stride, n_mine, wave, cD);
        ++t;
        WAIT_FRAGS(P ^ 1);
        if constexpr (!(N4_EXP & 1)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // this wave's pieces of stage t + 1 have landed
        if constexpr (!(N4_EXP & 4)) SEG_BARRIER();
    };
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};

    for (; j < n_mine; ++j) {
        stage(P0, std::true_type{}, 0);
        for (int q = 1; q < (a.nks >> 1); ++q) {
            stage(P1, std::false_type{}, 2 * q - 1);
            stage(P0, std::false_type{}, 2 * q);
        }
        stage(P1, std::false_type{}, a.nks - 1);
        // the accumulators stay in the accumulator file until the epilogue reads them: left to itself hipcc moves a few of them to spare
        // VGPRs right behind their last MFMA (s_nop 7 + four v_accvgpr_read each, inside the stage)
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("" : "+a"(acc[i]));

        // ---- epilogue of this item
        if constexpr ((N4_EXP & 16) != 0) {
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("" ::"v"(acc[i]));
        } else {
            int64_t m0, n0;
            item_origin(a, first_item + j * stride, m0, n0);
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            if constexpr ((N4_EXP & 512) != 0)
                dst_base = reinterpret_cast<char*>(a.C + (m0 + wr * 128 + (lane_e & 15)) * a.ldc + n0 + wc * 128 + 8 * ((((lane_e >> 4) & 1) << 1) | (lane_e >> 5)));
            if constexpr ((N4_EXP & 1024) != 0)
                dst_base = reinterpret_cast<char*>(a.C + (m0 + wr * 128 + (lane_e >> 4)) * a.ldc + n0 + wc * 128 + 8 * (lane_e & 15));
            constexpr bool NEED_AUX = EPI == LECCR_EPI_DQUICKGELU || EPI == LECCR_EPI_DGELU;
            constexpr bool PRE = NEED_AUX || (FLAGS & F_RES) != 0;
          if constexpr (N4_DIRECT) {
            // No LDS: the accumulator tiles 2 p and 2 p + 1 of a strip hold, per lane (row l & 15, q = l >> 4), columns 32 p + 4 q .. + 3 and
            // 32 p + 16 + 4 q .. + 3.  v_permlane16_swap (odd 16-lane rows of the first operand <-> even rows of the second) on the register
            // pairs leaves every lane with 8 CONSECUTIVE columns of its row: 32 p + 8 sigma(q), sigma = {0, 2, 1, 3} - one 16-byte store per
            // lane and tile pair, 16 rows x 64 contiguous bytes per store instruction, and the epilogue arithmetic / aux / residual
            // accesses in that same layout.
            const int erow = lane_e & 15, eq = lane_e >> 4;
            const int64_t ncol = n0 + wc * 128 + 8 * (((eq & 1) << 1) | (eq >> 1));
            const int64_t mrow = m0 + wr * 128 + erow;
            // tile pair by tile pair (outer), strip by strip (inner): 8 bias values and one prefetched aux / residual vector live at a time
            [[maybe_unused]] bf16x8 xpre[2];
            auto load_pre = [&](int it) {        // it = p * 8 + mi
                const bf16_t* pb = NEED_AUX ? a.aux : a.residual;
                const int64_t pld = NEED_AUX ? a.ldaux : a.ldr;
                return *reinterpret_cast<const bf16x8*>(pb + (mrow + 16 * (it & 7)) * pld + ncol + 32 * (it >> 3));
            };
            if constexpr (PRE) xpre[0] = load_pre(0);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
                if (a.bias) { b0 = load4(a.bias + ncol + 32 * p); b1 = load4(a.bias + ncol + 32 * p + 4); }
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) {
                    constexpr int dummy = 0; (void)dummy;
                    const int it = p * 8 + mi;
                    if constexpr (PRE) {
                        if (it + 1 < 32) xpre[(it + 1) & 1] = load_pre(it + 1);
                    }
                    const int64_t m = mrow + 16 * mi;
                    f32x4 v0 = acc[mi * 8 + 2 * p], v1 = acc[mi * 8 + 2 * p + 1];
                    if constexpr (!(N4_EXP & 256)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // (through named floats: __builtin_bit_cast applied to a vector ELEMENT is miscompiled by hipcc 7.2 - every lane of
                            // the vector becomes element 0)
                            const float x_ = v0[e], y_ = v1[e];
                            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x_), __float_as_uint(y_), false, false);
                            v0[e] = __uint_as_float(sw[0]);
                            v1[e] = __uint_as_float(sw[1]);
                        }
                    }
                    v0 = v0 * a.alpha + b0;
                    v1 = v1 * a.alpha + b1;
                    const int64_t off = m * a.ldc + ncol + 32 * p;
                    if constexpr ((FLAGS & F_AUXOUT) != 0) {
                        bf16x8 pv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pv[e] = (bf16_t)v0[e]; pv[4 + e] = (bf16_t)v1[e]; }
                        *reinterpret_cast<bf16x8*>(a.aux_out + off) = pv;
                    }
                    if constexpr (EPI != LECCR_EPI_NONE) {
                        f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
                        if constexpr (NEED_AUX) {
                            const bf16x8 xv = xpre[it & 1];
#pragma unroll
                            for (int e = 0; e < 4; ++e) { x0[e] = (float)xv[e]; x1[e] = (float)xv[4 + e]; }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = apply_epi<EPI>(v0[e], x0[e]); v1[e] = apply_epi<EPI>(v1[e], x1[e]); }
                    }
                    if constexpr ((FLAGS & F_RES) != 0) {
                        bf16x8 rv;
                        if constexpr (!NEED_AUX) rv = xpre[it & 1];
                        else rv = *reinterpret_cast<const bf16x8*>(a.residual + m * a.ldr + ncol + 32 * p);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] += (float)rv[e]; v1[e] += (float)rv[4 + e]; }
                    }
                    bf16x8 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ov[e] = (bf16_t)v0[e]; ov[4 + e] = (bf16_t)v1[e]; }
                    if constexpr ((N4_EXP & 128) != 0) { asm volatile("" ::"v"(ov)); }
                    else *reinterpret_cast<bf16x8*>(a.C + off) = ov;
                    __builtin_amdgcn_sched_barrier(0);          // one tile pair at a time: the unrolled strips must not pile their accumulator reads up
                }
            }
          } else {
            const uint32_t stg = smem_off + NSTAGE * STAGE + (uint32_t)wave * STG_WAVE;
            // write side: accumulator row 16 mi + (lane & 15), 16-byte chunk 4 ni + (lane >> 4) of the strip's 128 columns
            const int wrow = lane_e & 15, wq = lane_e >> 4;
            const uint32_t wbase = stg + (uint32_t)wrow * 512u;
            // read side: 16 consecutive lanes on ONE row (256 contiguous bytes of C, 4 rows per store instruction):
            // lane -> row 4 q + (lane >> 4) of the strip (q = 0..3), columns 8 (lane & 15) .. + 7
            const int rq = lane_e >> 4, rp = lane_e & 15;
            const int64_t ncol = n0 + wc * 128 + rp * 8;
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (a.bias) { b0 = load4(a.bias + ncol); b1 = load4(a.bias + ncol + 4); }
            // the saved pre-activation (derivative epilogues) / the residual: loaded one strip AHEAD
            [[maybe_unused]] bf16x8 xpre[2][4];
            auto load_pre = [&](int mi, bf16x8* dst) {
                const bf16_t* pb = NEED_AUX ? a.aux : a.residual;
                const int64_t pld = NEED_AUX ? a.ldaux : a.ldr;
#pragma unroll
                for (int h8 = 0; h8 < 4; ++h8) {
                    const int64_t mm = m0 + wr * 128 + mi * 16 + 4 * h8 + rq;
                    dst[h8] = *reinterpret_cast<const bf16x8*>(pb + mm * pld + ncol);
                }
            };
            if constexpr (PRE) load_pre(0, xpre[0]);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                if constexpr (PRE) {
                    if (mi + 1 < 8) load_pre(mi + 1, xpre[(mi + 1) & 1]);
                }
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) lds_w128(wbase + (uint32_t)(((4 * ni + wq) ^ wrow) << 4), acc[mi * 8 + ni]);
                f32x4 v[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {                                  // row 4 q + rq, chunks 2 rp and 2 rp + 1
                    const int row = 4 * q + rq;
                    v[2 * q] = lds_f128<0>(stg + (uint32_t)row * 512u + (uint32_t)(((2 * rp) ^ row) << 4));
                    v[2 * q + 1] = lds_f128<0>(stg + (uint32_t)row * 512u + (uint32_t)(((2 * rp + 1) ^ row) << 4));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] = v[2 * q] * a.alpha + b0;
                    v[2 * q + 1] = v[2 * q + 1] * a.alpha + b1;
                }
#pragma unroll
                for (int h8 = 0; h8 < 4; ++h8) {                              // 4 rows x (8 columns = 16 bytes of bf16)
                    f32x4 v0 = v[2 * h8], v1 = v[2 * h8 + 1];
                    const int64_t m = m0 + wr * 128 + mi * 16 + 4 * h8 + rq;
                    const int64_t off = m * a.ldc + ncol;
                    if constexpr ((FLAGS & F_AUXOUT) != 0) {
                        bf16x8 pv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pv[e] = (bf16_t)v0[e]; pv[4 + e] = (bf16_t)v1[e]; }
                        *reinterpret_cast<bf16x8*>(a.aux_out + off) = pv;
                    }
                    if constexpr (EPI != LECCR_EPI_NONE) {
                        f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
                        if constexpr (NEED_AUX) {
                            const bf16x8 xv = xpre[mi & 1][h8];
#pragma unroll
                            for (int e = 0; e < 4; ++e) { x0[e] = (float)xv[e]; x1[e] = (float)xv[4 + e]; }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = apply_epi<EPI>(v0[e], x0[e]); v1[e] = apply_epi<EPI>(v1[e], x1[e]); }
                    }
                    if constexpr ((FLAGS & F_RES) != 0) {
                        bf16x8 rv;
                        if constexpr (!NEED_AUX) rv = xpre[mi & 1][h8];
                        else rv = *reinterpret_cast<const bf16x8*>(a.residual + m * a.ldr + ncol);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] += (float)rv[e]; v1[e] += (float)rv[4 + e]; }
                    }
                    bf16x8 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ov[e] = (bf16_t)v0[e]; ov[4 + e] = (bf16_t)v1[e]; }
                    if constexpr ((N4_EXP & 128) != 0) { asm volatile("" ::"v"(ov)); }
                    else *reinterpret_cast<bf16x8*>(a.C + off) = ov;
                }
            }
          }
            // every ordinary load of the epilogue retired, visibly to hipcc's waitcnt pass (no vmcnt(0) sneaks into the K loop)
            if constexpr (PRE) __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-fetched tail stages still target this block's LDS
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

// whole tiles only (the step's shapes: M = 256 x tokens, N = 768 .. 4096, K = 768 .. 4096): everything else stays on the 8-wave kernel
bool gemm256n4_legal(const leccr_gemm_params& p) {
    if (p.trans_a != 0 || p.trans_b != 0 || p.c_dtype != LECCR_BF16 || p.ab_dtype != LECCR_BF16 || p.accumulate || p.c3 || p.c8 || p.colsum) return false;
    if (p.M < 256 || p.N < 256 || (p.M & 255) || (p.N & 255) || (p.K & 63) || p.K < 128) return false;
    if ((p.ldc & 7) || (p.ldr & 7) || (p.ldaux & 7) || (p.lda & 7) || (p.ldb & 7)) return false;
    if ((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.C) & 15)) return false;
    if (p.lda >= (1 << 23) || p.ldb >= (1 << 23)) return false;      // 32-bit (row * ld * 2) in the staging offsets
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
    a.nks = (int)(p.K / BKS);
    a.n_items = a.tiles_m * a.tiles_n;
    static const int env_group = [] { const char* e = getenv("LECCR_N4_COLGROUP"); return e ? atoi(e) : 0; }();
    a.col_group = env_group > 0 && env_group < a.tiles_n ? env_group : 0;
    static const int env_desync = [] { const char* e = getenv("LECCR_N4_DESYNC"); return e ? atoi(e) : 0; }();
    a.desync = env_desync;
    const int grid = a.n_items < cap ? a.n_items : cap;
    const int f = n4_flags(p);
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
}

}  // namespace leccr
