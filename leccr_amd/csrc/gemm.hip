// gemm.hip - GEMMs of the LECCR hot path for gfx950.
//
//   C[M,N] = epilogue(alpha * sum_k A(m,k) B(n,k))
//
// Two families:
//   * gemm_generic_kernel<TA,TC>: any layout / alignment / dtype, f32 FMA arithmetic.  The f32 parity path and the
//     fallback for shapes the MFMA kernels reject.
//   * gemm_mfma_kernel<TRANS,...>: bf16 operands, v_mfma_f32_16x16x32_bf16, 128x128x64 block tile, 4 waves (2x2),
//     each wave 64x64; operands go HBM -> LDS with global_load_lds_dwordx4 (no VGPR staging) into a double-buffered
//     XOR-swizzled image (swizzle applied on the SOURCE address, LDS image stays lane-linear); fragments are read
//     with ds_read_b128 (K-contiguous operands, "NT") or ds_read_b64_tr_b16 (reduction-major operands, "TN" = the
//     weight-gradient product dW = dY^T X, where neither operand is K-contiguous).
//     The MFMA is issued with swapped roles (D' = B A^T) so each lane ends up with 4 consecutive N of one row:
//     8-byte (bf16) / 16-byte (f32) epilogue accesses for C, bias, aux and residual.
//
// Reference op sequences replaced: see include/leccr_hip.h (leccr_gemm).
#include "common.h"
#include <stdlib.h>

namespace leccr {

// ------------------------------------------------------------------------------------------------ generic
template <typename TA, typename TC, int EPI>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const TA* __restrict__ A, const TA* __restrict__ B,
                                                           TC* __restrict__ C, int64_t M, int64_t N, int64_t K,
                                                           int64_t sam, int64_t sak, int64_t sbn, int64_t sbk,
                                                           int64_t ldc, const float* __restrict__ bias,
                                                           const TC* __restrict__ aux, int64_t ldaux,
                                                           TC* __restrict__ aux_out, const TC* __restrict__ residual,
                                                           int64_t ldr, float alpha, int accumulate) {
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ float As[BK][BM + 1];
    __shared__ float Bs[BK][BN + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;  // 16 x 16 threads, 4x4 outputs each
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    float acc[4][4] = {};
    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        // 64x16 elements per operand, 256 threads -> 4 each
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            int r, kk;
            // pick the mapping that makes consecutive threads touch consecutive memory
            if (sak == 1) { kk = e & 15; r = e >> 4; } else { r = e & 63; kk = e >> 6; }
            const int64_t m = m0 + r, k = k0 + kk;
            As[kk][r] = (m < M && k < K) ? to_f32(A[m * sam + k * sak]) : 0.f;
            if (sbk == 1) { kk = e & 15; r = e >> 4; } else { r = e & 63; kk = e >> 6; }
            const int64_t n = n0 + r, kb = k0 + kk;
            Bs[kk][r] = (n < N && kb < K) ? to_f32(B[n * sbn + kb * sbk]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = alpha * acc[i][j];
            if (bias) v += bias[n];
            if constexpr (epi_saves_deriv<EPI>) {
                const ActD t_ = apply_epi_d<EPI>(v);
                v = t_.a;
                aux_out[m * ldc + n] = from_f32<TC>(t_.d);
            } else {
                if (aux_out) aux_out[m * ldc + n] = from_f32<TC>(v);
                float ax = 0.f;
                if constexpr (epi_needs_aux<EPI>) ax = to_f32(aux[m * ldaux + n]);
                v = apply_epi<EPI>(v, ax);
            }
            if (residual) v += to_f32(residual[m * ldr + n]);
            if (accumulate) v += to_f32(C[m * ldc + n]);
            C[m * ldc + n] = from_f32<TC>(v);
        }
    }
}

// ------------------------------------------------------------------------------------------------ MFMA bf16
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[512];  // source for out-of-range tile rows

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kStageBytes = (BM + BN) * BK * 2;  // 32 KiB per pipeline stage

// XCD-aware, bijective remap of the linear block id: blocks that share an XCD (same id % 8) get a contiguous range
// of tiles, so neighbouring tiles (same A row panel) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// f(k) for the reduction-major ("TN") image: 8 row pieces read by one 32-lane half of ds_read_b64_tr_b16 are rows
// {0..3, 8..11} (+4, +16 variants); mapping them onto 8 distinct 32-byte blocks of the 256-byte bank row.
__device__ __forceinline__ int tn_swz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Issue one K-tile (64 deep) of both operands HBM -> LDS.
//  TRANS = 0 (NT): A [M,K], B [N,K], rows of 64 bf16 = 128 B = 8 chunks of 16 B; LDS image [128 rows][8 chunks],
//                  physical chunk = logical chunk ^ (row & 7).
//  TRANS = 1 (TN): A [K,M], B [K,N], tile [64 k][128 cols] = 256-B rows = 16 chunks; physical 32-B block =
//                  logical block ^ tn_swz(k).
template <int TRANS>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                           unsigned char* lds_stage, int64_t m0, int64_t n0, int64_t k0, int64_t M,
                                           int64_t N, int64_t K, int64_t lda, int64_t ldb) {
    const int tid = threadIdx.x;
    // wave-uniform in an SGPR: otherwise hipcc wraps every global_load_lds in a waterfall loop (M0 must be uniform)
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    unsigned char* ldsA = lds_stage;
    unsigned char* ldsB = lds_stage + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = i * 256 + tid;  // 16-byte slot index in the linear LDS image
        const bf16_t *srcA, *srcB;
        if constexpr (TRANS == 0) {
            const int row = slot >> 3, pc = slot & 7, lc = pc ^ (row & 7);
            int64_t m = m0 + row, n = n0 + row;
            m = m < M ? m : M - 1;  // clamped rows produce values that are never stored
            n = n < N ? n : N - 1;
            srcA = A + m * lda + k0 + lc * 8;
            srcB = B + n * ldb + k0 + lc * 8;
        } else {
            const int k = slot >> 4, pc = slot & 15;
            const int lc = (((pc >> 1) ^ tn_swz(k)) << 1) | (pc & 1);
            const int64_t kk = k0 + k;
            const int64_t m = m0 + lc * 8, n = n0 + lc * 8;
            const bf16_t* z = reinterpret_cast<const bf16_t*>(g_zero_page);
            srcA = (kk < K && m < M) ? A + kk * lda + m : z;  // reduction tail / column tail read zeros
            srcB = (kk < K && n < N) ? B + kk * ldb + n : z;
        }
        // LDS destination: wave-uniform base, the hardware adds lane*16
        unsigned char* dA = ldsA + (size_t)(i * 256 + wave_base) * 16;
        unsigned char* dB = ldsB + (size_t)(i * 256 + wave_base) * 16;
        __builtin_amdgcn_global_load_lds(GLB_PTR(srcA), LDS_PTR(dA), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(srcB), LDS_PTR(dB), 16, 0, 0);
    }
}

template <int TRANS>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* tile, int row0, int ks, int lane) {
    if constexpr (TRANS == 0) {
        // lane holds X[row0 + (lane&15)][k = ks*32 + 8*(lane>>4) + 0..7]
        const int row = row0 + (lane & 15);
        const int lc = ks * 4 + (lane >> 4);
        const int off = row * 128 + ((lc ^ (row & 7)) << 4);
        return *reinterpret_cast<const bf16x8*>(tile + off);
    } else {
        // tile[k][col]: lane needs col = row0 + (lane&15), k = ks*32 + 8*(lane>>4) + 0..7 -> two transposed reads of a
        // 4(k) x 16(col) block each; lane i of a 16-lane group supplies the address of block row i>>2, cols 4*(i&3)..+3
        const int i = lane & 15, h = lane >> 4;
        const int kA = ks * 32 + h * 8 + (i >> 2);
        const int col = row0 + 4 * (i & 3);  // element column
        const int blk = col >> 4, inblk = (col & 15) * 2;
        const int offA = kA * 256 + ((blk ^ tn_swz(kA)) << 5) + inblk;
        const int kB = kA + 4;
        const int offB = kB * 256 + ((blk ^ tn_swz(kB)) << 5) + inblk;
        short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(tile + offA));
        short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(tile + offB));
        short8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int TRANS, typename TC, int EPI, bool ATOMIC>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                        TC* __restrict__ C, int64_t M, int64_t N, int64_t K,
                                                        int64_t lda, int64_t ldb, int64_t ldc,
                                                        const float* __restrict__ bias, const TC* __restrict__ aux,
                                                        int64_t ldaux, TC* __restrict__ aux_out,
                                                        const TC* __restrict__ residual, int64_t ldr, float alpha,
                                                        int accumulate, int tiles_n, int ktiles_per_split) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(tile / tiles_n) * BM, n0 = (int64_t)(tile % tiles_n) * BN;
    const int64_t ktiles_total = (K + BK - 1) / BK;
    const int64_t kt_begin = (int64_t)blockIdx.y * ktiles_per_split;
    int64_t kt_end = kt_begin + ktiles_per_split;
    if (kt_end > ktiles_total) kt_end = ktiles_total;
    const int nkt = (int)(kt_end - kt_begin);
    if (nkt <= 0) return;

    f32x4 acc[4][4];  // [ni][mi]; element r of acc[ni][mi] = C[m = mi*16 + (lane&15)][n = ni*16 + (lane>>4)*4 + r]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_tile<TRANS>(A, B, smem, m0, n0, kt_begin * BK, M, N, K, lda, ldb);
    for (int t = 0; t < nkt; ++t) {
        unsigned char* cur = smem + (t & 1) * kStageBytes;
        __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)) and orders every wave's tile t; also frees buffer t+1
        if (t + 1 < nkt)
            stage_tile<TRANS>(A, B, smem + ((t + 1) & 1) * kStageBytes, m0, n0, (kt_begin + t + 1) * BK, M, N, K, lda, ldb);
        const unsigned char* tA = cur;
        const unsigned char* tB = cur + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag<TRANS>(tA, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i] = read_frag<TRANS>(tB, wn * 64 + i * 16, ks, lane);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
        }
    }

    // epilogue: lane owns rows m = .. + (lane&15), 4 consecutive columns n = .. + (lane>>4)*4
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = m0 + wm * 64 + mi * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int64_t n = n0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
            if (n >= N) continue;  // N % 4 == 0 is required by the dispatcher
            f32x4 v = acc[ni][mi] * alpha;
            if constexpr (ATOMIC) {
                if (bias && blockIdx.y == 0) v += load4(bias + n);
                float* c = reinterpret_cast<float*>(C) + m * ldc + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(c + r, v[r]);
            } else {
                if (bias) v += load4(bias + n);
                if constexpr (epi_saves_deriv<EPI>) {
                    f32x4 dv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const ActD t_ = apply_epi_d<EPI>(v[r]); v[r] = t_.a; dv[r] = t_.d; }
                    store4(aux_out + m * ldc + n, dv);
                } else {
                    if (aux_out) store4(aux_out + m * ldc + n, v);
                    if constexpr (EPI != LECCR_EPI_NONE) {
                        f32x4 ax = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (epi_needs_aux<EPI>) ax = load4(aux + m * ldaux + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = apply_epi<EPI>(v[r], ax[r]);
                    }
                }
                if (residual) v += load4(residual + m * ldr + n);
                if (accumulate) v += load4(C + m * ldc + n);
                store4(C + m * ldc + n, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ dispatch
static bool g_force_generic = false;

template <typename TA, typename TC>
static int launch_generic(const leccr_gemm_params& p, hipStream_t st) {
    const int64_t sam = p.trans_a ? 1 : p.lda, sak = p.trans_a ? p.lda : 1;
    const int64_t sbn = p.trans_b ? 1 : p.ldb, sbk = p.trans_b ? p.ldb : 1;
    dim3 grid((unsigned)cdiv(p.N, 64), (unsigned)cdiv(p.M, 64));
#define LG(E)                                                                                                         \
    hipLaunchKernelGGL((gemm_generic_kernel<TA, TC, E>), grid, dim3(256), 0, st, (const TA*)p.A, (const TA*)p.B,      \
                       (TC*)p.C, p.M, p.N, p.K, sam, sak, sbn, sbk, p.ldc, p.bias, (const TC*)p.aux, p.ldaux,        \
                       (TC*)p.aux_out, (const TC*)p.residual, p.ldr, p.alpha, p.accumulate)
    switch (p.epilogue) {
        case LECCR_EPI_NONE: LG(LECCR_EPI_NONE); break;
        case LECCR_EPI_QUICKGELU: LG(LECCR_EPI_QUICKGELU); break;
        case LECCR_EPI_GELU: LG(LECCR_EPI_GELU); break;
        case LECCR_EPI_DQUICKGELU: LG(LECCR_EPI_DQUICKGELU); break;
        case LECCR_EPI_DGELU: LG(LECCR_EPI_DGELU); break;
        case LECCR_EPI_QUICKGELU_D: LG(LECCR_EPI_QUICKGELU_D); break;
        case LECCR_EPI_GELU_D: LG(LECCR_EPI_GELU_D); break;
        case LECCR_EPI_MULAUX: LG(LECCR_EPI_MULAUX); break;
        default: set_error("gemm: bad epilogue %d", p.epilogue); return LECCR_EINVAL;
    }
#undef LG
    return check_launch("gemm_generic");
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int TRANS, typename TC>
static int launch_mfma(const leccr_gemm_params& p, hipStream_t st) {
    const int tiles_m = (int)cdiv(p.M, BM), tiles_n = (int)cdiv(p.N, BN);
    const int64_t ktiles = cdiv(p.K, BK);
    int splits = 1;
    if (p.accumulate && sizeof(TC) == 4 && p.epilogue == LECCR_EPI_NONE && !p.residual && !p.aux_out) {
        // weight-gradient shape: few output tiles, very deep K -> split K over blocks, f32 atomics into C
        const int64_t tiles = (int64_t)tiles_m * tiles_n;
        if (tiles < 512 && ktiles >= 16) {
            splits = (int)((1024 + tiles - 1) / tiles);
            if (splits > ktiles / 4) splits = (int)(ktiles / 4);
            if (splits < 1) splits = 1;
        }
    }
    const int kps = (int)cdiv(ktiles, splits);
    splits = (int)cdiv(ktiles, kps);
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)splits);
    const size_t shmem = 2 * kStageBytes;
#define LM(E, AT)                                                                                                     \
    do {                                                                                                              \
        auto kfn = gemm_mfma_kernel<TRANS, TC, E, AT>;                                                                \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);      \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL(kfn, grid, dim3(256), shmem, st, (const bf16_t*)p.A, (const bf16_t*)p.B, (TC*)p.C, p.M,    \
                           p.N, p.K, p.lda, p.ldb, p.ldc, p.bias, (const TC*)p.aux, p.ldaux, (TC*)p.aux_out,          \
                           (const TC*)p.residual, p.ldr, p.alpha, p.accumulate, tiles_n, kps);                        \
    } while (0)
    if (splits > 1) {
        if constexpr (sizeof(TC) == 4) LM(LECCR_EPI_NONE, true);
    } else {
        switch (p.epilogue) {
            case LECCR_EPI_NONE: LM(LECCR_EPI_NONE, false); break;
            case LECCR_EPI_QUICKGELU: LM(LECCR_EPI_QUICKGELU, false); break;
            case LECCR_EPI_GELU: LM(LECCR_EPI_GELU, false); break;
            case LECCR_EPI_DQUICKGELU: LM(LECCR_EPI_DQUICKGELU, false); break;
            case LECCR_EPI_DGELU: LM(LECCR_EPI_DGELU, false); break;
            case LECCR_EPI_QUICKGELU_D: LM(LECCR_EPI_QUICKGELU_D, false); break;
            case LECCR_EPI_GELU_D: LM(LECCR_EPI_GELU_D, false); break;
            case LECCR_EPI_MULAUX: LM(LECCR_EPI_MULAUX, false); break;
            default: set_error("gemm: bad epilogue %d", p.epilogue); return LECCR_EINVAL;
        }
    }
#undef LM
    return check_launch("gemm_mfma");
}

static bool mfma_ok(const leccr_gemm_params& p) {
    if (g_force_generic || p.ab_dtype != LECCR_BF16) return false;
    if (!(p.trans_a == p.trans_b)) return false;  // NT (0,0) and TN-gradient (1,1) only; mixed layouts -> generic
    if (!aligned16(p.A) || !aligned16(p.B) || !aligned16(p.C)) return false;
    if ((p.lda & 7) || (p.ldb & 7) || (p.N & 3)) return false;
    const int csz = p.c_dtype == LECCR_F32 ? 4 : 2;
    if ((p.ldc * csz) & (4 * csz - 1)) return false;
    if (p.aux && (!aligned16(p.aux) || (p.ldaux & 3))) return false;
    if (p.residual && (((uintptr_t)p.residual & (4 * csz - 1)) || (p.ldr & 3))) return false;
    if (p.aux_out && ((uintptr_t)p.aux_out & (4 * csz - 1))) return false;
    if (p.bias && !aligned16(p.bias)) return false;
    if (p.trans_a == 0) {
        if (p.K % BK) return false;  // K-contiguous operands: no K tail in the fast path
    } else {
        if ((p.M & 7) || (p.N & 7)) return false;
    }
    if (p.M < 1 || p.N < 1 || p.K < 1) return false;
    return true;
}

}  // namespace leccr

using namespace leccr;

namespace leccr {
extern bool g_force_generic_attn;
bool gemm256_wanted(const leccr_gemm_params& p);
bool gemm256_legal(const leccr_gemm_params& p);
bool gemm256_fuses_colsum(const leccr_gemm_params& p);
bool gemm256w4_fp8_legal(const leccr_gemm_params& p);
bool gemm256w4_c8_legal(const leccr_gemm_params& p);
bool gemm256_fuses_asum(const leccr_gemm_params& p);
int gemm256_launch(const leccr_gemm_params& p, hipStream_t st);
void gemm256_set_max_blocks(int n, int64_t launches);
int gemm256_max_blocks();
bool gemm256w4_legal(const leccr_gemm_params& p);
int gemm256w4_launch(const leccr_gemm_params& p, hipStream_t st);
bool gemm256t4_legal(const leccr_gemm_params& p);
int gemm256t4_launch(const leccr_gemm_params& p, int cap, hipStream_t st);
bool gemm256t4_fp8_legal(const leccr_gemm_params& p);
int gemm256t4_fp8_launch(const leccr_gemm_params& p, int cap, hipStream_t st);
bool gemm256n4_legal(const leccr_gemm_params& p);
int gemm256n4_launch(const leccr_gemm_params& p, int cap, hipStream_t st);
// Forward / dX (NT) products with whole 256 x 256 tiles and K >= 768 CAN run on the 4-wave 16x16x32 kernel of gemm256n4.hip (round 5):
// LECCR_GEMM_N4=1, or leccr_set_gemm_variant(5) for A/B runs and tests.  It is parity-green on every epilogue and carries the whole model, but
// measured on one box inside the step it is 3 % behind the 8-wave kernel (GEMM family 99.0 against 96.2 ms per step, 131.4 against 130.2 ms per
// step: profiles/r05_gemm_n4_step_ab.txt) - both structures end on the same per-CU L2 -> LDS path (profiles/r05_gemm_n4_v2.txt) - so the
// 8-wave kernel stays the default.
static int n4_default() {
    static const int v = [] { const char* e = getenv("LECCR_GEMM_N4"); return (e && e[0] == '1') ? 1 : 0; }();
    return v;
}
// Weight-gradient (TN) products: the 4-wave kernel of gemm256t4.hip wherever it is legal (default; LECCR_GEMM_TN4=0: the 8-wave TN
// instantiation of gemm256.hip instead).  leccr_set_gemm_variant(4) / (2) force one or the other for A/B runs and tests.
// Measured on the step's shapes (scripts/gemm_bench.py, same box): dW qkv 834 -> 1010 TFLOP/s, out 757 -> 818, fc1 960 -> 1076,
// fc2 1020 -> 1120, BERT dW fc1 618 -> 683, out 259 -> 292 (profiles/r03_gemm_tn4_vs_tn8.txt).
static int tn4_default() {
    static const int v = [] { const char* e = getenv("LECCR_GEMM_TN4"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}
// 0 auto, 1 force the 128x128 kernel, 2 force the 8-wave 256x256 kernel where legal, 3 force the 4-wave 256x256 kernel where legal
static int g_gemm_variant = 0;
static int g_last_kernel = 0;
}

namespace leccr { void gemm256_set_stamps(void* p); }
extern "C" int leccr_debug_set_stamps(void* p) {      // diagnostic builds (-DLECCR_G256_STAMPS) only
    leccr::gemm256_set_stamps(p);
    return LECCR_OK;
}

extern "C" int leccr_last_gemm_kernel(void) { return g_last_kernel; }

extern "C" int leccr_set_gemm_max_blocks(int n) {
    gemm256_set_max_blocks(n, -1);
    return gemm256_max_blocks();
}

extern "C" int leccr_get_gemm_max_blocks(void) { return gemm256_max_blocks(); }

extern "C" int leccr_set_gemm_max_blocks_for(int n, int64_t launches) {
    gemm256_set_max_blocks(n, launches < 1 ? 1 : launches);
    return gemm256_max_blocks();
}

extern "C" int leccr_set_gemm_variant(int v) {
    g_gemm_variant = v;
    return LECCR_OK;
}

extern "C" int leccr_set_force_generic(int on) {
    g_force_generic = on != 0;
    g_force_generic_attn = on != 0;
    return LECCR_OK;
}

static int gemm_dispatch(const leccr_gemm_params& p, hipStream_t st, bool* colsum_fused) {
    const double flops = 2.0 * (double)p.M * (double)p.N * (double)p.K;
    if (mfma_ok(p)) {
        ProfScope prof(p.trans_a == 0 ? PROF_GEMM_NT : PROF_GEMM_TN, flops, st);
        if (g_gemm_variant == 3 && gemm256w4_legal(p)) {
            g_last_kernel = 3;
            return gemm256w4_launch(p, st);
        }
        if ((g_gemm_variant == 5 || (g_gemm_variant == 0 && n4_default() && gemm256_wanted(p))) && gemm256n4_legal(p)) {
            g_last_kernel = 7;
            return gemm256n4_launch(p, gemm256_max_blocks(), st);
        }
        if ((g_gemm_variant == 4 || (g_gemm_variant == 0 && tn4_default())) && gemm256t4_legal(p)) {
            g_last_kernel = 5;
            return gemm256t4_launch(p, gemm256_max_blocks(), st);
        }
        if (((g_gemm_variant == 2 || g_gemm_variant == 3) && gemm256_legal(p)) || ((g_gemm_variant == 0 || g_gemm_variant == 4) && gemm256_wanted(p))) {
            *colsum_fused = gemm256_fuses_colsum(p);
            g_last_kernel = 2;
            return gemm256_launch(p, st);
        }
        g_last_kernel = 1;
        if (p.trans_a == 0)
            return p.c_dtype == LECCR_F32 ? launch_mfma<0, float>(p, st) : launch_mfma<0, bf16_t>(p, st);
        return p.c_dtype == LECCR_F32 ? launch_mfma<1, float>(p, st) : launch_mfma<1, bf16_t>(p, st);
    }
    ProfScope prof(PROF_GEMM_GENERIC, flops, st);
    g_last_kernel = 0;
    if (p.ab_dtype == LECCR_F32) {
        LECCR_REQUIRE(p.c_dtype == LECCR_F32, LECCR_EINVAL, "gemm: f32 operands need an f32 C");
        return launch_generic<float, float>(p, st);
    }
    return p.c_dtype == LECCR_F32 ? launch_generic<bf16_t, float>(p, st) : launch_generic<bf16_t, bf16_t>(p, st);
}

extern "C" int leccr_gemm(const leccr_gemm_params* pp, void* stream) {
    LECCR_REQUIRE(pp != nullptr, LECCR_EINVAL, "gemm: null params");
    const leccr_gemm_params& p = *pp;
    LECCR_REQUIRE(p.A && p.B && (p.C || (p.c8 && (p.ab_dtype == LECCR_FP8 || p.ab_dtype == LECCR_FP8_BF8A)) || p.c3), LECCR_EINVAL, "gemm: null operand");
    if (p.c3) {
        LECCR_REQUIRE(p.ab_dtype == LECCR_BF16 && p.c_dtype == LECCR_F32 && !p.trans_a && !p.trans_b && !p.accumulate && (p.c3_form == 0 || p.c3_form == 1) &&
                          (p.N & 3) == 0 && ((uintptr_t)p.c3 & 7) == 0,
                      LECCR_EINVAL, "gemm: c3 (split copy of C) needs bf16 operands, an f32 C, trans_a = trans_b = 0, no accumulate, N %% 4 == 0");
        LECCR_REQUIRE(p.C || (g_gemm_variant == 0 && mfma_ok(p) && gemm256_wanted(p)), LECCR_EINVAL,
                      "gemm: a null C beside c3 needs a shape the 8-wave 256x256 kernel takes (M, N >= 256, >= 192 tiles, N %% 8 == 0)");
    }
    LECCR_REQUIRE(p.M >= 0 && p.N >= 0 && p.K >= 0, LECCR_EINVAL, "gemm: negative size");
    // checks every operand format shares (they used to sit below the fp8 branch, which returned before reaching them)
    LECCR_REQUIRE(p.lda >= (p.trans_a ? p.M : p.K) && p.ldb >= (p.trans_b ? p.N : p.K) && p.ldc >= p.N, LECCR_EINVAL,
                  "gemm: leading dimension smaller than the row length");
    LECCR_REQUIRE(!(p.aux && p.ldaux < p.N) && !(p.residual && p.ldr < p.N), LECCR_EINVAL,
                  "gemm: ldaux / ldr smaller than the row length");
    if ((p.epilogue == LECCR_EPI_DQUICKGELU || p.epilogue == LECCR_EPI_DGELU || p.epilogue == LECCR_EPI_MULAUX))
        LECCR_REQUIRE(p.aux != nullptr, LECCR_EINVAL, "gemm: derivative epilogue needs aux");
    if ((p.epilogue == LECCR_EPI_QUICKGELU_D || p.epilogue == LECCR_EPI_GELU_D))
        LECCR_REQUIRE(p.aux_out != nullptr, LECCR_EINVAL, "gemm: the derivative-saving epilogues need aux_out");
    const bool fp8_ab = p.ab_dtype == LECCR_FP8 || p.ab_dtype == LECCR_FP8_BF8A;
    LECCR_REQUIRE(fp8_ab || (p.alpha_dev[0] == nullptr && p.alpha_dev[1] == nullptr), LECCR_EINVAL,
                  "gemm: alpha_dev (device-side operand scales) exists for fp8 operands only");
    if (fp8_ab && p.trans_a == 1 && p.trans_b == 1) {
        // fp8 weight-gradient product dW += alpha dY^T X: e5m2 dY [K, M] against e4m3 X [K, N], f32 C accumulated in place (gemm256t4.hip)
        LECCR_REQUIRE(gemm256t4_fp8_legal(p), LECCR_EINVAL, "gemm: fp8 TN operands need ab_dtype LECCR_FP8_BF8A, an accumulated f32 C, no epilogue, "
                      "M, N %% 256 == 0, K %% 128 == 0, K >= 2048 and 16-byte aligned rows");
        hipStream_t st8 = as_stream(stream);
        ProfScope prof(PROF_GEMM_FP8_TN, 2.0 * (double)p.M * (double)p.N * (double)p.K, st8);
        g_last_kernel = 6;
        return gemm256t4_fp8_launch(p, gemm256_max_blocks(), st8);
    }
    if (fp8_ab) {
        // fp8 MFMA GEMM inputs: only the 4-wave 256x256 kernel has them (no generic / small-shape fp8 path: callers keep those in bf16)
        LECCR_REQUIRE(gemm256w4_fp8_legal(p), LECCR_EINVAL, "gemm: fp8 operands need trans_a = trans_b = 0, a bf16 C, M, N >= 256, K %% 128 == 0, "
                      "16-byte aligned rows and a compiled epilogue");
        LECCR_REQUIRE(!p.asum, LECCR_EINVAL, "gemm: fp8 NT operands with asum");
        LECCR_REQUIRE(gemm256w4_c8_legal(p), LECCR_EINVAL, "gemm: c8 needs a site, 8-byte aligned rows and an (d-)activation epilogue of matching format; "
                      "colsum / a null C exist beside c8 only (colsum: the dX form)");
        if (p.M == 0 || p.N == 0) return LECCR_OK;
        hipStream_t st8 = as_stream(stream);
        ProfScope prof(PROF_GEMM_FP8_NT, 2.0 * (double)p.M * (double)p.N * (double)p.K, st8);
        g_last_kernel = 4;
        return gemm256w4_launch(p, st8);
    }
    LECCR_REQUIRE(p.ab_dtype == LECCR_F32 || p.ab_dtype == LECCR_BF16, LECCR_EINVAL, "gemm: bad ab_dtype");
    LECCR_REQUIRE(p.c8 == nullptr, LECCR_EINVAL, "gemm: c8 (fused fp8 copy of C) exists for fp8 operands only");
    LECCR_REQUIRE(p.c_dtype == LECCR_F32 || p.c_dtype == LECCR_BF16, LECCR_EINVAL, "gemm: bad c_dtype");
    LECCR_REQUIRE(!(p.accumulate && p.c_dtype != LECCR_F32), LECCR_EINVAL, "gemm: accumulate needs an f32 C");
    if (p.M == 0 || p.N == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    LECCR_REQUIRE(!(p.colsum && p.accumulate), LECCR_EINVAL, "gemm: colsum with accumulate is not defined");
    LECCR_REQUIRE(!(p.asum && !p.trans_a), LECCR_EINVAL, "gemm: asum is defined for trans_a = 1 only");
    LECCR_REQUIRE(!p.asum_x3 || (p.asum && p.ab_dtype == LECCR_BF16 && p.K % 3 == 0), LECCR_EINVAL,
                  "gemm: asum_x3 needs asum, bf16 operands (a 3-segment split copy read as [3 rows, M]) and K a multiple of 3");
    bool colsum_fused = false;
    int rc = gemm_dispatch(p, st, &colsum_fused);
    if (rc == LECCR_OK && p.c3 && g_last_kernel != 2)      // kernels without the split epilogue: leccr_split3 over the C just written
        rc = leccr_split3((const float*)p.C, p.c3, p.M, p.N, p.ldc, p.c3_form, 0, nullptr, stream);
    if (rc == LECCR_OK && p.colsum && !colsum_fused)     // paths without the fused epilogue: one pass over the C just written
        rc = leccr_colsum(p.C, p.colsum, p.M, p.N, p.ldc, p.c_dtype, stream);
    if (rc == LECCR_OK && p.asum && p.asum_x3 && g_last_kernel != 5) {
        // asum over the hi and lo segments of a 3-segment split copy (leccr_gemm_params.asum_x3) outside the 4-wave TN kernel: A [3 rows, M] read as
        // [rows, 3 M] - two passes over its first two column blocks
        if (p.alpha != 1.0f) { set_error("gemm: asum_x3 with alpha != 1 needs the 4-wave TN kernel"); return LECCR_EINVAL; }
        const char* a0 = (const char*)p.A;
        rc = leccr_colsum(a0, p.asum, p.K / 3, p.M, 3 * p.lda, p.ab_dtype, stream);
        if (rc == LECCR_OK) rc = leccr_colsum(a0 + 2 * p.lda, p.asum, p.K / 3, p.M, 3 * p.lda, p.ab_dtype, stream);
        return rc;
    }
    if (rc == LECCR_OK && p.asum && !(g_last_kernel == 5 || (g_last_kernel == 2 && gemm256_fuses_asum(p)))) {
        // kernels without the fused sum: one pass over A ([K, M] row-major), scaled like the product
        if (p.alpha == 1.0f) rc = leccr_colsum(p.A, p.asum, p.K, p.M, p.lda, p.ab_dtype, stream);
        else { set_error("gemm: asum with alpha != 1 needs the 256x256 kernel"); rc = LECCR_EINVAL; }
    }
    return rc;
}
