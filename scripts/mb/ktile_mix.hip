// Micro-benchmark of the 256x256x64 K-tile instruction mix on one CU-filling grid (diagnostic, not part of the library):
// per wave and iteration  N_MFMA matrix instructions + 24 ds_read_b128 + 8 LDS-DMA pieces + 2 barriers, 8 waves per block,
// one block per CU.  Variants isolate what bounds the loop.   hipcc --offload-arch=gfx950 -O3 ktile_mix.hip -o ktile_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) int int4_t;
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int IMM>
__device__ __forceinline__ bf16x8 lds_b128(uint32_t addr) {
    int4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
    return __builtin_bit_cast(bf16x8, v);
}

// MODE: 0 = 64 x mfma_16x16x32, 1 = 32 x mfma_32x32x16.   READS / DMA / BARS switch the fillers off.
// STAG: 0 = every wave issues its 8 pieces in filler slots 24..31; 1 = waves 0-3 use slots 8..15, waves 4-7 slots 24..31
//       (the two waves of a SIMD never issue pieces in the same stretch); 2 = all 8 pieces back to back at slot 24;
//       3 = register staging instead of LDS-DMA: 8 global_load_dwordx4 in slots 24..31, their 8 ds_write_b128 in slots 8..15
//       of the next iteration
template <int MODE, bool READS, bool DMA, bool BARS, int STAG = 0>
__global__ __launch_bounds__(512, 2) void mix(const char* __restrict__ src, float* __restrict__ out, int iters, size_t span) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lbase = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t rd = lbase + (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4) + wave * 2048;
    const uint32_t wave_off = lbase + (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u;
    // fill this block's LDS from src (with random src data the MFMA operands toggle as in a real GEMM: power, hence clocks)
    for (uint32_t o = tid * 16; o < 131072; o += blockDim.x * 16)
        *reinterpret_cast<int4_t*>(smem + o) = *reinterpret_cast<const int4_t*>(src + ((size_t)blockIdx.x * 131072 + o) % span);
    __syncthreads();
    const char* g = src + ((size_t)blockIdx.x * 65536 + (size_t)tid * 16) % span;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc4[32];
    f32x16 acc16[8];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc4[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
    bf16x8 fa[8], fb[4];
    int4_t stg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) stg[i] = int4_t{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = lds_b128<0>(rd + i * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i) fb[i] = lds_b128<64>(rd + i * 2048);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        const uint32_t slot = (it & 1) * 65536;
        constexpr int NM = MODE == 0 ? 64 : 32;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if constexpr (MODE == 0) acc4[i & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i & 3], fa[i & 7], acc4[i & 31], 0, 0, 0);
            else acc16[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i & 3], fa[i & 7], acc16[i & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int STEP = NM / 32;                       // fillers spread evenly: 24 reads + 8 pieces over the MFMAs
            if (i % STEP == 0) {
                const int f = i / STEP;                         // 0..31
                if (READS && f < 24) {
                    if (f < 16) fa[f & 7] = lds_b128<0>(rd + slot + (f & 7) * 2048 + (f >> 3) * 64);
                    else fb[f & 3] = lds_b128<32>(rd + slot + (f & 3) * 2048 + ((f >> 2) & 1) * 64);
                }
                if constexpr (DMA) {
                    auto piece = [&](int pi) {
                        __builtin_amdgcn_global_load_lds(GLB_PTR(g + (size_t)pi * 8192), LDS_PTR((void*)(size_t)(wave_off + (65536 - slot) + pi * 8192)), 16, 0, 0);
                    };
                    if (STAG == 0 && f >= 24) piece(f - 24);
                    if (STAG == 1 && ((wave < 4 && f >= 8 && f < 16) || (wave >= 4 && f >= 24))) piece(f & 7);
                    if (STAG == 2 && f == 24) { for (int pi = 0; pi < 8; ++pi) piece(pi); }
                    if constexpr (STAG == 3) {
                        if (f >= 24) stg[f - 24] = *reinterpret_cast<const int4_t*>(g + (size_t)(f - 24) * 8192);
                        if (f == 8) asm volatile("s_waitcnt vmcnt(0)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]), "+v"(stg[4]), "+v"(stg[5]), "+v"(stg[6]), "+v"(stg[7]));
                        if (f >= 8 && f < 16) asm volatile("ds_write_b128 %0, %1" :: "v"(wave_off + lane * 16 + slot + (f - 8) * 8192), "v"(stg[f - 8]) : "memory");
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (BARS && (i == NM / 2 - 1 || i == NM * 3 / 4 - 1)) {
                if (i == NM / 2 - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else if (STAG != 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g += 65536;
        if (g >= src + span - 131072) g -= span - 262144;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc4[i][0];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc16[i][0];
    if (s == 12345.f) out[tid] = s;
    if (blockIdx.x == 0 && tid == 0) {
        reinterpret_cast<unsigned long long*>(out)[256] = __builtin_amdgcn_s_memtime() - t0;
        reinterpret_cast<unsigned long long*>(out)[257] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}


// The alternative block shape for the same 256x256x64 K-tile: 4 waves (one per SIMD), wave tile 128x128 = 16 accumulators of
// v_mfma_f32_32x32x16_bf16 (256 accumulator registers), per wave and iteration 64 MFMAs + 32 ds_read_b128 + 16 DMA pieces.
template <bool READS, bool DMA, int BARS, int SHAPE = 0>
__global__ __launch_bounds__(256) void mix4(const char* __restrict__ src, float* __restrict__ out, int iters, size_t span) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lbase = (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint32_t rd = lbase + (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4) + wave * 2048;
    const uint32_t wave_off = lbase + (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u;
    // fill this block's LDS from src (with random src data the MFMA operands toggle as in a real GEMM: power, hence clocks)
    for (uint32_t o = tid * 16; o < 131072; o += blockDim.x * 16)
        *reinterpret_cast<int4_t*>(smem + o) = *reinterpret_cast<const int4_t*>(src + ((size_t)blockIdx.x * 131072 + o) % span);
    __syncthreads();
    const char* g = src + ((size_t)blockIdx.x * 65536 + (size_t)tid * 16) % span;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[16];
    f32x4 acc4[64];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[8], fb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = lds_b128<0>(rd + i * 2048); fb[i] = lds_b128<64>(rd + i * 2048); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < iters; ++it) {
        const uint32_t slot = (it & 1) * 65536;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if constexpr (SHAPE == 0) {
                acc[i & 15] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i & 7], fa[(i >> 1) & 7], acc[i & 15], 0, 0, 0);
            } else {     // the same K-tile as 128 x 16x16x32: two per filler slot
                acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i & 7], fa[(i >> 3) & 7], acc4[i], 0, 0, 0);
                acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[(i + 1) & 7], fa[(i >> 3) & 7], acc4[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (READS && i < 32) {
                if (i < 16) fa[i & 7] = lds_b128<0>(rd + slot + (i & 7) * 2048 + (i >> 3) * 64);
                else fb[i & 7] = lds_b128<32>(rd + slot + (i & 7) * 2048 + ((i >> 3) & 1) * 64);
            }
            if (DMA && i >= 32 && (i & 1) == 0) {
                const int pi = (i - 32) >> 1;                       // 16 pieces of 256 threads x 16 B = 4 KiB each
                __builtin_amdgcn_global_load_lds(GLB_PTR(g + (size_t)pi * 4096), LDS_PTR((void*)(size_t)(wave_off + (65536 - slot) + pi * 4096)), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((BARS >= 1 && i == 31) || (BARS == 2 && i == 47)) {
                if (i == 31 && BARS == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else if (i == 31) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // one barrier: slot free + next tile landed
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g += 65536;
        if (g >= src + span - 131072) g -= span - 262144;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0];
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc4[i][0];
    if (s == 12345.f) out[tid] = s;
    if (tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(out) + 512 + blockIdx.x * 4;
        o[0] = r0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = __builtin_amdgcn_s_memtime() - t0;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[3] = xcc;
    }
}

template <bool R, bool D, int B, int SHAPE = 0>
static void run4(const char* name, const char* src, float* out, size_t span) {
    auto k = mix4<R, D, B, SHAPE>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 131072, 0, src, out, 50, span);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 131072, 0, src, out, iters, span);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us_per_tile = ms * 1e3 / iters;
    printf("%-44s %7.3f us/K-tile  %6.0f TF/s-equivalent\n", name, us_per_tile, 2.0 * 256 * 256 * 64 * 256 / (us_per_tile * 1e-6) / 1e12);
    std::vector<unsigned long long> h(256 * 4);
    hipMemcpy(h.data(), reinterpret_cast<unsigned long long*>(out) + 512, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long rmin = ~0ull;
    for (int b = 0; b < 256; ++b) rmin = h[b * 4] < rmin ? h[b * 4] : rmin;
    printf("   per block (start us, duration us, GHz, xcc):");
    for (int b = 0; b < 256; b += 1) {
        if (b % 8 == 0) printf("\n     ");
        printf(" [%5.0f %5.0f %.2f %llu]", (h[b * 4] - rmin) / 100.0, (h[b * 4 + 1] - h[b * 4]) / 100.0, h[b * 4 + 2] / ((h[b * 4 + 1] - h[b * 4]) * 10.0), h[b * 4 + 3] & 15);
    }
    printf("\n");
}

template <int MODE, bool R, bool D, bool B, int STAG = 0>
static void run(const char* name, const char* src, float* out, size_t span) {
    auto k = mix<MODE, R, D, B, STAG>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, 0, src, out, 50, span);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 131072, 0, src, out, iters, span);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us_per_tile = ms * 1e3 / iters;
    const double tf = 2.0 * 256 * 256 * 64 * 256 / (us_per_tile * 1e-6) / 1e12;
    unsigned long long cyc[2] = {0, 0};
    hipMemcpy(cyc, reinterpret_cast<unsigned long long*>(out) + 256, 16, hipMemcpyDeviceToHost);
    printf("%-44s %7.3f us/K-tile  %5.0f memtime ticks/K-tile (%.2f G/s; memrealtime %.1f M/s)   %6.0f TF/s-equivalent\n", name, us_per_tile,
           (double)cyc[0] / iters, (double)cyc[0] / (ms * 1e6), (double)cyc[1] / (ms * 1e3), tf);
}

int main(int argc, char** argv) {
    const size_t span = 512ull << 20;
    char* src; float* out;
    hipMalloc(&src, span); hipMalloc(&out, 65536);
    hipMemset(src, 0, span);
    if (argc > 1 && atoi(argv[1]) == 1) {           // random bf16 in [-1, 1): realistic operand toggling
        std::vector<unsigned short> h(32u << 20);
        unsigned long long x = 88172645463325252ull;
        for (auto& v : h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const float f = (float)((x >> 11) & 0xFFFFFF) / 8388608.f - 1.f;
            unsigned int u; memcpy(&u, &f, 4);
            v = (unsigned short)(u >> 16);
        }
        for (size_t o = 0; o < span; o += h.size() * 2) hipMemcpy(src + o, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        printf("operands: random bf16\n");
    } else printf("operands: zeros\n");
    run<0, false, false, false>("16x16x32 x64, MFMA only", src, out, span);
    run<1, false, false, false>("32x32x16 x32, MFMA only", src, out, span);
    run<0, true, false, false>("16x16x32 + 24 reads", src, out, span);
    run<1, true, false, false>("32x32x16 + 24 reads", src, out, span);
    run<0, true, true, false>("16x16x32 + reads + 8 DMA", src, out, span);
    run<1, true, true, false>("32x32x16 + reads + 8 DMA", src, out, span);
    run<0, true, true, true>("16x16x32 + reads + DMA + 2 barriers", src, out, span);
    run<1, true, true, true>("32x32x16 + reads + DMA + 2 barriers", src, out, span);
    run<0, true, true, true, 1>("16x16x32 full mix, DMA staggered by wave group", src, out, span);
    run<1, true, true, true, 1>("32x32x16 full mix, DMA staggered by wave group", src, out, span);
    run<0, true, true, true, 2>("16x16x32 full mix, 8 pieces back to back", src, out, span);
    run<0, true, true, true, 3>("16x16x32 full mix, register-staged loads", src, out, span);
    run<1, true, true, true, 3>("32x32x16 full mix, register-staged loads", src, out, span);
    run<0, true, true, false, 3>("16x16x32 reads + register-staged, no barriers", src, out, span);
    run<0, false, true, true>("16x16x32 + DMA + barriers (no reads)", src, out, span);
    run<1, false, true, true>("32x32x16 + DMA + barriers (no reads)", src, out, span);
    run4<false, false, 0>("4 waves: 32x32x16 x64, MFMA only", src, out, span);
    run4<true, false, 0>("4 waves: + 32 reads", src, out, span);
    run4<true, true, 0>("4 waves: + reads + 16 DMA", src, out, span);
    run4<true, true, 2>("4 waves: + reads + DMA + 2 barriers", src, out, span);
    run4<true, true, 1>("4 waves: + reads + DMA + 1 barrier (vmcnt 0)", src, out, span);
    run4<false, false, 0, 1>("4 waves 16x16x32 x128, MFMA only", src, out, span);
    run4<true, false, 0, 1>("4 waves 16x16x32: + 32 reads", src, out, span);
    run4<true, true, 0, 1>("4 waves 16x16x32: + reads + 16 DMA", src, out, span);
    run4<true, true, 1, 1>("4 waves 16x16x32: + reads + DMA + 1 barrier", src, out, span);
    return 0;
}
