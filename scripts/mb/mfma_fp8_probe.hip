// mfma_fp8_probe.hip - what v_mfma_scale_f32_16x16x128_f8f6f4 computes from a given lane -> byte assignment (gfx950).
//   hipcc --offload-arch=gfx950 -O2 scripts/mb/mfma_fp8_probe.hip -o /tmp/fp8probe && /tmp/fp8probe
// One wave, A[16][128] and B[16][128] in OCP fp8 (e4m3, or e5m2 for A), scales = 1.  Fragment of lane l (row r = l & 15, group
// g = l >> 4), two assignments of its 32 bytes:
//   natural : row bytes [32 g, 32 g + 32)
//   chunked : row bytes [16 g, 16 g + 16) then [64 + 16 g, 64 + 16 g + 16)      (what the bf16 kernel's two k-step reads hold)
// Both must give D = A B^T if the instruction pairs element e of group g of A with element e of group g of B.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

__device__ __host__ static float e4m3_to_f(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : (e == 15 && m == 7 ? NAN : ldexpf(1.f + m / 8.f, e - 7));
    return s ? -x : x;
}
__device__ __host__ static float e5m2_to_f(uint8_t v) {
    int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
    float x = e == 0 ? ldexpf((float)m, -16) : (e == 31 ? (m ? NAN : INFINITY) : ldexpf(1.f + m / 4.f, e - 15));
    return s ? -x : x;
}

template <int AFMT, int CHUNKED>
__global__ void probe(const uint8_t* A, const uint8_t* B, float* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v8i a, b;
    const int* ar = reinterpret_cast<const int*>(A + r * 128);
    const int* br = reinterpret_cast<const int*>(B + r * 128);
    for (int i = 0; i < 8; ++i) {
        const int w = CHUNKED ? (i < 4 ? 4 * g + i : 16 + 4 * g + (i - 4)) : 8 * g + i;      // dword index inside the row
        a[i] = ar[w];
        b[i] = br[w];
    }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    // swapped roles as in gemm256: first operand = B rows (weights, e4m3), second = A rows: a lane ends with D[m = l & 15][n = 4 g + i]
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, 0 /*fmt of 1st: e4m3*/, AFMT /*fmt of 2nd*/, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int i = 0; i < 4; ++i) D[(l & 15) * 16 + 4 * g + i] = c[i];
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128], hA5[16 * 128];
    srand(3);
    for (int i = 0; i < 16 * 128; ++i) {
        do { hA[i] = rand() & 0xFF; } while (((hA[i] >> 3) & 15) == 15 || ((hA[i] >> 3) & 15) > 9);      // finite, modest magnitude
        do { hB[i] = rand() & 0xFF; } while (((hB[i] >> 3) & 15) == 15 || ((hB[i] >> 3) & 15) > 9);
        do { hA5[i] = rand() & 0xFF; } while (((hA5[i] >> 2) & 31) > 17);
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int afmt = cfg >> 1, chunked = cfg & 1;
        hipMemcpy(dA, afmt ? hA5 : hA, sizeof hA, hipMemcpyHostToDevice);
        if (cfg == 0) probe<0, 0><<<1, 64>>>(dA, dB, dD);
        if (cfg == 1) probe<0, 1><<<1, 64>>>(dA, dB, dD);
        if (cfg == 2) probe<1, 0><<<1, 64>>>(dA, dB, dD);
        if (cfg == 3) probe<1, 1><<<1, 64>>>(dA, dB, dD);
        float hD[256];
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0;
                for (int k = 0; k < 128; ++k) ref += (double)(afmt ? e5m2_to_f(hA5[m * 128 + k]) : e4m3_to_f(hA[m * 128 + k])) * e4m3_to_f(hB[n * 128 + k]);
                maxerr = fmax(maxerr, fabs(ref - hD[m * 16 + n]));
                maxref = fmax(maxref, fabs(ref));
            }
        printf("A fmt %s, %s fragment: max |D - A B^T| = %.3e (max |ref| %.3e)\n", afmt ? "e5m2" : "e4m3", chunked ? "chunked" : "natural", maxerr, maxref);
    }
    return 0;
}
