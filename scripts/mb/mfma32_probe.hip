// Probe of v_mfma_f32_32x32x16_bf16's register layout (diagnostic): which (row, col) of D = A x B does register i of lane l hold,
// with the A operand lane holding row (l & 31), k = 8 * (l >> 5) + 0..7 and the B operand lane holding column (l & 31), same k.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)0.f; b[j] = (__bf16)0.f; }
    if ((l >> 5) == 0) { a[0] = (__bf16)(float)(((l & 31) + 1) * 64); b[0] = (__bf16)(float)((l & 31) + 1); }   // k = 0 only
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}
int main() {
    float* d; float h[1024];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // value = (row + 1) * 64 * (col + 1), row from A, col from B
    bool ok = true;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) {
            const int v = (int)h[l * 16 + i];
            int row = -1, col = -1;
            for (int r = 1; r <= 32 && row < 0; ++r)
                if (v % (r * 64) == 0 && v / (r * 64) >= 1 && v / (r * 64) <= 32) {
                    // ambiguous factorisations exist; test the conjectured layout instead
                }
            const int er = 8 * (i >> 2) + 4 * (l >> 5) + (i & 3), ec = l & 31;      // conjecture: row = 8*(i/4) + 4*(l/32) + i%4, col = l%32
            if (v != (er + 1) * 64 * (ec + 1)) ok = false;
            (void)row; (void)col;
        }
    printf("layout row = 8*(i>>2) + 4*(l>>5) + (i&3), col = l&31 : %s\n", ok ? "CONFIRMED" : "NOT confirmed");
    if (!ok) for (int l = 0; l < 64; l += 21) { printf("lane %d:", l); for (int i = 0; i < 16; ++i) printf(" %d", (int)h[l * 16 + i]); printf("\n"); }
    return 0;
}
