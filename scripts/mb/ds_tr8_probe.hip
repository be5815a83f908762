// ds_tr8_probe.hip - what ds_read_b64_tr_b8 returns (gfx950): every lane supplies the address 64 * lane; LDS byte a holds a (two passes:
// low byte, high byte), so each returned byte names the lane whose address it was read through and the offset behind that address.
//   hipcc --offload-arch=gfx950 -O2 scripts/mb/ds_tr8_probe.hip -o /tmp/tr8probe && /tmp/tr8probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) int int2_t;

__global__ void probe(uint8_t* out, int hi) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = hi ? (uint8_t)(i >> 8) : (uint8_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)lds + 64u * threadIdx.x;
    int2_t v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    reinterpret_cast<int2_t*>(out)[threadIdx.x] = v;
}

int main() {
    uint8_t *d, lo[512], hi[512];
    hipMalloc(&d, 512);
    probe<<<1, 64>>>(d, 0); hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
    probe<<<1, 64>>>(d, 1); hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 8; ++e) {
            const int a = lo[l * 8 + e] | (hi[l * 8 + e] << 8);
            printf("  (L%2d,+%d)", a / 64, a % 64);
        }
        printf("\n");
    }
    return 0;
}
