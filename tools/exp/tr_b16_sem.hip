// experiment: what ds_read_b64_tr_b16 returns.  LDS holds its own element index (uint16 i at element i); lane l passes the byte address
// 8 l (its own 4-element slot); print, for every lane, the four 16-bit values it gets back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint16_t *out, int stride_elems) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)lds + threadIdx.x * stride_elems * 2;
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = r.x & 0xffff; out[threadIdx.x * 4 + 1] = r.x >> 16;
    out[threadIdx.x * 4 + 2] = r.y & 0xffff; out[threadIdx.x * 4 + 3] = r.y >> 16;
}
int main() {
    uint16_t *d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {4, 16}) {
        k<<<1, 64>>>(d, stride);
        uint16_t h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("lane address = %d * lane elements:\n", stride);
        for (int l = 0; l < 64; l++) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
