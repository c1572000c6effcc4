// what does __builtin_amdgcn_permlane32_swap(a, b) return?  (a = lane id, b = 100 + lane id)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("r[0]: lane0 %u lane1 %u lane31 %u lane32 %u lane33 %u lane63 %u\n", h[0], h[1], h[31], h[32], h[33], h[63]);
    printf("r[1]: lane0 %u lane1 %u lane31 %u lane32 %u lane33 %u lane63 %u\n", h[64], h[65], h[95], h[96], h[97], h[127]);
    return 0;
}
