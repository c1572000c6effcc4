// experiment: random float atomic rate vs how the address space is partitioned over the 8 XCDs (each XCD has its own L2).
//   shared    : every block scatters over the whole region
//   per-XCD   : block b scatters only inside slice (xcc_id) of the region  (slice size swept)
// Also checks that blockIdx % 8 == XCC_ID for a plain launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20); }
__global__ void k_map(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }
// region = 8 slices of `slice_mask+1` float2 cells; mode 0: shared (slice chosen by the random index), 1: by XCC_ID, 2: by blockIdx % 8
__global__ void k_scatter(float* t, const unsigned* idx, int n, unsigned slice_mask, int mode, int per_thread) {
    const unsigned x = mode == 1 ? xcc_id() : (blockIdx.x & 7);
    size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x);
    for (int r = 0; r < per_thread; r++) {
        size_t i = i0 + (size_t)r * gridDim.x * 256;
        if (i >= (size_t)n) return;
        unsigned j = idx[i];
        unsigned slice = mode == 0 ? (j >> 28) & 7 : x;
        size_t cell = (size_t)slice * (slice_mask + 1) + (j & slice_mask);
        unsafeAtomicAdd(t + 2 * cell, 1.0f);
        unsafeAtomicAdd(t + 2 * cell + 1, 0.5f);
    }
}
int main() {
    const int n = 1 << 24;
    std::vector<unsigned> h(n); unsigned s = 12345; for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = s ^ (s >> 15); }
    unsigned* idx; float* t; unsigned* map;
    const size_t max_cells = 8ull << 21;   // 8 slices x 2M cells x 8 B = 128 MB
    hipMalloc(&idx, n * 4); hipMalloc(&t, max_cells * 8); hipMalloc(&map, 4096 * 4);
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    k_map<<<4096, 64>>>(map); std::vector<unsigned> m(4096); hipMemcpy(m.data(), map, 4096 * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int b = 0; b < 4096; b++) bad += (m[b] != (unsigned)(b & 7));
    printf("blockIdx%%8 != XCC_ID for %d of 4096 blocks; first 16 ids:", bad); for (int b = 0; b < 16; b++) printf(" %u", m[b]); printf("\n");
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int lg = 16; lg <= 21; lg++) {           // cells per slice: 64K (512 KB) .. 2M (16 MB)
        const unsigned mask = (1u << lg) - 1;
        for (int mode = 0; mode < 3; mode++) {
            for (int grid : {n / 256, 2048}) {
                const int per = (n + grid * 256 - 1) / (grid * 256);
                hipMemset(t, 0, max_cells * 8);
                k_scatter<<<grid, 256>>>(t, idx, n, mask, mode, per); hipDeviceSynchronize();
                hipEventRecord(a); for (int r = 0; r < 3; r++) k_scatter<<<grid, 256>>>(t, idx, n, mask, mode, per); hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                // correctness: total of channel 0 must be 4 * n
                std::vector<float> ht((size_t)8 * (mask + 1) * 2); hipMemcpy(ht.data(), t, ht.size() * 4, hipMemcpyDeviceToHost);
                double tot = 0; for (size_t c = 0; c < ht.size(); c += 2) tot += ht[c];
                printf("slice %6.1f MB  mode %s grid %6d: %7.1f us  %6.1f G atomics/s  sum %s\n", (mask + 1) * 8 / 1048576.0,
                       mode == 0 ? "shared " : mode == 1 ? "xcc_id " : "blk%8  ", grid, ms / 3 * 1e3, 2.0 * n / (ms / 3 * 1e-3) / 1e9,
                       tot == 4.0 * n ? "ok" : "WRONG");
            }
        }
    }
    return 0;
}
