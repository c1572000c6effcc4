// calibration kernels for the FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your
// own access pattern"): a wide coalesced stream, and random 8-byte gathers over a 48.8 MB table (the hash-grid access pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void calib_stream16(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = src[i]; v.x += 1.f; dst[i] = v; }
}
__global__ void calib_gather8(const float2* __restrict__ table, const unsigned* __restrict__ idx, float2* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table[idx[i]];
}
int main() {
    const size_t n_stream = (size_t)256 << 20 >> 4;          // 256 MB read + 256 MB written
    const size_t entries = 6098108, n_gather = (size_t)16 << 20;  // 48.8 MB table, 16.8 M gathers (134 MB of payload)
    float4 *a, *b; float2 *t, *o; unsigned* idx;
    hipMalloc(&a, n_stream * 16); hipMalloc(&b, n_stream * 16); hipMalloc(&t, entries * 8); hipMalloc(&o, n_gather * 8); hipMalloc(&idx, n_gather * 4);
    hipMemset(a, 0, n_stream * 16); hipMemset(t, 0, entries * 8);
    std::vector<unsigned> h(n_gather); unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s ^ (s >> 15)) % entries; }
    hipMemcpy(idx, h.data(), n_gather * 4, hipMemcpyHostToDevice);
    for (int r = 0; r < 3; r++) {
        calib_stream16<<<256 * 8, 256>>>(a, b, n_stream);
        calib_gather8<<<(n_gather + 255) / 256, 256>>>(t, idx, o, n_gather);
    }
    hipDeviceSynchronize();
    printf("calib_stream16: reads %zu B, writes %zu B per launch; calib_gather8: %zu gathers of 8 B (payload %zu B, idx %zu B read, %zu B written) over a %zu B table\n",
           n_stream * 16, n_stream * 16, n_gather, n_gather * 8, n_gather * 4, n_gather * 8, entries * 8);
    return 0;
}
