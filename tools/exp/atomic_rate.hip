// experiment: random float atomics, AoS (2 adjacent channels) vs SoA (channels in separate planes) vs single channel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_aos(float* t, const unsigned* idx, const float* v, int n, unsigned mask) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    unsigned j = idx[i] & mask; float a = v[i];
    unsafeAtomicAdd(t + 2 * j, a); unsafeAtomicAdd(t + 2 * j + 1, a * 0.5f);
}
__global__ void k_soa(float* t, const unsigned* idx, const float* v, int n, unsigned mask) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    unsigned j = idx[i] & mask; float a = v[i];
    unsafeAtomicAdd(t + j, a); unsafeAtomicAdd(t + (mask + 1) + j, a * 0.5f);
}
__global__ void k_one(float* t, const unsigned* idx, const float* v, int n, unsigned mask) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    unsafeAtomicAdd(t + (idx[i] & mask), v[i]);
}
__global__ void k_f64(double* t, const unsigned* idx, const float* v, int n, unsigned mask) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    unsafeAtomicAdd(t + (idx[i] & mask), (double)v[i]);
}
int main() {
    const int n = 1 << 24; const unsigned mask = (1u << 19) - 1;
    std::vector<unsigned> h(n); unsigned s = 12345; for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = s >> 8; }
    unsigned* idx; float* v; float* t; hipMalloc(&idx, n * 4); hipMalloc(&v, n * 4); hipMalloc(&t, (mask + 1) * 16);
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(v, 0, n * 4); hipMemset(t, 0, (mask + 1) * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch, double atomics) {
        launch(); hipDeviceSynchronize(); hipEventRecord(a); for (int r = 0; r < 5; r++) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("%s: %.1f us  %.1f G atomics/s\n", name, ms / 5 * 1e3, atomics / (ms / 5 * 1e-3) / 1e9);
    };
    run("AoS 2ch", [&] { k_aos<<<n / 256, 256>>>(t, idx, v, n, mask); }, 2.0 * n);
    run("SoA 2ch", [&] { k_soa<<<n / 256, 256>>>(t, idx, v, n, mask); }, 2.0 * n);
    run("one ch ", [&] { k_one<<<n / 256, 256>>>(t, idx, v, n, mask); }, 1.0 * n);
    run("f64 one", [&] { k_f64<<<n / 256, 256>>>((double*)t, idx, v, n, mask); }, 1.0 * n);
    return 0;
}
