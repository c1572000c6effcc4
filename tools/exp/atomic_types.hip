// experiment: random-address atomic rate by operand type (is the ~21 G/s float rate a float-ALU limit or a path limit?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <class T> __global__ void k_add(T* t, const unsigned* idx, int n, unsigned mask) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    atomicAdd(t + (idx[i] & mask), (T)1);
}
__global__ void k_f32(float* t, const unsigned* idx, int n, unsigned mask) {
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    unsafeAtomicAdd(t + (idx[i] & mask), 1.0f);
}
__global__ void k_store(float* t, const unsigned* idx, int n, unsigned mask) {   // plain scattered 4-byte stores for comparison
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    t[idx[i] & mask] = 1.0f;
}
__global__ void k_load(float* t, const unsigned* idx, int n, unsigned mask, float* out) {   // scattered 4-byte loads
    int i = blockIdx.x * 256 + threadIdx.x; if (i >= n) return;
    float v = t[idx[i] & mask]; if (v == 123.f) out[0] = v;
}
__global__ void k_lds(float* t, const unsigned* idx, int n) {   // LDS float atomics, 32 KB table per block
    __shared__ float tab[8192];
    for (int j = threadIdx.x; j < 8192; j += 256) tab[j] = 0.f;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) atomicAdd(&tab[idx[i] & 8191], 1.0f);
    __syncthreads();
    if (threadIdx.x == 0) t[blockIdx.x] = tab[5];
}
int main() {
    const int n = 1 << 24; const unsigned mask = (1u << 20) - 1;
    std::vector<unsigned> h(n); unsigned s = 12345; for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = s ^ (s >> 15); }
    unsigned* idx; void* t; float* out; hipMalloc(&idx, n * 4); hipMalloc(&t, (mask + 1) * 8); hipMalloc(&out, 4);
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(t, 0, (mask + 1) * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize(); hipEventRecord(a); for (int r = 0; r < 5; r++) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("%-12s %8.1f us  %6.1f G ops/s\n", name, ms / 5 * 1e3, n / (ms / 5 * 1e-3) / 1e9);
    };
    run("f32 hw", [&] { k_f32<<<n / 256, 256>>>((float*)t, idx, n, mask); });
    run("i32", [&] { k_add<int><<<n / 256, 256>>>((int*)t, idx, n, mask); });
    run("u64", [&] { k_add<unsigned long long><<<n / 256, 256>>>((unsigned long long*)t, idx, n, mask); });
    run("f64", [&] { k_add<double><<<n / 256, 256>>>((double*)t, idx, n, mask); });
    run("store f32", [&] { k_store<<<n / 256, 256>>>((float*)t, idx, n, mask); });
    run("load f32", [&] { k_load<<<n / 256, 256>>>((float*)t, idx, n, mask, out); });
    run("lds f32", [&] { k_lds<<<2048, 256>>>((float*)t, idx, n); });
    return 0;
}
