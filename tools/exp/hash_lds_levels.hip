// experiment: do the two dense coarse levels of the stock hash grid (16^3 = 4 096 and 23^3 = 12 167 entries of 8 B: 32 KB + 95 KB)
// gather faster from an LDS-staged copy than through the L1/L2 path k_hash_fwd uses?  131 072 ray-ordered points (1 024 rays x 128
// samples), smoothstep trilinear blend as csrc/hash_encode.hip, level-major output.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
struct Lvl { float scale; unsigned res, offset; };
__device__ __forceinline__ float sstep(float t) { return t * t * (3.f - 2.f * t); }
template <bool LDS>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, const float2* __restrict__ emb, float2* __restrict__ out, unsigned B, Lvl l0, Lvl l1,
                                         unsigned blocks_per_level) {
    extern __shared__ float2 tab[];
    const unsigned level = blockIdx.x / blocks_per_level, chunk = blockIdx.x % blocks_per_level;
    const Lvl L = level ? l1 : l0;
    const unsigned n = L.res * L.res * L.res;
    const float2* g = emb + L.offset;
    if (LDS) {
        for (unsigned i = threadIdx.x; i < n; i += 256) tab[i] = g[i];
        __syncthreads();
        g = tab;
    }
    // persistent over the level's points when staging (amortise the fill), one block per 256 points otherwise
    const unsigned stride = LDS ? blocks_per_level * 256 : B;
    for (unsigned b = chunk * 256 + threadIdx.x; b < B; b += stride) {
        float w[3]; unsigned c[3];
#pragma unroll
        for (int d = 0; d < 3; d++) { float p = x[b * 3 + d] * L.scale; float f = floorf(p); c[d] = (unsigned)f; w[d] = sstep(p - f); }
        float2 e[8];
#pragma unroll
        for (int k8 = 0; k8 < 8; k8++) {
            const unsigned i = (c[0] + (k8 & 1)) + (c[1] + ((k8 >> 1) & 1)) * L.res + (c[2] + (k8 >> 2)) * L.res * L.res;
            e[k8] = g[i < n ? i : n - 1];
        }
        float2 a = make_float2(0.f, 0.f);
#pragma unroll
        for (int k8 = 0; k8 < 8; k8++) {
            const float wt = ((k8 & 1) ? w[0] : 1 - w[0]) * (((k8 >> 1) & 1) ? w[1] : 1 - w[1]) * ((k8 >> 2) ? w[2] : 1 - w[2]);
            a.x += wt * e[k8].x; a.y += wt * e[k8].y;
        }
        out[(size_t)level * B + b] = a;
        if (!LDS) break;
    }
}
int main() {
    const unsigned R = 1024, S = 128, B = R * S;
    std::vector<float> hx(B * 3);
    unsigned s = 7; auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.f; };
    for (unsigned r = 0; r < R; r++) {      // rays through the unit cube, samples ordered along the ray (as the sampler hands them over)
        float o[3] = {rnd(), rnd(), 0.02f}, d[3] = {rnd() - 0.5f, rnd() - 0.5f, 1.f};
        for (unsigned i = 0; i < S; i++) for (int k3 = 0; k3 < 3; k3++) { float v = o[k3] + d[k3] * (i / (float)S) * 0.9f; hx[(r * S + i) * 3 + k3] = fminf(fmaxf(v, 0.f), 0.999f); }
    }
    Lvl l0{15.f, 16, 0}, l1{22.f, 23, 4096};
    const unsigned entries = 4096 + 12167;
    float* x; float2 *emb, *out; hipMalloc(&x, B * 12); hipMalloc(&emb, entries * 8); hipMalloc(&out, 2 * B * 8);
    hipMemcpy(x, hx.data(), B * 12, hipMemcpyHostToDevice); hipMemset(emb, 0, entries * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 20; i++) launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("%-44s %7.1f us per launch (2 levels x %u points)\n", name, ms / 20 * 1e3, B);
    };
    run("global gathers (L1/L2 path), 1 block/256 pts", [&] { k<false><<<2 * (B / 256), 256>>>(x, emb, out, B, l0, l1, B / 256); });
    hipFuncSetAttribute((const void*)k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 12167 * 8);
    for (unsigned bpl : {64u, 128u, 256u})
        run(bpl == 64 ? "LDS-staged, 64 persistent blocks per level" : (bpl == 128 ? "LDS-staged, 128 persistent blocks per level" : "LDS-staged, 256 persistent blocks per level"),
            [&] { k<true><<<2 * bpl, 256, 12167 * 8>>>(x, emb, out, B, l0, l1, bpl); });
    return 0;
}
