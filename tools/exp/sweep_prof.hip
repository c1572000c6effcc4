// experiment: where a wave tile of the fused sweep (k_sdf_mlp2<false, true>: hash gather inside the trunk kernel) spends its time -- s_memtime stamps
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I holoscene_amd/csrc tools/exp/sweep_prof.hip -o /tmp/sweep_prof && /tmp/sweep_prof [rays]
#define HS_SDF2_PROFILE 1
#define HS_SWEEP_PROFILE 1
#include "../../holoscene_amd/csrc/sdf_mlp2.hip"
#include <cmath>
#include <cstdio>
#include <vector>
#include <algorithm>
int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 1024, S = 128;
    const int64_t B = (int64_t)R * S;
    // stock grid: 16 levels, base 16 -> 2048, 2^19 entries per level at most
    const double pls = std::exp2(std::log2(2048.0 / 16.0) / 15.0);
    std::vector<int32_t> offs(17, 0);
    for (int l = 0; l < 16; l++) {
        const long res = (long)std::ceil(16.0 * std::pow(pls, l));
        offs[l + 1] = offs[l] + (int32_t)std::min<long>(1L << 19, res * res * res);
    }
    const int64_t T = offs[16];
    unsigned s = 1;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f; };
    std::vector<float> hx(B * 3), hx01(B * 3), ht(T * 2), h(256 * 256);
    for (int r = 0; r < R; r++) {       // rays from (0.7, 0, 0) towards the origin, jittered; 128 sorted depths in [0, 2.2]
        float d[3] = {-1.f + (rnd() - 0.5f), rnd() - 0.5f, rnd() - 0.5f};
        const float n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        std::vector<float> z(S);
        for (auto &v : z) v = rnd() * 2.2f;
        std::sort(z.begin(), z.end());
        for (int i = 0; i < S; i++)
            for (int c = 0; c < 3; c++) {
                const float v = (c == 0 ? 0.7f : 0.f) + z[i] * d[c] / n;
                hx[((int64_t)r * S + i) * 3 + c] = v;
                hx01[((int64_t)r * S + i) * 3 + c] = (v + 1.f) * 0.5f;
            }
    }
    for (auto &v : ht) v = (rnd() - 0.5f) * 2e-4f;
    float *x, *x01, *tab, *W0, *W1, *W2, *b, *bias, *out, *feat;
    int32_t *doffs;
    void *W0f, *W12f;
    hipMalloc(&x, B * 12); hipMalloc(&x01, B * 12); hipMalloc(&tab, T * 8); hipMalloc(&out, B * 4); hipMalloc(&doffs, 17 * 4); hipMalloc(&feat, B * 64);
    hipMalloc(&W0, 256 * 71 * 4); hipMalloc(&W1, 256 * 256 * 4); hipMalloc(&W2, 32 * 256 * 4); hipMalloc(&b, 1024 * 4);
    hipMalloc(&W0f, hs_sdf_mlp2_pack_bytes(0)); hipMalloc(&W12f, hs_sdf_mlp2_pack_bytes(1) + hs_sdf_mlp2_pack_bytes(2)); hipMalloc(&bias, hs_sdf_mlp2_pack_bytes(3));
    hipMemcpy(x, hx.data(), B * 12, hipMemcpyHostToDevice); hipMemcpy(x01, hx01.data(), B * 12, hipMemcpyHostToDevice);
    hipMemcpy(tab, ht.data(), T * 8, hipMemcpyHostToDevice); hipMemcpy(doffs, offs.data(), 17 * 4, hipMemcpyHostToDevice);
    hipMemset(feat, 0, B * 64);
    for (auto &v : h) v = (rnd() - 0.5f) * 0.1f;
    hipMemcpy(W0, h.data(), 256 * 71 * 4, hipMemcpyHostToDevice); hipMemcpy(W1, h.data(), 256 * 256 * 4, hipMemcpyHostToDevice);
    hipMemcpy(W2, h.data(), 32 * 256 * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), 1024 * 4, hipMemcpyHostToDevice);
    void *W2f = (char *)W12f + hs_sdf_mlp2_pack_bytes(1);
    printf("pack rc %d\n", hs_sdf_mlp2_pack(W0, 71, b, W1, b + 256, W2, b + 512, 32, W0f, W12f, W2f, bias, 1, nullptr));
    const float Sl = (float)std::log2(pls);
    auto sweep = [&] { return hs_sdf_sweep_fwd(x, x01, tab, doffs, Sl, 16, W0f, W12f, W2f, bias, nullptr, nullptr, 32, -1, 0, out, nullptr, B, nullptr, nullptr); };
    auto plain = [&] { return hs_sdf_mlp2_fwd(x, feat, W0f, W12f, W2f, bias, 32, -1, 0, out, nullptr, B, nullptr, 2, nullptr); };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    printf("rc %d %d\n", sweep(), plain());
    for (int which = 0; which < 2; which++) {
        for (int r = 0; r < 3; r++) which ? sweep() : plain();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; r++) which ? sweep() : plain();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s B=%lld: %.1f us per launch\n", which ? "fused sweep" : "trunk on ready words", (long long)B, ms * 100);
    }
    std::vector<unsigned long long> p(256 * 8 * 16), g(256 * 8 * 8);
    hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(g_sdf2_prof), p.size() * 8);
    hipMemcpyFromSymbol(g.data(), HIP_SYMBOL(g_sweep_prof), g.size() * 8);
    for (int t = 0; t < 2; t++) {       // first / second tile of every wave
        double a[7] = {0, 0, 0, 0, 0, 0, 0}; int n = 0;
        double start = 0;
        for (int w = 0; w < 256 * 8; w++) {
            const unsigned long long *q = &p[w * 16 + (t ? 0 : 8)], *gg = &g[w * 8 + (t ? 4 : 0)];
            if (q[4] <= q[0] || gg[2] <= q[0]) continue;        // (a wave whose lane 0 lies outside the cube writes no gather stamps)
            a[0] += (double)(gg[0] - q[0]); a[1] += (double)(gg[1] - gg[0]); a[2] += (double)(gg[2] - gg[1]); a[3] += (double)(q[1] - gg[2]);
            a[4] += (double)(q[2] - q[1]); a[5] += (double)(q[3] - q[2]); a[6] += (double)(q[4] - q[3]);
            start += (double)(q[0] - p[w * 16 + 7]);
            n++;
        }
        printf("%s tile, mean over %d waves (shader ticks): start at +%.0f | locate+index+issue %.0f  posenc %.0f  wait+blend %.0f  pack %.0f | layer0 %.0f  layer1 %.0f  layer2+out %.0f\n",
               t ? "second" : "first", n, start / n, a[0] / n, a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n);
    }
    double life = 0, real = 0;
    for (int w = 0; w < 256 * 8; w++) { life += p[w * 16 + 5]; real += p[w * 16 + 6]; }
    printf("wave lifetime: %.0f shader ticks, %.2f us => clock %.2f GHz\n", life / 2048, real / 2048 / 100.0, life / real * 0.1);
    return 0;
}
