// experiment: per-phase cycle counts (s_memtime) of one wave tile of k_sdf_mlp2 in steady state (second tile of every wave)
#define HS_SDF2_PROFILE 1
#include "../../holoscene_amd/csrc/sdf_mlp2.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 131072;
    float *x, *feat, *W0, *W1, *W2, *b, *bias, *out;
    void *W0f, *W12f;
    hipMalloc(&x, B * 12); hipMalloc(&feat, B * 128); hipMalloc(&out, B * 4);
    hipMalloc(&W0, 256 * 71 * 4); hipMalloc(&W1, 256 * 256 * 4); hipMalloc(&W2, 32 * 256 * 4); hipMalloc(&b, 1024 * 4);
    hipMalloc(&W0f, hs_sdf_mlp2_pack_bytes(0)); hipMalloc(&W12f, hs_sdf_mlp2_pack_bytes(1) + hs_sdf_mlp2_pack_bytes(2)); hipMalloc(&bias, hs_sdf_mlp2_pack_bytes(3));
    std::vector<float> h(B * 32);
    unsigned s = 1;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto &v : h) v = rnd();
    hipMemcpy(x, h.data(), B * 12, hipMemcpyHostToDevice); hipMemcpy(feat, h.data(), B * 128, hipMemcpyHostToDevice);
    for (auto &v : h) v = rnd() * 0.1f;
    hipMemcpy(W0, h.data(), 256 * 71 * 4, hipMemcpyHostToDevice); hipMemcpy(W1, h.data(), 256 * 256 * 4, hipMemcpyHostToDevice);
    hipMemcpy(W2, h.data(), 32 * 256 * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), 1024 * 4, hipMemcpyHostToDevice);
    void *W2f = (char *)W12f + hs_sdf_mlp2_pack_bytes(1);
    printf("pack rc %d\n", hs_sdf_mlp2_pack(W0, 71, b, W1, b + 256, W2, b + 512, 32, W0f, W12f, W2f, bias, 1, nullptr));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) hs_sdf_mlp2_fwd(x, feat, W0f, W12f, W2f, bias, 32, -1, 0, out, nullptr, B, nullptr, 1, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) hs_sdf_mlp2_fwd(x, feat, W0f, W12f, W2f, bias, 32, -1, 0, out, nullptr, B, nullptr, 1, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B=%lld: %.1f us per launch\n", (long long)B, ms * 100);
    std::vector<unsigned long long> p(256 * 8 * 16);
    hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(g_sdf2_prof), p.size() * 8);
    double life = 0, real = 0;
    double first[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 256 * 8; w++) {
        life += p[w * 16 + 5]; real += p[w * 16 + 6];
        first[0] += (double)(p[w * 16 + 8] - p[w * 16 + 7]);                       // kernel entry -> first tile's start
        for (int i = 0; i < 4; i++) first[i + 1] += (double)(p[w * 16 + 9 + i] - p[w * 16 + 8 + i]);
    }
    printf("first tile: entry->tile %.0f  inputs %.0f  layer0 %.0f  layer1(+rendezvous) %.0f  layer2+out %.0f\n", first[0] / 2048, first[1] / 2048, first[2] / 2048,
           first[3] / 2048, first[4] / 2048);
    printf("wave lifetime: %.0f shader ticks, %.0f realtime ticks (100 MHz => %.2f us) => shader clock %.2f GHz\n", life / 2048, real / 2048, real / 2048 / 100.0,
           life / real * 0.1);
    double acc[4] = {0, 0, 0, 0}; int n = 0;
    for (int w = 0; w < 256 * 8; w++) {
        const unsigned long long *q = &p[w * 16];
        if (q[4] <= q[0]) continue;
        for (int i = 0; i < 4; i++) acc[i] += (double)(q[i + 1] - q[i]);
        n++;
    }
    printf("steady-state tile, mean over %d waves (s_memtime ticks = shader cycles/constant clock): inputs %.0f  layer0 %.0f  layer1 %.0f  layer2+out %.0f  total %.0f\n",
           n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, (acc[0] + acc[1] + acc[2] + acc[3]) / n);
    return 0;
}
