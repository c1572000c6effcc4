// experiment: per-phase cycle counts (s_memtime) of the last wave tile of k_trunk_fwd2
#define HS_TRUNK2_PROFILE 1
#include "../../holoscene_amd/csrc/trunk_mlp2.hip"
#include <cstdio>
#include <vector>
extern "C" int hs_sdf_mlp2_pack(const float *, int32_t, const float *, const float *, const float *, const float *, const float *, int32_t, void *, void *, void *, float *, int32_t, void *);
// minimal copy of the pack launch (the pack kernel lives in wave_tile.h)
static int pack(const float *W0, const float *b, const float *W1, const float *W2, void *W0f, void *W1f, void *W2f, float *bias) {
    const int slots = K0S * NT * 64 + HS * NT * 64 + HS * 64 + kBias;
    k_sdf_pack2<<<(slots + 255) / 256, 256>>>(W0, 71, b, W1, b + 256, W2, b + 512, 32, (uint16_t *)W0f, (uint16_t *)W1f, (uint16_t *)W2f, bias, 1.f);
    return 0;
}
int main(int argc, char **argv) {
    const int64_t Bp = argc > 1 ? atoll(argv[1]) : 104448, M = 4 * Bp;
    float *x, *feat, *dydx, *W0, *W1, *W2, *b, *bias, *Y;
    void *W0f, *W12f, *H0, *H1, *Xp;
    hipMalloc(&x, Bp * 12); hipMalloc(&feat, Bp * 128); hipMalloc(&dydx, Bp * 16 * 24); hipMalloc(&Y, M * 32 * 4);
    hipMalloc(&H0, M * 512); hipMalloc(&H1, M * 512); hipMalloc(&Xp, M * 160);
    hipMalloc(&W0, 256 * 71 * 4); hipMalloc(&W1, 256 * 256 * 4); hipMalloc(&W2, 32 * 256 * 4); hipMalloc(&b, 1024 * 4);
    hipMalloc(&W0f, kW0F * 2); hipMalloc(&W12f, (kW1F + kW2F) * 2); hipMalloc(&bias, kBias * 4);
    std::vector<float> h(Bp * 96);
    unsigned s = 1;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto &v : h) v = rnd();
    hipMemcpy(x, h.data(), Bp * 12, hipMemcpyHostToDevice); hipMemcpy(feat, h.data(), Bp * 128, hipMemcpyHostToDevice); hipMemcpy(dydx, h.data(), Bp * 384, hipMemcpyHostToDevice);
    for (auto &v : h) v = rnd() * 0.1f;
    hipMemcpy(W0, h.data(), 256 * 71 * 4, hipMemcpyHostToDevice); hipMemcpy(W1, h.data(), 256 * 256 * 4, hipMemcpyHostToDevice);
    hipMemcpy(W2, h.data(), 32 * 256 * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), 1024 * 4, hipMemcpyHostToDevice);
    void *W2f = (char *)W12f + (size_t)kW1F * 2;
    pack(W0, b, W1, W2, W0f, W12f, W2f, bias);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) hs_trunk_mlp2_fwd(x, feat, dydx, W0f, W12f, W2f, bias, 32, H0, H1, Y, Xp, M, 0.5f, nullptr, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) hs_trunk_mlp2_fwd(x, feat, dydx, W0f, W12f, W2f, bias, 32, H0, H1, Y, Xp, M, 0.5f, nullptr, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%lld rows: %.1f us per launch\n", (long long)M, ms * 100);
    std::vector<unsigned long long> p(256 * 8 * 8);
    hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(g_trunk2_prof), p.size() * 8);
    double acc[4] = {0, 0, 0, 0}; int n = 0;
    for (int w = 0; w < 256 * 8; w++) {
        const unsigned long long *q = &p[w * 8];
        if (q[4] <= q[0]) continue;
        for (int i = 0; i < 4; i++) acc[i] += (double)(q[i + 1] - q[i]);
        n++;
    }
    printf("stamped tile, mean over %d waves (shader cycles): inputs %.0f  layer0 %.0f  layer1 %.0f  layer2+out %.0f  total %.0f\n", n, acc[0] / n, acc[1] / n, acc[2] / n,
           acc[3] / n, (acc[0] + acc[1] + acc[2] + acc[3]) / n);
    return 0;
}
