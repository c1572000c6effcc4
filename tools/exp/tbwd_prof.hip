// experiment: s_memtime stamps of the phases of one 128-row tile of k_trunk_bwd (the third tile of every workgroup), stock shapes
#define HS_TBWD_PROFILE 1
#include "../../holoscene_amd/csrc/sdf_mlp.hip"
#include <cstdio>
#include <vector>
int main() {
    const int64_t Bp = 104448, M = 4 * Bp;
    void *g, *H1, *H0, *W2t, *W1t, *W0t, *gA1, *gA0; float *gb, *gf, *gd, *dw2;
    hipMalloc(&g, M * 64); hipMalloc(&H1, M * 512); hipMalloc(&H0, M * 512); hipMalloc(&gA1, M * 512); hipMalloc(&gA0, M * 512);
    hipMalloc(&W2t, 256 * 32 * 2); hipMalloc(&W1t, 256 * 256 * 2); hipMalloc(&W0t, 256 * 256 * 2); hipMalloc(&gb, 4096);
    hipMalloc(&gf, 16 * Bp * 2 * 4); hipMalloc(&gd, 16 * Bp * 6 * 4); hipMalloc(&dw2, (size_t)256 * 32 * 256 * 4);
    std::vector<uint16_t> h(M * 256);
    unsigned s = 1;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((s >> 12) & 0x1ff)) ^ (uint16_t)((s >> 3) & 0x8000); }   // +-0.01..0.03
    hipMemcpy(H1, h.data(), M * 512, hipMemcpyHostToDevice); hipMemcpy(H0, h.data(), M * 512, hipMemcpyHostToDevice);
    hipMemcpy(g, h.data(), M * 64, hipMemcpyHostToDevice);
    hipMemcpy(W2t, h.data(), 256 * 32 * 2, hipMemcpyHostToDevice); hipMemcpy(W1t, h.data(), 256 * 256 * 2, hipMemcpyHostToDevice);
    hipMemcpy(W0t, h.data(), 256 * 256 * 2, hipMemcpyHostToDevice);
    hipMemset(gb, 0, 4096);
    auto run = [&] { return hs_trunk_mlp_bwd(g, 32, H1, H0, W2t, W1t, gA1, gA0, gb, gb + 256, W0t, gf, gd, 16, 2, 0.5f, M, gb + 512, dw2, nullptr); };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("rc %d\n", run()); run(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) run();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%lld: %.1f us per launch\n", (long long)M, ms * 100);
    std::vector<unsigned long long> p(256 * 16);
    hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(g_tbwd_prof), p.size() * 8);
    const char *names[9] = {"stage g", "G1 = g.W2 (+H1 tile in)", "dW2 += g^T.H1", "epilogue 1", "store gA1 + colsum", "G0 = gA1.W1 (+H0 tile in)", "epilogue 0", "store gA0 + colsum",
                            "gX = gA0.W0 + g_feat/g_dydx"};
    double acc[9] = {0}; int n = 0;
    for (int b = 0; b < 256; b++) {
        const unsigned long long *q = &p[b * 16];
        if (q[9] <= q[0]) continue;
        for (int i = 0; i < 9; i++) acc[i] += (double)(q[i + 1] - q[i]);
        n++;
    }
    double tot = 0; for (int i = 0; i < 9; i++) tot += acc[i] / n;
    for (int i = 0; i < 9; i++) printf("  %-32s %8.0f cycles  %4.1f %%\n", names[i], acc[i] / n, 100 * acc[i] / n / tot);
    { double a = 0, b = 0, c = 0; int m = 0; for (int bb = 0; bb < 256; bb++) { const unsigned long long *q = &p[bb * 16]; if (q[9] <= q[0]) continue; a += q[10] - q[8]; b += q[11] - q[10]; c += q[9] - q[11]; m++; }
      printf("    of the last: product %.0f, pack + barrier %.0f, g_feat / g_dydx stores %.0f\n", a / m, b / m, c / m); }
    printf("  tile total %.0f cycles over %d workgroups\n", tot, n);
    return 0;
}
