// experiment: s_memtime stamps of the phases of k_sampler_update (one workgroup = one ray), stock-like sizes
#define HS_SAMPLER_PROFILE 1
#include "../../holoscene_amd/csrc/sampler.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main() {
    const int m_old = getenv("M_OLD") ? atoi(getenv("M_OLD")) : 320, s_new = getenv("S_NEW") ? atoi(getenv("S_NEW")) : 64;
    const int R = 1024, ld = m_old + s_new;
    std::vector<float> z(R * ld), sd(R * ld), nz(R * s_new), ns(R * s_new), beta(R);
    unsigned s = 1;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (int r = 0; r < R; r++) {
        const float c = 0.5f + 2.f * rnd(), rad = 0.05f + 0.4f * rnd(), off = std::max(0.f, rnd() - 0.3f);
        std::vector<float> a(m_old), b(s_new);
        for (auto &v : a) v = 3.f * rnd();
        for (auto &v : b) v = 3.f * rnd();
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        for (int i = 0; i < m_old; i++) { z[r * ld + i] = a[i]; sd[r * ld + i] = std::fabs(a[i] - c) - rad + off; }
        for (int i = 0; i < s_new; i++) { nz[r * s_new + i] = b[i]; ns[r * s_new + i] = std::fabs(b[i] - c) - rad + off; }
        beta[r] = 0.05f + 0.5f * rnd();
    }
    float *dz, *ds, *dnz, *dns, *db, *db0, *dmax;
    hipMalloc(&dz, z.size() * 4); hipMalloc(&ds, z.size() * 4); hipMalloc(&dnz, nz.size() * 4); hipMalloc(&dns, nz.size() * 4);
    hipMalloc(&db, R * 4); hipMalloc(&db0, 4); hipMalloc(&dmax, 4);
    const float b0 = 0.01f;
    hipMemcpy(db0, &b0, 4, hipMemcpyHostToDevice); hipMemset(dmax, 0, 4);
    hipMemcpy(dnz, nz.data(), nz.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dns, ns.data(), ns.size() * 4, hipMemcpyHostToDevice);
    auto reset = [&] { hipMemcpy(dz, z.data(), z.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ds, sd.data(), z.size() * 4, hipMemcpyHostToDevice);
                       hipMemcpy(db, beta.data(), R * 4, hipMemcpyHostToDevice); };
    float *dout, *dcam, *ddir, *dx, *dx01;
    hipMalloc(&dout, R * 128 * 4); hipMalloc(&dcam, R * 3 * 4); hipMalloc(&ddir, R * 3 * 4); hipMalloc(&dx, R * 128 * 3 * 4); hipMalloc(&dx01, R * 128 * 3 * 4);
    hipMemset(dcam, 0, R * 3 * 4); hipMemset(ddir, 0, R * 3 * 4);
    const bool fused = getenv("FUSED_DRAW") != nullptr;      // FUSED_DRAW=1: hs_sampler_update_draw (the launch of the device-controlled loop)
    auto run = [&] { return fused ? hs_sampler_update_draw(dz, ds, ld, m_old, dnz, dns, s_new, db, db0, 0.1f, 10, dmax, R, nullptr, 1e-7f, 128, dout, dcam, ddir, 1.f, dx, dx01, nullptr)
                                  : hs_sampler_update(dz, ds, ld, m_old, dnz, dns, s_new, db, db0, 0.1f, 10, dmax, R, nullptr, nullptr, nullptr); };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) { reset(); run(); }
    hipDeviceSynchronize();
    float best = 1e9;
    for (int i = 0; i < 10; i++) { reset(); hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms); }
    printf("m = %d: %.1f us per launch (best of 10)\n", ld, best * 1e3);
    std::vector<unsigned long long> p(1024 * 8);
    hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(g_sampler_prof), p.size() * 8);
    std::vector<float> bo(R); hipMemcpy(bo.data(), db, R * 4, hipMemcpyDeviceToHost);
    const char *names[6] = {"stage old + new into LDS", "merge by rank", "write back + d*", "bound at beta0", "line search", "fused draw"};
    const int np = fused ? 6 : 5;
    for (int pass = 0; pass < 2; pass++) {
        double acc[6] = {0}; int n = 0; unsigned long long t0 = ~0ull, t1 = 0;
        for (int r = 0; r < R; r++) {
            const bool searched = bo[r] != b0;
            if (searched != (pass == 0)) continue;
            const unsigned long long *q = &p[r * 8];
            for (int i = 0; i < np; i++) acc[i] += (double)(q[i + 1] - q[i]);
            t0 = std::min(t0, q[0]); t1 = std::max(t1, q[np]); n++;
        }
        printf("%s rays: %d; first start .. last end %llu ticks\n", pass == 0 ? "searching" : "settled at beta0", n, n ? t1 - t0 : 0ull);
        for (int i = 0; i < np; i++) printf("  %-28s %8.0f ticks\n", names[i], n ? acc[i] / n : 0.0);
    }
    return 0;
}
