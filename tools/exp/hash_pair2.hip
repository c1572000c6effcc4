// experiment (round 5, session 3): what the value-only gather sweep (k_hash_fwd_pair<2, false>, 131 072 ray-ordered points x 16 levels, bf16-word
// output) costs in three forms.  The sweep is bound by the texture addresser (busy 0.81; a divergent 8-byte gather = 33 cycles per
// wave instruction, the three coordinate loads and the half-empty store ~70 more per wave):
//   A  the shipped form: a wave = 32 points x ONE level, lanes (2p, 2p + 1) = corners (x, x + 1)
//   B  a wave = 32 points x the TWO levels its XCD owns (coordinates loaded once, the two levels' words stored by the even / odd lane)
//   C  form B over a table of bf16 WORDS (both channels of an entry in 4 bytes: dword gathers, 16 addresser cycles instead of 32)
// build: hipcc -O3 --offload-arch=gfx950 tools/exp/hash_pair2.hip -o tools/exp/hash_pair2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstring>

struct Lv { float scale; unsigned res, offset, table, hashed; };
struct Lvs { Lv v[16]; };

__device__ __forceinline__ float sstep(float t) { return t * t * (3.f - 2.f * t); }
__device__ __forceinline__ unsigned cell(const Lv &l, unsigned gx, unsigned gy, unsigned gz) {
    unsigned idx;
    if (l.hashed) idx = gx ^ (gy * 2654435761u) ^ (gz * 805459861u);
    else idx = gx + gy * l.res + gz * l.res * l.res;
    if ((l.table & (l.table - 1u)) == 0u) return idx & (l.table - 1u);
    return idx >= l.table ? idx % l.table : idx;
}
__device__ __forceinline__ float dpp_swap(float p) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0xB1, 0xf, 0xf, true)); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_t));
}

// ---- A
template <bool WORDS>
__global__ __launch_bounds__(256) void kA(const float *__restrict__ x, const void *__restrict__ embv, unsigned *__restrict__ out, unsigned B, Lvs ls,
                                          unsigned n_chunks, unsigned xcd_mask = 0xffu) {
    const unsigned bid = blockIdx.x, xcd = bid & 7u, j = bid >> 3, slot = j / n_chunks, chunk = j - slot * n_chunks;
    if (!((xcd_mask >> xcd) & 1u)) return;
    const unsigned level = (slot & 1u) ? (slot * 8u + 7u - xcd) : (slot * 8u + xcd);
    const unsigned t = chunk * 256 + threadIdx.x, b = t >> 1, xb = t & 1u;
    if (b >= B) return;
    const Lv l = ls.v[level];
    float w[3]; unsigned c[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { float p = x[b * 3 + d] * l.scale; float f = floorf(p); c[d] = (unsigned)f; w[d] = sstep(p - f); }
    float2 e[4];
#pragma unroll
    for (int yz = 0; yz < 4; yz++) {
        const unsigned ci = cell(l, c[0] + xb, c[1] + (yz & 1), c[2] + (yz >> 1));
        if (WORDS) {
            const unsigned wd = reinterpret_cast<const unsigned *>(embv)[l.offset + ci];
            e[yz] = make_float2(__uint_as_float(wd << 16), __uint_as_float(wd & 0xffff0000u));
        } else {
            e[yz] = reinterpret_cast<const float2 *>(embv)[l.offset + ci];
        }
    }
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int yz = 0; yz < 4; yz++) {
        const float wt = (xb ? w[0] : 1 - w[0]) * ((yz & 1) ? w[1] : 1 - w[1]) * ((yz >> 1) ? w[2] : 1 - w[2]);
        const float p0 = wt * e[yz].x, p1 = wt * e[yz].y;
        a0 += p0; a0 += dpp_swap(p0);
        a1 += p1; a1 += dpp_swap(p1);
    }
    if (xb == 0) out[(size_t)level * B + b] = pack_bf16(a0, a1);
}


// ---- E: form A with the coordinate loads and the store on HALF the lanes: lane l < 32 loads the coordinates of the wave's point l (three dword
// loads over 8 quads instead of 16) and hands them to lanes 2l, 2l + 1 through the LDS crossbar (ds_bpermute: no addresser work); the 32 results
// are moved to lanes 0..31 the same way and stored from there
__global__ __launch_bounds__(256) void kE(const float *__restrict__ x, const float2 *__restrict__ emb, unsigned *__restrict__ out, unsigned B, Lvs ls,
                                          unsigned n_chunks) {
    const unsigned bid = blockIdx.x, xcd = bid & 7u, j = bid >> 3, slot = j / n_chunks, chunk = j - slot * n_chunks;
    const unsigned level = (slot & 1u) ? (slot * 8u + 7u - xcd) : (slot * 8u + xcd);
    const unsigned t = chunk * 256 + threadIdx.x, b = t >> 1, xb = t & 1u, lane = threadIdx.x & 63u;
    const unsigned wave_b0 = (chunk * 256 + (threadIdx.x & ~63u)) >> 1;        // first point of this wave
    if (wave_b0 >= B) return;
    const Lv l = ls.v[level];
    const float2 *g = emb + l.offset;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (lane < 32 && wave_b0 + lane < B) { const float *xp = x + (size_t)(wave_b0 + lane) * 3; px = xp[0]; py = xp[1]; pz = xp[2]; }
    const int src = (int)(lane >> 1) << 2;
    const float ps[3] = {__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(px))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(py))),
                         __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(pz)))};
    float w[3]; unsigned c[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { float p = ps[d] * l.scale; float f = floorf(p); c[d] = (unsigned)f; w[d] = sstep(p - f); }
    float2 e[4];
#pragma unroll
    for (int yz = 0; yz < 4; yz++) e[yz] = g[cell(l, c[0] + xb, c[1] + (yz & 1), c[2] + (yz >> 1))];
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int yz = 0; yz < 4; yz++) {
        const float wt = (xb ? w[0] : 1 - w[0]) * ((yz & 1) ? w[1] : 1 - w[1]) * ((yz >> 1) ? w[2] : 1 - w[2]);
        const float p0 = wt * e[yz].x, p1 = wt * e[yz].y;
        a0 += p0; a0 += dpp_swap(p0);
        a1 += p1; a1 += dpp_swap(p1);
    }
    const unsigned word = pack_bf16(a0, a1);
    const unsigned mine = (unsigned)__builtin_amdgcn_ds_bpermute((int)((lane & 31u) << 3), (int)word);     // lane l < 32 <- lane 2 l
    if (lane < 32 && wave_b0 + lane < B) out[(size_t)level * B + wave_b0 + lane] = mine;
    (void)b;
}


// ---- V: ONE pass over the points for TWO tables of the same geometry (the geometry table's and the colour table's gathers of the rendered
// samples share positions, cells and weights): locate and index once, eight gathers per lane
__global__ __launch_bounds__(256) void kV(const float *__restrict__ x, const float2 *__restrict__ embA, const float2 *__restrict__ embB, unsigned *__restrict__ outA,
                                          unsigned *__restrict__ outB, unsigned B, Lvs ls, unsigned n_chunks) {
    const unsigned bid = blockIdx.x, xcd = bid & 7u, j = bid >> 3, slot = j / n_chunks, chunk = j - slot * n_chunks;
    const unsigned level = (slot & 1u) ? (slot * 8u + 7u - xcd) : (slot * 8u + xcd);
    const unsigned t = chunk * 256 + threadIdx.x, b = t >> 1, xb = t & 1u;
    if (b >= B) return;
    const Lv l = ls.v[level];
    float w[3]; unsigned c[3];
#pragma unroll
    for (int d = 0; d < 3; d++) { float p = x[b * 3 + d] * l.scale; float f = floorf(p); c[d] = (unsigned)f; w[d] = sstep(p - f); }
    float2 ea[4], eb[4];
#pragma unroll
    for (int yz = 0; yz < 4; yz++) {
        const unsigned ci = l.offset + cell(l, c[0] + xb, c[1] + (yz & 1), c[2] + (yz >> 1));
        ea[yz] = embA[ci];
        eb[yz] = embB[ci];
    }
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
    for (int yz = 0; yz < 4; yz++) {
        const float wt = (xb ? w[0] : 1 - w[0]) * ((yz & 1) ? w[1] : 1 - w[1]) * ((yz >> 1) ? w[2] : 1 - w[2]);
        float p;
        p = wt * ea[yz].x; a0 += p; a0 += dpp_swap(p);
        p = wt * ea[yz].y; a1 += p; a1 += dpp_swap(p);
        p = wt * eb[yz].x; b0 += p; b0 += dpp_swap(p);
        p = wt * eb[yz].y; b1 += p; b1 += dpp_swap(p);
    }
    if (xb == 0) { outA[(size_t)level * B + b] = pack_bf16(a0, a1); outB[(size_t)level * B + b] = pack_bf16(b0, b1); }
}

// ---- B / C: two levels per wave.  WORDS: the table is uint32 (bf16 pair) per entry
template <bool WORDS>
__global__ __launch_bounds__(256) void kB(const float *__restrict__ x, const void *__restrict__ embv, unsigned *__restrict__ out, unsigned B, Lvs ls,
                                          unsigned n_chunks) {
    const unsigned bid = blockIdx.x, xcd = bid & 7u, chunk = bid >> 3;
    const unsigned t = chunk * 256 + threadIdx.x, b = t >> 1, xb = t & 1u;
    if (b >= B) return;
    const float px = x[b * 3], py = x[b * 3 + 1], pz = x[b * 3 + 2];
    float res0[2], res1[2];
    float2 e[2][4];
    float w[2][3];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const Lv l = ls.v[k ? 15 - xcd : xcd];
        unsigned c[3];
        const float ps[3] = {px, py, pz};
#pragma unroll
        for (int d = 0; d < 3; d++) { float p = ps[d] * l.scale; float f = floorf(p); c[d] = (unsigned)f; w[k][d] = sstep(p - f); }
#pragma unroll
        for (int yz = 0; yz < 4; yz++) {
            const unsigned ci = cell(l, c[0] + xb, c[1] + (yz & 1), c[2] + (yz >> 1));
            if (WORDS) {
                const unsigned wd = reinterpret_cast<const unsigned *>(embv)[l.offset + ci];
                e[k][yz] = make_float2(__uint_as_float(wd << 16), __uint_as_float(wd & 0xffff0000u));
            } else {
                e[k][yz] = reinterpret_cast<const float2 *>(embv)[l.offset + ci];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int yz = 0; yz < 4; yz++) {
            const float wt = (xb ? w[k][0] : 1 - w[k][0]) * ((yz & 1) ? w[k][1] : 1 - w[k][1]) * ((yz >> 1) ? w[k][2] : 1 - w[k][2]);
            const float p0 = wt * e[k][yz].x, p1 = wt * e[k][yz].y;
            a0 += p0; a0 += dpp_swap(p0);
            a1 += p1; a1 += dpp_swap(p1);
        }
        res0[k] = a0; res1[k] = a1;
    }
    // even lane: level xcd's word; odd lane: level 15 - xcd's (taken from its even neighbour)
    const float o0 = xb ? dpp_swap(res0[1]) : res0[0], o1 = xb ? dpp_swap(res1[1]) : res1[0];
    const unsigned level = xb ? 15 - xcd : xcd;
    out[(size_t)level * B + b] = pack_bf16(o0, o1);
}

int main() {
    const unsigned R = 1024, S = 128, B = R * S;
    std::vector<float> hx((size_t)B * 3);
    unsigned s = 7; auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.f; };
    for (unsigned r = 0; r < R; r++) {
        float o[3] = {0.3f + 0.4f * rnd(), 0.3f + 0.4f * rnd(), 0.02f}, d[3] = {0.5f * (rnd() - 0.5f), 0.5f * (rnd() - 0.5f), 0.9f};
        std::vector<float> zs(S);
        for (auto &z : zs) z = rnd();
        std::sort(zs.begin(), zs.end());
        for (unsigned i = 0; i < S; i++)
            for (int k = 0; k < 3; k++) { float v = o[k] + zs[i] * d[k]; hx[((size_t)r * S + i) * 3 + k] = fminf(fmaxf(v, 0.f), 1.f); }
    }
    Lvs ls; unsigned off = 0;
    const double pls = std::exp2(std::log2(2048.0 / 16.0) / 15.0);
    for (int l = 0; l < 16; l++) {
        const float scale = (float)(16.0 * std::pow(pls, l) - 1.0);
        const unsigned res = (unsigned)std::ceil(scale) + 1u;
        unsigned long long full = (unsigned long long)(res + 1) * (res + 1) * (res + 1), dense = (unsigned long long)res * res * res;
        unsigned tab = full > (1u << 19) ? (1u << 19) : (unsigned)((full + 7) / 8 * 8);
        ls.v[l] = {scale, res, off, tab, dense > tab};
        off += tab;
    }
    printf("entries %u  B %u\n", off, B);
    std::vector<float> he((size_t)off * 2);
    for (auto &v : he) v = (rnd() - 0.5f) * 2e-4f;
    std::vector<unsigned> hw(off);
    for (unsigned i = 0; i < off; i++) {
        unsigned a, b2; memcpy(&a, &he[2 * i], 4); memcpy(&b2, &he[2 * i + 1], 4);
        hw[i] = ((a + 0x8000u) >> 16) | ((b2 + 0x8000u) & 0xffff0000u);
    }
    float *dx; float2 *de; unsigned *dw, *dout;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&de, he.size() * 4); hipMalloc(&dw, hw.size() * 4); hipMalloc(&dout, (size_t)16 * B * 4);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(de, he.data(), he.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const unsigned n_chunks = (2 * B + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        hipDeviceSynchronize();
        float best = 1e9f, tot = 0.f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; i++) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = fminf(best, ms / 20); tot += ms / 20;
        }
        std::vector<unsigned> ho(16 * (size_t)B);
        hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        unsigned long long cs = 0; for (auto v : ho) cs += v;
        printf("%-28s best %.1f us  mean %.1f us  checksum %llx\n", name, best * 1e3, tot / 5 * 1e3, cs);
    };
    run("A one level per wave", [&] { kA<false><<<n_chunks * 16, 256>>>(dx, de, dout, B, ls, n_chunks); });
    std::vector<unsigned> refo(16 * (size_t)B);
    hipMemcpy(refo.data(), dout, refo.size() * 4, hipMemcpyDeviceToHost);
    run("B two levels per wave", [&] { kB<false><<<n_chunks * 8, 256>>>(dx, de, dout, B, ls, n_chunks); });
    {
        std::vector<unsigned> ho(16 * (size_t)B);
        hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0, first = 0; for (size_t i = 0; i < ho.size(); i++) if (ho[i] != refo[i]) { if (!bad) first = i; bad++; }
        printf("B vs A: %zu differing words (first at level %zu point %zu: %08x vs %08x)\n", bad, first / B, first % B, bad ? ho[first] : 0, bad ? refo[first] : 0);
    }
    run("C two levels, bf16 words", [&] { kB<true><<<n_chunks * 8, 256>>>(dx, dw, dout, B, ls, n_chunks); });
    run("D one level, bf16 words", [&] { kA<true><<<n_chunks * 16, 256>>>(dx, dw, dout, B, ls, n_chunks); });
    for (unsigned m : {0x1fu, 0xe0u, 0x01u, 0x10u, 0x20u, 0x80u}) {
        char nm[64]; snprintf(nm, sizeof nm, "A, XCD mask %02x only", m);
        run(nm, [&] { kA<false><<<n_chunks * 16, 256>>>(dx, de, dout, B, ls, n_chunks, m); });
    }
    run("E half-wave loads / store", [&] { kE<<<n_chunks * 16, 256>>>(dx, de, dout, B, ls, n_chunks); });
    float2 *de2; unsigned *dout2; hipMalloc(&de2, he.size() * 4); hipMalloc(&dout2, (size_t)16 * B * 4);
    hipMemcpy(de2, he.data(), he.size() * 4, hipMemcpyHostToDevice);
    run("A twice (two tables, two launches)", [&] { kA<false><<<n_chunks * 16, 256>>>(dx, de, dout, B, ls, n_chunks); kA<false><<<n_chunks * 16, 256>>>(dx, de2, dout2, B, ls, n_chunks); });
    run("V two tables, one pass", [&] { kV<<<n_chunks * 16, 256>>>(dx, de, de2, dout, dout2, B, ls, n_chunks); });
    run("A again", [&] { kA<false><<<n_chunks * 16, 256>>>(dx, de, dout, B, ls, n_chunks); });
    return 0;
}
