// holoscene_amd/csrc/trunk_rr.hip -- the SDF trunk of the RENDERED samples in reverse-over-reverse form, "wave tile" kernels (gfx950).
//
// What the rendered samples need from the trunk (model/network.py:273-301: ObjectImplicitNetworkGrid.get_outputs) is the K per-object
// SDFs, their minimum and ONE gradient, d min / dx -- which the reference takes by reverse mode (autograd.grad(..., create_graph=True),
// :293-299) and differentiates once more in loss.backward().  The value+Jacobian kernels (trunk_mlp2.hip, sdf_mlp.hip: k_trunk_bwd)
// carry three input tangents beside the value instead -- 4 rows per sample through every layer, forward and backward, which also yields
// all K gradients, but only the 4 R Eikonal points need those (network.py:212-254).  Here the samples go the reference's way, in closed
// form (tools/exp/rr_trunk_math.py checks the formulas against autograd's double backward in float64):
//
//   forward   a0 = W0 xt + b0, h0 = sp(a0);  a1 = W1 h0 + b1, h1 = sp(a1);  y = W2 h1 + b2;  k* = argmin y            (k_rr_fwd_value)
//             u1 = W2[k*], v1 = u1 s1;  u0 = W1^T v1, v0 = u0 s0;  ux = W0^T v0;  d min/dx = E^T ux                    (k_rr_fwd_grad)
//   backward  ux~ = E g~;  v0~ = W0 ux~;  u0~ = v0~ s0, a0' = v0~ u0 s0';  v1~ = W1 u0~;  u1~ = v1~ s1, a1' = v1~ u1 s1'    (k_rr_bwd_grad)
//             h1~ = W2^T y~;  a1~ = a1' + h1~ s1;  h0~ = W1^T a1~;  a0~ = a0' + h0~ s0;  xt~ = W0^T a0~                      (k_rr_bwd_value)
//   weights   dW1 = a1~^T h0 + v1^T u0~,  dW0 = a0~^T xt + v0^T ux~,  dW2 = y~^T h1 + onehot(k*)^T u1~                        (wgrad_pairs.hip)
//
// with s = sigmoid(100 a) = 1 - exp(-100 h), s' = 100 s (1 - s), E = d xt / dx (positional encoding + hash dy_dx): 6 row-passes of the
// MLP per sample instead of 12, on rows that are SAMPLES (100 352 at the stock size) instead of value+tangent quadruples (417 792).
// Two kernel skeletons, each the structure of sdf_mlp2.hip (a wave owns 32 samples end to end, activations stay in registers as the next
// product's B fragments, the 256 x 256 matrix LDS-resident in fragment order, the narrow one streamed from L2):
//   "down"  W0 (L2) -> W1 (LDS) [-> W2 (LDS)]      k_rr_fwd_value, k_rr_bwd_grad
//   "up"    [W2^T (LDS) ->] W1^T (LDS) -> W0^T (L2)    k_rr_fwd_grad, k_rr_bwd_value
// Activations that cross kernels are stored TILE-PACKED ("TP"): [tile of 32 samples][k-step 0..15][lane 0..63] x 16 bytes = exactly
// the four packed words lane (sample, half) holds for that k-step -- one coalesced 1 KB wave access per k-step in every producer and
// consumer, no layout conversion anywhere (the weight-gradient kernel scatters the 8-byte runs into its row-major LDS tiles).
#include "launch_util.h"
#include "wave_tile.h"
#include "trunk_pack.h"
#include "split_bwd.h"

namespace {

__global__ __launch_bounds__(256) void k_rr_pack(const float *__restrict__ W0, int ld0, const float *__restrict__ W1, const float *__restrict__ W2, int d_out,
                                                 uint16_t *__restrict__ W1Tf, uint16_t *__restrict__ W0Tf, uint16_t *__restrict__ W2Tf,
                                                 float *__restrict__ W2tab) {
    rr_pack_slot(blockIdx.x * 256 + threadIdx.x, W0, ld0, W1, W2, d_out, W1Tf, W0Tf, W2Tf, W2tab);
}

// Every weight image a training pass of the trunk needs, in ONE launch (trunk_pack.h: trunk_pack_all_slot).  Three launches of ~5 us before; a
// dispatch costs that whatever it does.
__global__ __launch_bounds__(256) void k_trunk_pack_all(const float *__restrict__ W0, int ld0, int f_in, const float *__restrict__ b0, const float *__restrict__ W1,
                                                        const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2, int d_out,
                                                        uint16_t *__restrict__ W0f, uint16_t *__restrict__ W1f, uint16_t *__restrict__ W2f,
                                                        float *__restrict__ bias, uint16_t *__restrict__ W1Tf, uint16_t *__restrict__ W0Tf,
                                                        uint16_t *__restrict__ W2Tf, float *__restrict__ W2tab, uint16_t *__restrict__ w1t,
                                                        uint16_t *__restrict__ w2t, uint16_t *__restrict__ w0t) {
    trunk_pack_all_slot(blockIdx.x * 256 + threadIdx.x, W0, ld0, f_in, b0, W1, b1, W2, b2, d_out, W0f, W1f, W2f, bias, W1Tf, W0Tf, W2Tf, W2tab, w1t, w2t, w0t);
}

// ---------------------------------------------------------------------------------------------------------------- small helpers
__device__ __forceinline__ float softplus100(float v) {      // torch.nn.Softplus(beta=100): linear above 100 v = 20
    const float e = __builtin_amdgcn_exp2f(fminf(v * kC, 28.8539008f));
    const float lg = __builtin_amdgcn_logf(1.f + e) * (0.69314718f * 0.01f);
    return fmaxf(v, lg);
}
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// s = sigmoid(100 a) from the stored h = softplus100(a): exp(-100 h) = 1 - s
__device__ __forceinline__ float sig_of_h(float h) { return 1.f - __builtin_amdgcn_exp2f(-kC * h); }

// k-steps an LDS-resident weight fragment is read ahead of its MFMA in the backward kernels (and the separate forward-gradient kernel)
#ifndef HS_RR_AH
#define HS_RR_AH 2
#endif
#ifndef HS_NT_LOAD_LAST
#define HS_NT_LOAD_LAST 1
#endif
template <bool NT = false>      // NT: a non-temporal load -- a saved tensor at its LAST read
__device__ __forceinline__ uint4 tp_load(const uint16_t *__restrict__ T, int64_t tile, int s, int lane) {
    if constexpr (NT) {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(T + (((size_t)tile * HS + s) * 64 + lane) * 8));
        return make_uint4(v[0], v[1], v[2], v[3]);
    } else {
        return *reinterpret_cast<const uint4 *>(T + (((size_t)tile * HS + s) * 64 + lane) * 8);
    }
}
// NT: a non-temporal store -- for the saved activations whose reader is far away (the backward pass, the weight-gradient pass): 0.26 GB per
// k_rr_fwd launch that would otherwise push the hash tables, the weight images and the neighbours' tiles out of the caches on their way to memory
#ifndef HS_NT_FWD
#define HS_NT_FWD 1
#endif
#ifndef HS_NT_BG_U
#define HS_NT_BG_U 1
#endif
#ifndef HS_NT_BG_A
#define HS_NT_BG_A 1
#endif
#ifndef HS_NT_BV
#define HS_NT_BV 1
#endif
template <bool NT = false>
__device__ __forceinline__ void tp_store(uint16_t *__restrict__ T, int64_t tile, int s, int lane, const uint32_t *w4, bool ok) {
    uint4 v = ok ? make_uint4(w4[0], w4[1], w4[2], w4[3]) : make_uint4(0u, 0u, 0u, 0u);      // rows past the end hold zeros: the weight gradients sum whole tiles
    if constexpr (NT) {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t vv = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(vv, reinterpret_cast<u32x4_t *>(T + (((size_t)tile * HS + s) * 64 + lane) * 8));
    } else {
        *reinterpret_cast<uint4 *>(T + (((size_t)tile * HS + s) * 64 + lane) * 8) = v;
    }
}
__device__ __forceinline__ uint32_t word_of(const uint4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// resident image by LDS-DMA (sdf_mlp2.hip), `bytes` a multiple of kWaves KB
__device__ __forceinline__ void dma_fill(const void *src, void *dst, int bytes, int wave, int lane) {
    const int chunks = bytes / 1024;
    for (int c = wave; c < chunks; c += kWaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)src + (size_t)c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)((char *)dst + (size_t)c * 1024), 16, 0, 0);
}

// The derivative factors E[col][d] of this lane's 40 input columns are functions of x and dy_dx; both directions (d min / dx = E^T ux in
// the forward, ux~ = E g~ in the backward) walk them in input_column order: j < 18 positional encoding (octave 3 hh + j / 6; sin
// components, then cos), 18..33 hash level 8 hh + (j - 18) / 2, 34..36 the raw coordinate (half 0 only).
struct EFac {
    float pe[18];          // d (sin | cos)(f x_d) / d x_d of this lane's three octaves: f cos / -f sin  (one component each: column j <-> d = j % 3)
    float2 dy[8][3];       // jac_scale * dy_dx[level 8 hh + i][sample][d][c]
};
__device__ __forceinline__ EFac load_efac(const float *__restrict__ x, const float *__restrict__ dydx, int64_t gp, int64_t ld, int hh, float jac_scale, bool ok) {
    EFac E;
    asm volatile("" : "+v"(hh));        // opaque: the frequency factors and level offsets are otherwise per-lane loop invariants, kept (and spilled) across the tile loop
    const int64_t b = ok ? gp : 0;
    const float xs[3] = {x[b * 3], x[b * 3 + 1], x[b * 3 + 2]};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float f = hh ? (float)(8 << k) : (float)(1 << k);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float sn, cs;
            __sincosf(xs[d] * f, &sn, &cs);
            E.pe[6 * k + d] = f * cs;
            E.pe[6 * k + 3 + d] = -f * sn;
        }
    }
    const float *dp = dydx + ((size_t)(8 * hh) * ld + b) * 6;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const float2 t = *reinterpret_cast<const float2 *>(dp + (size_t)i * ld * 6 + 2 * d);
            E.dy[i][d] = make_float2(t.x * jac_scale, t.y * jac_scale);
        }
    return E;
}

// ================================================================================================================ tile machinery
// Workgroup b owns the CONTIGUOUS tiles [b T / G, (b + 1) T / G): at the stock size 3 136 tiles on 256 workgroups are 12 or 13 each -- one
// round of all eight waves and one of four or five, a wave alone on its SIMD -- instead of two full rounds on 136 compute units while 120
// stand idle (the strided assignment: 392 super-tiles of eight on 256 workgroups).
__device__ __forceinline__ int64_t tile_begin(int64_t ntiles) { return uniform64(ntiles * (int64_t)blockIdx.x / (int64_t)gridDim.x); }
__device__ __forceinline__ int64_t tile_end(int64_t ntiles) { return uniform64(ntiles * ((int64_t)blockIdx.x + 1) / (int64_t)gridDim.x); }
// One PHASE = the k-steps of ONE 32-neuron tile (one accumulator) with, in their shadow, slices of the previous tile's epilogue -- the
// single-tile form of wave_tile.h's phase2: half the accumulator and epilogue-staging registers of a quarter phase.  These kernels move
// 0.3-0.4 GB each for ~20 GFLOP: what they need registers for is loads in flight, not MFMA operands (a quarter-phase version spilled 40-80
// registers and ran at 1.5-3 TB/s).  A fragments `frag(s)` travel AHEAD k-steps in front of their MFMA.
template <int KS, int AHEAD, int E, int NSL, bool EPI, bool ZERO_C, class FragFn, class EpiFn>
__device__ __forceinline__ void phase1r(f32x16 &cur, const uint32_t *bin, FragFn frag, EpiFn epi) {
    bf16x8 ring[AHEAD + 1];
    static_for<(AHEAD < KS ? AHEAD : KS)>([&](auto sc) { constexpr int s = decltype(sc)::value; ring[s] = frag(s); });
    static_for<KS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AHEAD < KS) ring[(s + AHEAD) % (AHEAD + 1)] = frag(s + AHEAD);
        const bf16x8 b = frag_of(bin + 4 * s);
        if constexpr (ZERO_C && s == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)], b, zero, 0, 0, 0);
        } else {
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)], b, cur, 0, 0, 0);
        }
        if constexpr (EPI && s < E) {
            constexpr int lo = (s * NSL) / E, hi = ((s + 1) * NSL) / E;
            static_for<hi - lo>([&](auto jc) { epi(std::integral_constant<int, lo + decltype(jc)::value>{}); });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// the two TP pieces (k-steps 2 nt, 2 nt + 1) that hold tile nt's 8 words of this lane
struct TilePair { uint4 a, b; };
template <bool NT = false>
__device__ __forceinline__ TilePair tp_load_tile(const uint16_t *__restrict__ T, int64_t tile, int nt, int lane) {
    TilePair r;
    r.a = tp_load<NT>(T, tile, 2 * nt, lane);
    r.b = tp_load<NT>(T, tile, 2 * nt + 1, lane);
    return r;
}
// The loads of a phase's saved activations are `const __restrict__` and would otherwise be scheduled to the top of the tile (all 8 tiles of
// two tensors = 128 registers in flight: 60-70 spills).  An index the compiler cannot see through, produced where the request belongs,
// keeps each one there (volatile asm statements keep their order relative to the epilogues' anchors).
__device__ __forceinline__ int64_t here(int64_t tile) {
    uint32_t z = 0;
    asm volatile("" : "+v"(z));
    return tile + z;
}
__device__ __forceinline__ uint32_t tile_word(const TilePair &t, int p) { return p < 4 ? word_of(t.a, p) : word_of(t.b, p - 4); }   // p = 0..7

// ================================================================================================================ "down" kernels
// ---------------------------------------------------------------------------------------------------------------- forward, values
// sdf_mlp2.hip's function with plain-domain Softplus; H0 / H1 leave tile-packed, the assembled inputs as Xp [n, 80] (trunk_mlp2.hip's column order)
__global__ __launch_bounds__(kThreadsW, 2) void k_rr_fwd_value(const float *__restrict__ x, const float *__restrict__ feat, const uint16_t *__restrict__ W0f,
                                                                const uint16_t *__restrict__ W1f, const uint16_t *__restrict__ W2f,
                                                                const float *__restrict__ biasg, int d_out, uint16_t *__restrict__ H0t,
                                                                uint16_t *__restrict__ H1t, uint16_t *__restrict__ Xp, float *__restrict__ sdf_raw,
                                                                float *__restrict__ sdf, int64_t *__restrict__ idx, uint16_t *__restrict__ onehot,
                                                                int64_t n) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *W1l = lds;
    uint16_t *W2l = lds + kW1F;
    float *bias = reinterpret_cast<float *>(W2l + kW2F);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, h = lane >> 5;
    dma_fill(W1f, W1l, (kW1F + kW2F) * 2, wave, lane);
    for (int i = threadIdx.x; i < kBias; i += kThreadsW) bias[i] = biasg[i];
    __syncthreads();
    bool resident = false;
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const bf16x8 *W0v = reinterpret_cast<const bf16x8 *>(W0f) + lane;
    const bf16x8 *W1v = reinterpret_cast<const bf16x8 *>(W1l) + lane;
    const bf16x8 *W2v = reinterpret_cast<const bf16x8 *>(W2l) + lane;
    for (int64_t tile = tile_begin(ntiles) + wave, tile_e = tile_end(ntiles); tile < tile_e; tile += kWaves) {
        const int64_t gp = tile * kRows + row;
        const bool ok = gp < n;
        uint32_t hin[4 * K0S];
        {
            float v[40];
            const float x0 = ok ? x[gp * 3] : 0.f, x1 = ok ? x[gp * 3 + 1] : 0.f, x2 = ok ? x[gp * 3 + 2] : 0.f;
            const float xs[3] = {x0, x1, x2};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float f = h ? (float)(8 << k) : (float)(1 << k);
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    float sn, cs;
                    __sincosf(xs[d] * f, &sn, &cs);
                    v[6 * k + d] = sn;
                    v[6 * k + 3 + d] = cs;
                }
            }
            const float4 *fp = reinterpret_cast<const float4 *>(feat + (ok ? gp : 0) * 32 + 16 * h);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 t = ok ? fp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                v[18 + 4 * i] = t.x; v[19 + 4 * i] = t.y; v[20 + 4 * i] = t.z; v[21 + 4 * i] = t.w;
            }
            v[34] = h ? 0.f : x0; v[35] = h ? 0.f : x1; v[36] = h ? 0.f : x2;
            v[37] = v[38] = v[39] = 0.f;
#pragma unroll
            for (int j = 0; j < 40; j += 2) hin[j >> 1] = ok ? pack2(v[j], v[j + 1]) : 0u;
            if (ok) {
                uint4 *xp = reinterpret_cast<uint4 *>(Xp + gp * 80 + 40 * h);
#pragma unroll
                for (int i = 0; i < 5; i++) xp[i] = make_uint4(hin[4 * i], hin[4 * i + 1], hin[4 * i + 2], hin[4 * i + 3]);
            }
        }
        uint32_t h0p[64], h1p[64];
        f32x16 acc[2];
        // epilogue of finished tile nd: slices 0..7 Softplus + pack of one register pair, 8..9 its two k-steps leave as TP
        auto epi = [&](auto slc, const f32x16 &src, uint32_t *hp, int nd, uint16_t *T) {
            constexpr int sl = decltype(slc)::value;
            if constexpr (sl < 8) hp[8 * nd + sl] = anchor(pack2(softplus100(src[2 * sl]), softplus100(src[2 * sl + 1])));
            else tp_store<HS_NT_FWD>(T, tile, 2 * nd + (sl - 8), lane, hp + 8 * nd + 4 * (sl - 8), ok);
        };
        {
            bf16x8 w0[3][K0S];
            uint32_t zoff = 0;
            asm volatile("" : "+v"(zoff));
            const bf16x8 *W0q = W0v + zoff;
            static_for<K0S>([&](auto sc) { constexpr int s = decltype(sc)::value; w0[0][s] = W0q[(size_t)(s * NT + 0) * 64]; w0[1][s] = W0q[(size_t)(s * NT + 1) * 64]; });
            static_for<NT>([&](auto nc) {
                constexpr int nt = decltype(nc)::value;
                init_acc(acc[nt & 1], bias + 32 * nt, h);
                if constexpr (nt + 2 < NT)
                    static_for<K0S>([&](auto sc) { constexpr int s = decltype(sc)::value; w0[(nt + 2) % 3][s] = W0q[(size_t)(s * NT + nt + 2) * 64]; });
                auto f0 = [&](int s) { return w0[nt % 3][s]; };
                if constexpr (nt == 0) phase1r<K0S, 0, K0S, 10, false, false>(acc[0], hin, f0, [](auto) {});
                else phase1r<K0S, 0, K0S, 10, true, false>(acc[nt & 1], hin, f0, [&](auto slc) { epi(slc, acc[(nt & 1) ^ 1], h0p, nt - 1, H0t); });
            });
        }
        if (!resident) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            resident = true;
        }
        static_for<NT>([&](auto nc) {
            constexpr int nt = decltype(nc)::value;
            init_acc(acc[nt & 1], bias + 256 + 32 * nt, h);
            auto f1 = [&](int s) { return W1v[(size_t)(s * NT + nt) * 64]; };
            // tile 0 also finishes layer 0's last tile (its words are k-steps 14, 15 of this product) within the first 10 k-steps
            if constexpr (nt == 0) phase1r<HS, 2, 10, 10, true, false>(acc[0], h0p, f1, [&](auto slc) { epi(slc, acc[1], h0p, 7, H0t); });
            else phase1r<HS, 2, HS, 10, true, false>(acc[nt & 1], h0p, f1, [&](auto slc) { epi(slc, acc[(nt & 1) ^ 1], h1p, nt - 1, H1t); });
        });
        f32x16 y;
        {
            // layer 1's tile 7 sits in acc[1]; layer 2 on two fresh partial accumulators (even / odd k-steps), tile 7's epilogue in the shadow of k-steps 0..9
            f32x16 y0, y1;
            auto f2 = [&](int s, int j) { return W2v[(size_t)(2 * s + j) * 64]; };
            bf16x8 ring[3][2];
            static_for<2>([&](auto sc) { constexpr int s = decltype(sc)::value; ring[s][0] = f2(s, 0); ring[s][1] = f2(s, 1); });
            static_for<HS / 2>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 2 < HS / 2) { ring[(s + 2) % 3][0] = f2(s + 2, 0); ring[(s + 2) % 3][1] = f2(s + 2, 1); }
                if constexpr (s < 5) static_for<2>([&](auto jc) { epi(std::integral_constant<int, 2 * s + decltype(jc)::value>{}, acc[1], h1p, 7, H1t); });
                if constexpr (s == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][0], frag_of(h1p), zero, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][1], frag_of(h1p + 4), zero, 0, 0, 0);
                } else {
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % 3][0], frag_of(h1p + 4 * (2 * s)), y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % 3][1], frag_of(h1p + 4 * (2 * s + 1)), y1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // the low plane of W2 (wave_tile.h: behind the high plane in the packed image): same k-step order as k_rr_fwd, whose results this
            // kernel reproduces bit for bit; its fragments come from global memory here (16 KB, cache-resident; this kernel's LDS is full)
            {
                uint32_t zlo = 0;
                asm volatile("" : "+v"(zlo));
                const bf16x8 *W2q = reinterpret_cast<const bf16x8 *>(W2f + kW2F) + lane + zlo;
                static_for<HS / 2>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s) * 64], frag_of(h1p + 4 * (2 * s)), y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s + 1) * 64], frag_of(h1p + 4 * (2 * s + 1)), y1, 0, 0, 0);
                });
            }
#pragma unroll
            for (int i = 0; i < 16; i++) y[i] = y0[i] + y1[i];
        }
        // ---- the K SDFs of the sample, their minimum and its index (lowest among equals): 16 outputs per lane half, one shuffle across the halves
        float best = INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int nn = 8 * (i >> 2) + 4 * h + (i & 3);
            y[i] += bias[512 + nn];
            if (nn < d_out && y[i] < best) { best = y[i]; bi = nn; }
        }
        {
            const float ob = __shfl_xor(best, 32);
            const int oi = __shfl_xor(bi, 32);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (ok) {
            // (opaque lane half: written as 8 q + 4 h the eight 64-bit store offsets are loop invariants the compiler keeps in registers it does
            // not have -- and a scratch reload waits on the vector-memory counter, i.e. for every activation store in front of it)
            int hq = h;
            asm volatile("" : "+v"(hq));
            float *dst = sdf_raw + gp * d_out;
            uint16_t *oh = onehot + gp * 32;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n0 = 8 * q + 4 * hq;
                if ((d_out & 3) == 0) {
                    if (n0 < d_out) *reinterpret_cast<float4 *>(dst + n0) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (n0 + j < d_out) dst[n0 + j] = y[4 * q + j];
                }
                uint2 o;      // bf16 1.0 = 0x3f80
                o.x = (bi == n0 ? 0x3f80u : 0u) | (bi == n0 + 1 ? 0x3f800000u : 0u);
                o.y = (bi == n0 + 2 ? 0x3f80u : 0u) | (bi == n0 + 3 ? 0x3f800000u : 0u);
                *reinterpret_cast<uint2 *>(oh + n0) = o;
            }
            if (h == 0) { sdf[gp] = best; idx[gp] = bi; }
        }
    }
    if (!resident) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward, gradient part
// ux~ = E g~ as the "input layer" of a down pass WITHOUT biases; the epilogues multiply by the derivative factors of the saved activations
__global__ __launch_bounds__(kThreadsW, 2) void k_rr_bwd_grad(const float *__restrict__ x, const float *__restrict__ dydx, const float *__restrict__ g_grad,
                                                               const float *__restrict__ uxh, const int64_t *__restrict__ idx,
                                                               const float *__restrict__ W2tab, const uint16_t *__restrict__ W0f,
                                                               const uint16_t *__restrict__ W1f, const uint16_t *__restrict__ H0t,
                                                               const uint16_t *__restrict__ H1t, const uint16_t *__restrict__ U0t,
                                                               uint16_t *__restrict__ U0bt, uint16_t *__restrict__ A0pt, uint16_t *__restrict__ A1pt,
                                                               uint16_t *__restrict__ U1bt, uint16_t *__restrict__ UXb, float *__restrict__ g_dydx,
                                                               float jac_scale, int64_t n, int64_t ld) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *W1l = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, h = lane >> 5;
    dma_fill(W1f, W1l, kW1F * 2, wave, lane);
    bool resident = false;
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const bf16x8 *W0v = reinterpret_cast<const bf16x8 *>(W0f) + lane;
    const bf16x8 *W1v = reinterpret_cast<const bf16x8 *>(W1l) + lane;
    for (int64_t tile = tile_begin(ntiles) + wave, tile_e = tile_end(ntiles); tile < tile_e; tile += kWaves) {
        const int64_t gp = tile * kRows + row;
        const bool ok = gp < n;
        const int64_t b = ok ? gp : 0;
        const int bi = (int)idx[b];
        const float g[3] = {ok ? g_grad[b * 3] : 0.f, ok ? g_grad[b * 3 + 1] : 0.f, ok ? g_grad[b * 3 + 2] : 0.f};
        uint32_t hin[4 * K0S];
        {
            const EFac E = load_efac(x, dydx, gp, ld, h, jac_scale, ok);
            float v[40];
#pragma unroll
            for (int j = 0; j < 18; j++) v[j] = E.pe[j] * g[j % 3];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v[18 + 2 * i] = E.dy[i][0].x * g[0] + E.dy[i][1].x * g[1] + E.dy[i][2].x * g[2];
                v[19 + 2 * i] = E.dy[i][0].y * g[0] + E.dy[i][1].y * g[1] + E.dy[i][2].y * g[2];
            }
#pragma unroll
            for (int d = 0; d < 3; d++) v[34 + d] = h ? 0.f : g[d];
            v[37] = v[38] = v[39] = 0.f;
#pragma unroll
            for (int j = 0; j < 40; j += 2) hin[j >> 1] = ok ? pack2(v[j], v[j + 1]) : 0u;
            if (ok) {
                uint4 *xp = reinterpret_cast<uint4 *>(UXb + gp * 80 + 40 * h);
#pragma unroll
                for (int i = 0; i < 5; i++) xp[i] = make_uint4(hin[4 * i], hin[4 * i + 1], hin[4 * i + 2], hin[4 * i + 3]);
                // cotangent of dy_dx: rank one, jac_scale * ux[level, c] * g~[d]  (the reference's second-backward cotangent, hashgrid.py:87-101)
                if (g_dydx != nullptr) {      // (NULL: the consumer takes the rank-one form itself -- hsHashLayout::r1_ux -- from uxh and g~)
                    const float4 *up = reinterpret_cast<const float4 *>(uxh + gp * 32 + 16 * h);
                    float u[16];
#pragma unroll
                    for (int i = 0; i < 4; i++) { const float4 t = up[i]; u[4 * i] = t.x; u[4 * i + 1] = t.y; u[4 * i + 2] = t.z; u[4 * i + 3] = t.w; }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        float2 *dp = reinterpret_cast<float2 *>(g_dydx + ((size_t)(8 * h + i) * ld + gp) * 6);
#pragma unroll
                        for (int d = 0; d < 3; d++) dp[d] = make_float2(jac_scale * u[2 * i] * g[d], jac_scale * u[2 * i + 1] * g[d]);
                    }
                }
            }
        }
        uint32_t u0p[64], wa[8], wb[8];
        f32x16 acc[2];
        TilePair hw[2], uw[2];          // saved activations of the tile whose epilogue runs next / the one after (requested a phase ahead)
        // layer 0, finished tile nd: acc = v0~ ; u0~ = v0~ s0 (next product's input, and TP), a0' = v0~ u0 s0' (TP)
        auto epi0 = [&](auto slc, const f32x16 &src, int nd) {
            constexpr int sl = decltype(slc)::value;
            if constexpr (sl < 8) {
                const uint32_t hwd = tile_word(hw[nd & 1], sl), uwd = tile_word(uw[nd & 1], sl);
                const float sa = sig_of_h(lo_bf(hwd)), sb = sig_of_h(hi_bf(hwd));
                const float va = src[2 * sl], vb = src[2 * sl + 1];
                u0p[8 * nd + sl] = anchor(pack2(va * sa, vb * sb));
                wa[sl] = anchor(pack2(va * lo_bf(uwd) * (100.f * sa * (1.f - sa)), vb * hi_bf(uwd) * (100.f * sb * (1.f - sb))));
            } else if constexpr (sl < 10) {
                tp_store<HS_NT_BG_U>(U0bt, tile, 2 * nd + (sl - 8), lane, u0p + 8 * nd + 4 * (sl - 8), ok);
            } else {
                tp_store<HS_NT_BG_A>(A0pt, tile, 2 * nd + (sl - 10), lane, wa + 4 * (sl - 10), ok);
            }
        };
        {
            bf16x8 w0[3][K0S];
            uint32_t zoff = 0;
            asm volatile("" : "+v"(zoff));
            const bf16x8 *W0q = W0v + zoff;
            static_for<K0S>([&](auto sc) { constexpr int s = decltype(sc)::value; w0[0][s] = W0q[(size_t)(s * NT + 0) * 64]; w0[1][s] = W0q[(size_t)(s * NT + 1) * 64]; });
            static_for<NT>([&](auto nc) {
                constexpr int nt = decltype(nc)::value;
                if constexpr (nt + 2 < NT)
                    static_for<K0S>([&](auto sc) { constexpr int s = decltype(sc)::value; w0[(nt + 2) % 3][s] = W0q[(size_t)(s * NT + nt + 2) * 64]; });
                // tile nt's saved activations are consumed by its epilogue in phase nt + 1 (buffer nt & 1; phase nt's epilogue reads the other one)
                { const int64_t tl = here(tile); hw[nt & 1] = tp_load_tile(H0t, tl, nt, lane); uw[nt & 1] = tp_load_tile<HS_NT_LOAD_LAST>(U0t, tl, nt, lane); }
                auto f0 = [&](int s) { return w0[nt % 3][s]; };
                if constexpr (nt == 0) phase1r<K0S, 0, K0S, 12, false, true>(acc[0], hin, f0, [](auto) {});
                else phase1r<K0S, 0, K0S, 12, true, true>(acc[nt & 1], hin, f0, [&](auto slc) { epi0(slc, acc[(nt & 1) ^ 1], nt - 1); });
            });
        }
        if (!resident) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            resident = true;
        }
        // layer 1, finished tile nd: acc = v1~ ; u1~ = v1~ s1 (TP), a1' = v1~ u1 s1' (TP), u1 = W2[k*] gathered
        // u1 of a tile (eight neurons of W2's arg-min row per lane) is requested a phase ahead beside the saved activations: a load that is
        // issued and consumed inside an epilogue slice makes the slice wait for EVERY vector-memory operation before it -- the counter retires
        // in order -- including the stores of the slices just before (measured: 70 full drains per tile in this kernel)
        TilePair h1w[2];
        float4 u1w[2][4];
        const float *u1row = W2tab + (size_t)bi * 256 + 4 * h;
        auto epi1 = [&](auto slc, const f32x16 &src, int nd) {
            constexpr int sl = decltype(slc)::value;
            if constexpr (sl < 8) {
                const uint32_t hwd = tile_word(h1w[nd & 1], sl);
                const float sa = sig_of_h(lo_bf(hwd)), sb = sig_of_h(hi_bf(hwd));
                const float4 uq = u1w[nd & 1][sl >> 1];        // neurons 32 nd + 8 (sl >> 1) + 4 h + 0..3
                const float2 u1 = (sl & 1) ? make_float2(uq.z, uq.w) : make_float2(uq.x, uq.y);
                const float va = src[2 * sl], vb = src[2 * sl + 1];
                wb[sl] = anchor(pack2(va * sa, vb * sb));
                wa[sl] = anchor(pack2(va * u1.x * (100.f * sa * (1.f - sa)), vb * u1.y * (100.f * sb * (1.f - sb))));
            } else if constexpr (sl < 10) {
                tp_store<HS_NT_BG_U>(U1bt, tile, 2 * nd + (sl - 8), lane, wb + 4 * (sl - 8), ok);
            } else {
                tp_store<HS_NT_BG_A>(A1pt, tile, 2 * nd + (sl - 10), lane, wa + 4 * (sl - 10), ok);
            }
        };
        static_for<NT>([&](auto nc) {
            constexpr int nt = decltype(nc)::value;
            auto f1 = [&](int s) { return W1v[(size_t)(s * NT + nt) * 64]; };
            h1w[nt & 1] = tp_load_tile(H1t, here(tile), nt, lane);
            {
                const float *ur = u1row + here(0) + 32 * nt;
#pragma unroll
                for (int q = 0; q < 4; q++) u1w[nt & 1][q] = *reinterpret_cast<const float4 *>(ur + 8 * q);
            }
            if constexpr (nt == 0) phase1r<HS, HS_RR_AH, 10, 12, true, true>(acc[0], u0p, f1, [&](auto slc) { epi0(slc, acc[1], 7); });
            else phase1r<HS, HS_RR_AH, HS, 12, true, true>(acc[nt & 1], u0p, f1, [&](auto slc) { epi1(slc, acc[(nt & 1) ^ 1], nt - 1); });
        });
        static_for<12>([&](auto slc) { epi1(slc, acc[1], 7); });
    }
    if (!resident) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// ================================================================================================================ "up" kernels
// W0^T product: 16 k-steps x 3 slot tiles, fragments from L2 (48 KB image), AH k-steps ahead of their MFMAs; `side(s)` runs in the shadow of k-step s
template <int AH, class SideFn>
__device__ __forceinline__ void up_input_product(const bf16x8 *__restrict__ W0Tv, const uint32_t *vp, f32x16 &o0, f32x16 &o1, f32x16 &o2, SideFn side) {
    bf16x8 ring[AH + 1][XS];
    static_for<AH>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        static_for<XS>([&](auto tc) { constexpr int t = decltype(tc)::value; ring[s][t] = W0Tv[(size_t)(s * XS + t) * 64]; });
    });
    static_for<HS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AH < HS)
            static_for<XS>([&](auto tc) { constexpr int t = decltype(tc)::value; ring[(s + AH) % (AH + 1)][t] = W0Tv[(size_t)((s + AH) * XS + t) * 64]; });
        side(sc);
        const bf16x8 bfrag = frag_of(vp + 4 * s);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (s == 0) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][0], bfrag, zero, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][1], bfrag, zero, 0, 0, 0);
            o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][2], bfrag, zero, 0, 0, 0);
        } else {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AH + 1)][0], bfrag, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AH + 1)][1], bfrag, o1, 0, 0, 0);
            o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AH + 1)][2], bfrag, o2, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}
#ifndef HS_RR_UP_AH
#define HS_RR_UP_AH 3
#endif
constexpr int kUpAhead = HS_RR_UP_AH;

// slot sigma (0..47) of this lane half sits in accumulator (sigma >> 4), register (sigma & 15)
#define HS_SLOT(o0, o1, o2, sigma) ((sigma) < 16 ? o0[(sigma) & 15] : (sigma) < 32 ? o1[(sigma) & 15] : o2[(sigma) & 15])

// ---------------------------------------------------------------------------------------------------------------- forward, gradient chain
__global__ __launch_bounds__(kThreadsW, 2) void k_rr_fwd_grad(const float *__restrict__ x, const float *__restrict__ dydx, const int64_t *__restrict__ idx,
                                                               const float *__restrict__ W2tab, const uint16_t *__restrict__ W1Tf,
                                                               const uint16_t *__restrict__ W0Tf, const uint16_t *__restrict__ H0t,
                                                               const uint16_t *__restrict__ H1t, uint16_t *__restrict__ U0t, uint16_t *__restrict__ V1t,
                                                               uint16_t *__restrict__ V0t, float *__restrict__ grad, float *__restrict__ uxh,
                                                               float jac_scale, int64_t n, int64_t ld) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *W1l = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, h = lane >> 5;
    dma_fill(W1Tf, W1l, kW1F * 2, wave, lane);
    bool resident = false;
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const bf16x8 *W1v = reinterpret_cast<const bf16x8 *>(W1l) + lane;
    const bf16x8 *W0Tv = reinterpret_cast<const bf16x8 *>(W0Tf) + lane;
    for (int64_t tile = tile_begin(ntiles) + wave, tile_e = tile_end(ntiles); tile < tile_e; tile += kWaves) {
        const int64_t gp = tile * kRows + row;
        const bool ok = gp < n;
        const int bi = (int)idx[ok ? gp : 0];
        // ---- v1 = W2[k*] * s1 in B-fragment order (k-step s: neurons 16 s + 4 h + 0..3 and 16 s + 8 + 4 h + 0..3); eight H1 pieces in flight
        uint32_t vin[64];
        {
            const float *wrow = W2tab + (size_t)bi * 256 + 4 * h;
            static_for<2>([&](auto hc) {
                constexpr int half = decltype(hc)::value;
                uint4 hwv[8];
#pragma unroll
                for (int i = 0; i < 8; i++) hwv[i] = tp_load(H1t, here(tile), 8 * half + i, lane);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int s = 8 * half + i;
                    const uint4 hw = hwv[i];
                    const float4 ua = *reinterpret_cast<const float4 *>(wrow + 16 * s), ub = *reinterpret_cast<const float4 *>(wrow + 16 * s + 8);
                    vin[4 * s] = pack2(ua.x * sig_of_h(lo_bf(hw.x)), ua.y * sig_of_h(hi_bf(hw.x)));
                    vin[4 * s + 1] = pack2(ua.z * sig_of_h(lo_bf(hw.y)), ua.w * sig_of_h(hi_bf(hw.y)));
                    vin[4 * s + 2] = pack2(ub.x * sig_of_h(lo_bf(hw.z)), ub.y * sig_of_h(hi_bf(hw.z)));
                    vin[4 * s + 3] = pack2(ub.z * sig_of_h(lo_bf(hw.w)), ub.w * sig_of_h(hi_bf(hw.w)));
                    tp_store<HS_NT_FWD>(V1t, tile, s, lane, vin + 4 * s, ok);
                }
            });
        }
        if (!resident) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            resident = true;
        }
        uint32_t v0p[64], uw8[8];
        f32x16 acc[2];
        TilePair hw[2];
        // finished tile nd of u0 = W1^T v1: u0 leaves as TP, v0 = u0 s0 is the next product's input (and TP, for the weight gradient)
        auto epi = [&](auto slc, const f32x16 &src, int nd) {
            constexpr int sl = decltype(slc)::value;
            if constexpr (sl < 8) {
                const uint32_t hwd = tile_word(hw[nd & 1], sl);
                const float ua = src[2 * sl], ub = src[2 * sl + 1];
                uw8[sl] = anchor(pack2(ua, ub));
                v0p[8 * nd + sl] = anchor(pack2(ua * sig_of_h(lo_bf(hwd)), ub * sig_of_h(hi_bf(hwd))));
            } else if constexpr (sl < 10) {
                tp_store<HS_NT_FWD>(U0t, tile, 2 * nd + (sl - 8), lane, uw8 + 4 * (sl - 8), ok);
            } else {
                tp_store<HS_NT_FWD>(V0t, tile, 2 * nd + (sl - 10), lane, v0p + 8 * nd + 4 * (sl - 10), ok);
            }
        };
        static_for<NT>([&](auto nc) {
            constexpr int nt = decltype(nc)::value;
            auto f1 = [&](int s) { return W1v[(size_t)(s * NT + nt) * 64]; };
            hw[nt & 1] = tp_load_tile(H0t, here(tile), nt, lane);        // consumed by tile nt's epilogue, in phase nt + 1
            if constexpr (nt == 0) phase1r<HS, HS_RR_AH, HS, 12, false, true>(acc[0], vin, f1, [](auto) {});
            else phase1r<HS, HS_RR_AH, HS, 12, true, true>(acc[nt & 1], vin, f1, [&](auto slc) { epi(slc, acc[(nt & 1) ^ 1], nt - 1); });
        });
        // ---- ux = W0^T v0 (tile 7's epilogue -- k-steps 14, 15 of this product -- in the shadow of k-steps 0..11)
        f32x16 &o0 = acc[0];
        f32x16 o1, o2;
        uint32_t zw = 0;
        asm volatile("" : "+v"(zw));         // opaque zero: the (tile-invariant) fragment loads stay inside the tile loop, at this point
        up_input_product<kUpAhead>(W0Tv + zw, v0p, o0, o1, o2, [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s < 12) epi(sc, acc[1], 7);
        });
        // ---- d min / dx = E^T ux over this half's 37 columns, the other half's share by one shuffle; the hash columns of ux are kept
        //      for the backward pass (cotangent of dy_dx).  (Opaque offset: the loads must not be hoisted above the products.)
        int64_t gpo = gp;
        asm volatile("" : "+v"(gpo));
        const EFac E = load_efac(x, dydx, gpo, ld, h, jac_scale, ok);
        float gd[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 18; j++) gd[j % 3] += E.pe[j] * HS_SLOT(o0, o1, o2, j);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int d = 0; d < 3; d++) gd[d] += E.dy[i][d].x * HS_SLOT(o0, o1, o2, 18 + 2 * i) + E.dy[i][d].y * HS_SLOT(o0, o1, o2, 19 + 2 * i);
        if (h == 0) { gd[0] += HS_SLOT(o0, o1, o2, 34); gd[1] += HS_SLOT(o0, o1, o2, 35); gd[2] += HS_SLOT(o0, o1, o2, 36); }
#pragma unroll
        for (int d = 0; d < 3; d++) gd[d] += __shfl_xor(gd[d], 32);
        if (ok) {
            if (h == 0) { grad[gp * 3] = gd[0]; grad[gp * 3 + 1] = gd[1]; grad[gp * 3 + 2] = gd[2]; }
            float4 *up = reinterpret_cast<float4 *>(uxh + gp * 32 + 16 * h);
#pragma unroll
            for (int i = 0; i < 4; i++)
                up[i] = make_float4(HS_SLOT(o0, o1, o2, 18 + 4 * i), HS_SLOT(o0, o1, o2, 19 + 4 * i), HS_SLOT(o0, o1, o2, 20 + 4 * i), HS_SLOT(o0, o1, o2, 21 + 4 * i));
        }
    }
    if (!resident) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward, value part
// PRIME: the gradient part ran (a0', a1' exist); without it (no cotangent on d min / dx) they are zero
// WIDE (33..64 objects): y~ arrives as two planes [2][n][32]; the second plane's product h1~ += W2b^T y~[32..] takes the fragments of W2Tf_b (the W2^T image of
// rows 32..63, packed like W2Tf) from memory -- 16 KB, cache-resident; the LDS holds W1^T and the first image
template <bool PRIME, bool WIDE = false>
__global__ __launch_bounds__(kThreadsW, 2) void k_rr_bwd_value(const uint16_t *__restrict__ gy, const uint16_t *__restrict__ W2Tf,
                                                                const uint16_t *__restrict__ W1Tf, const uint16_t *__restrict__ W0Tf,
                                                                const uint16_t *__restrict__ H0t, const uint16_t *__restrict__ H1t,
                                                                const uint16_t *__restrict__ A0pt, const uint16_t *__restrict__ A1pt,
                                                                uint16_t *__restrict__ A0t, uint16_t *__restrict__ A1t, float *__restrict__ g_feat,
                                                                int64_t n, int64_t ld, const uint16_t *__restrict__ W2Tf_b = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *W1l = lds;
    uint16_t *W2l = lds + kW1F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, h = lane >> 5;
    dma_fill(W1Tf, W1l, kW1F * 2, wave, lane);
    dma_fill(W2Tf, W2l, kW2TF * 2, wave, lane);
    bool resident = false;
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const bf16x8 *W1v = reinterpret_cast<const bf16x8 *>(W1l) + lane;
    const bf16x8 *W2v = reinterpret_cast<const bf16x8 *>(W2l) + lane;
    const bf16x8 *W0Tv = reinterpret_cast<const bf16x8 *>(W0Tf) + lane;
    for (int64_t tile = tile_begin(ntiles) + wave, tile_e = tile_end(ntiles); tile < tile_e; tile += kWaves) {
        const int64_t gp = tile * kRows + row;
        const bool ok = gp < n;
        // B fragments of the output cotangent: k-step s = objects 16 s + 8 h + 0..7 of this sample
        uint32_t gin[WIDE ? 16 : 8];
        {
            const uint4 a = ok ? *reinterpret_cast<const uint4 *>(gy + gp * 32 + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
            const uint4 b = ok ? *reinterpret_cast<const uint4 *>(gy + gp * 32 + 16 + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
            gin[0] = a.x; gin[1] = a.y; gin[2] = a.z; gin[3] = a.w; gin[4] = b.x; gin[5] = b.y; gin[6] = b.z; gin[7] = b.w;
            if constexpr (WIDE) {
                const uint16_t *gy2 = gy + (size_t)n * 32;          // plane 1: objects 32..63
                const uint4 c = ok ? *reinterpret_cast<const uint4 *>(gy2 + gp * 32 + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
                const uint4 d = ok ? *reinterpret_cast<const uint4 *>(gy2 + gp * 32 + 16 + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
                gin[8] = c.x; gin[9] = c.y; gin[10] = c.z; gin[11] = c.w; gin[12] = d.x; gin[13] = d.y; gin[14] = d.z; gin[15] = d.w;
            }
        }
        uint32_t a1p[64], a0p[64];
        f32x16 acc[2];
        TilePair hw[2], pw[2];
        auto load = [&](const uint16_t *H, const uint16_t *P, int nd) {
            const int64_t tl = here(tile);
            hw[nd & 1] = tp_load_tile(H, tl, nd, lane);
            if constexpr (PRIME) pw[nd & 1] = tp_load_tile<HS_NT_LOAD_LAST>(P, tl, nd, lane);
        };
        // a~ = a' + h~ s for the finished tile nd: next product's input and TP (weight gradient)
        auto epi = [&](auto slc, const f32x16 &src, uint32_t *ap, uint16_t *T, int nd) {
            constexpr int sl = decltype(slc)::value;
            if constexpr (sl < 8) {
                const uint32_t hwd = tile_word(hw[nd & 1], sl);
                float va = src[2 * sl] * sig_of_h(lo_bf(hwd)), vb = src[2 * sl + 1] * sig_of_h(hi_bf(hwd));
                if constexpr (PRIME) {
                    const uint32_t pwd = tile_word(pw[nd & 1], sl);
                    va += lo_bf(pwd);
                    vb += hi_bf(pwd);
                }
                ap[8 * nd + sl] = anchor(pack2(va, vb));
            } else {
                tp_store<HS_NT_BV>(T, tile, 2 * nd + (sl - 8), lane, ap + 8 * nd + 4 * (sl - 8), ok);
            }
        };
        load(H1t, A1pt, 0);
        load(H1t, A1pt, 1);
        if (!resident) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            resident = true;
        }
        // ---- h1~ = W2^T y~ (K = 32: two k-steps per tile), tile by tile, each followed by its epilogue; saved activations two tiles ahead
        static_for<NT>([&](auto nc) {
            constexpr int nt = decltype(nc)::value;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 &cur = acc[nt & 1];
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2v[(size_t)(0 * NT + nt) * 64], frag_of(gin), zero, 0, 0, 0);
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2v[(size_t)(1 * NT + nt) * 64], frag_of(gin + 4), cur, 0, 0, 0);
            if constexpr (WIDE) {
                uint32_t zb = 0;
                asm volatile("" : "+v"(zb));      // opaque zero: the (tile-invariant) fragment loads stay at this point of the tile loop
                const bf16x8 *W2bv = reinterpret_cast<const bf16x8 *>(W2Tf_b) + lane + zb;
                cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2bv[(size_t)(0 * NT + nt) * 64], frag_of(gin + 8), cur, 0, 0, 0);
                cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2bv[(size_t)(1 * NT + nt) * 64], frag_of(gin + 12), cur, 0, 0, 0);
            }
            static_for<10>([&](auto slc) { epi(slc, cur, a1p, A1t, nt); });
            if constexpr (nt + 2 < NT) load(H1t, A1pt, nt + 2);
        });
        // ---- h0~ = W1^T a1~
        static_for<NT>([&](auto nc) {
            constexpr int nt = decltype(nc)::value;
            auto f1 = [&](int s) { return W1v[(size_t)(s * NT + nt) * 64]; };
            load(H0t, A0pt, nt);        // consumed by tile nt's epilogue, in phase nt + 1
            if constexpr (nt == 0) phase1r<HS, HS_RR_AH, HS, 10, false, true>(acc[0], a1p, f1, [](auto) {});
            else phase1r<HS, HS_RR_AH, HS, 10, true, true>(acc[nt & 1], a1p, f1, [&](auto slc) { epi(slc, acc[(nt & 1) ^ 1], a0p, A0t, nt - 1); });
        });
        // ---- xt~ = W0^T a0~; only the hash-feature slots are wanted (x is a constant): g_feat [L, ld, 2], level-major
        f32x16 &o0 = acc[0];
        f32x16 o1, o2;
        uint32_t zw = 0;
        asm volatile("" : "+v"(zw));         // opaque zero: the (tile-invariant) fragment loads stay inside the tile loop, at this point
        up_input_product<kUpAhead>(W0Tv + zw, a0p, o0, o1, o2, [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s < 10) epi(sc, acc[1], a0p, A0t, 7);
        });
        if (ok) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                *reinterpret_cast<float2 *>(g_feat + ((size_t)(8 * h + i) * ld + gp) * 2) = make_float2(HS_SLOT(o0, o1, o2, 18 + 2 * i), HS_SLOT(o0, o1, o2, 19 + 2 * i));
        }
    }
    if (!resident) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------- output cotangent image
// gy [n, 32] bf16 = g_raw (K columns, zero beyond) with the minimum's cotangent added at its index; gb2_part [blocks, 32] = per-block column
// sums in fp32 (the last layer's bias gradient = their sum: no same-address atomics).  Thread = (row, group of eight columns): two 16-byte
// loads and ONE 16-byte store per row piece (the first version wrote two bytes per thread: 21 us for 19 MB), column sums in registers,
// folded over the wave's sixteen rows by shuffles at the end.
template <int CG>       // column groups of eight per row: 4 (K <= 32) or 8 (K <= 64: gy is then TWO planes [2][n][32], objects 0..31 | 32..63)
__device__ __forceinline__ void rr_gy_body(int block, int nblocks, const float *__restrict__ g_raw, const float *__restrict__ g_sdf,
                                           const int64_t *__restrict__ idx, int K, uint16_t *__restrict__ gy, float *__restrict__ gb2_part, int64_t n) {
    constexpr int RP = 256 / CG, RW = 64 / CG;       // rows per pass of the workgroup / of a wave
    __shared__ float part[4][8 * CG];
    const int cg = threadIdx.x & (CG - 1), rl = threadIdx.x / CG, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per = ((n + nblocks - 1) / nblocks + RP - 1) / RP * RP;
    const int64_t r0 = (int64_t)block * per, r1 = r0 + per < n ? r0 + per : n;
    const bool vec = (K & 3) == 0;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint16_t *gyp = gy + (size_t)(cg >> 2) * (size_t)n * 32;        // this column group's plane
#pragma unroll 2
    for (int64_t r = r0 + rl; r < r1; r += RP) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (g_raw) {
            const float *src = g_raw + r * K + 8 * cg;
            if (vec) {
                if (8 * cg < K) { const float4 t = *reinterpret_cast<const float4 *>(src); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
                if (8 * cg + 4 < K) { const float4 t = *reinterpret_cast<const float4 *>(src + 4); v[4] = t.x; v[5] = t.y; v[6] = t.z; v[7] = t.w; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    if (8 * cg + j < K) v[j] = src[j];
            }
        }
        if (g_sdf) {
            const int k = (int)idx[r] - 8 * cg;
            const float g = g_sdf[r];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] += (k == j) ? g : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] += v[j];
        *reinterpret_cast<uint4 *>(gyp + r * 32 + 8 * (cg & 3)) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
    if (gb2_part) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int off = CG; off < 64; off <<= 1) acc[j] += __shfl_xor(acc[j], off);
            if (lane < CG) part[wave][8 * lane + j] = acc[j];
        }
        __syncthreads();
        if (threadIdx.x < 8 * CG)
            gb2_part[(size_t)block * (8 * CG) + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    }
    (void)RW;
}

template <int CG>
__global__ __launch_bounds__(256) void k_rr_gy(const float *__restrict__ g_raw, const float *__restrict__ g_sdf, const int64_t *__restrict__ idx, int K,
                                               uint16_t *__restrict__ gy, float *__restrict__ gb2_part, int64_t n) {
    rr_gy_body<CG>((int)blockIdx.x, (int)gridDim.x, g_raw, g_sdf, idx, K, gy, gb2_part, n);
}

// The two output-cotangent images of a trunk backward pass in ONE launch: workgroups [0, HS_RR_GY_BLOCKS) form gy of the rendered samples (above),
// the rest the [4 Be, KPe] image of the Eikonal points' value+Jacobian rows (split_bwd.h).  Both read only what the loss and the compositing
// backward left; as two launches the second cost its ~7 us whatever it did.
template <int CG>
__global__ __launch_bounds__(256) void k_rr_gy_split(const float *__restrict__ g_raw, const float *__restrict__ g_sdf, const int64_t *__restrict__ idx, int K,
                                                     uint16_t *__restrict__ gy, float *__restrict__ gb2_part, int64_t n,
                                                     const int64_t *__restrict__ idx_e, const float *__restrict__ g_yeik, const float *__restrict__ g_mineik,
                                                     const float *__restrict__ g_theta, int64_t Be, int KPe, __hip_bfloat16 *__restrict__ g_img) {
    if ((int)blockIdx.x < HS_RR_GY_BLOCKS) rr_gy_body<CG>((int)blockIdx.x, HS_RR_GY_BLOCKS, g_raw, g_sdf, idx, K, gy, gb2_part, n);
    else trunk_split_bwd_body((int)blockIdx.x - HS_RR_GY_BLOCKS, (int)gridDim.x - HS_RR_GY_BLOCKS, nullptr, nullptr, idx_e, nullptr, g_yeik, g_mineik, g_theta, Be, 0, K,
                              KPe, g_img);
}

// ================================================================================================================ fused forward
// k_rr_fwd_value + k_rr_fwd_grad in ONE kernel per tile: the activations the gradient pass needs (h0, h1 of the SAME samples) are still in the
// wave's registers when the value pass ends, so they are not read back (102 MB), and v1 = u1 . s1, v0 = u0 . s0 overwrite them in place.  Two
// 256 x 256 matrices (W1, W1^T) cannot both be LDS-resident: the workgroup shares the weight pipeline of appearance2.hip -- seven chunks of
// whole neuron tiles cycling through two 64 KB LDS buffers by LDS-DMA, the eight waves meeting once per chunk -- which also frees the
// registers of the streamed-weight rings, so the value layers run in QUARTER phases (two accumulator chains) like sdf_mlp2.hip.
constexpr int kFBuf = 4 * 16 * 1024;
constexpr int kW2Rows = 16 * 1024, kW2Pitch = 1024 + 16;      // chunk 3: W2 fragments | 32 fp32 rows of W2, padded against bank conflicts
#ifndef HS_FA
#define HS_FA 2
#endif
#ifndef HS_COUNTED
#define HS_COUNTED 1
#endif
constexpr bool kCountedWait = HS_COUNTED;
constexpr int kFA = HS_FA;      // k-steps a weight fragment is read from LDS ahead of its MFMAs
// the 1 KB pieces (tile nt0 + ntl, k-step s) of a [k-step][tile] fragment image -> LDS [ntl][s], dealt round-robin to the eight waves
__device__ __forceinline__ void dma_tiles(const uint16_t *__restrict__ img, char *dst, int ks, int ntot, int nt0, int ntn, int wave, int lane) {
    uint32_t lo = (uint32_t)lane * 16u;       // opaque: the request addresses are loop invariants the compiler would keep (and spill)
    asm volatile("" : "+v"(lo));
    const char *src = reinterpret_cast<const char *>(img);
    for (int p = wave; p < ntn * ks; p += kWaves) {
        const int ntl = p / ks, s_ = p - ntl * ks;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)(s_ * ntot + nt0 + ntl) * 1024 + lo),
                                         (__attribute__((address_space(3))) void *)(dst + (size_t)p * 1024), 16, 0, 0);
    }
}

// WIDE (33..64 objects): the last layer's second 32-row tile after the first, as in k_sdf_mlp2<true> -- its fragments (W2f_b: [high | low] planes of rows
// 32..63) and biases (bias_b: a pack's bias block whose b2 slots hold rows 32..63) from memory (the LDS is full); the arg-min runs over both tiles (strict <:
// equal minima keep the lower index); sdf_raw is [n, d_out], the one-hot image two planes [2][n][32]; the arg-min row of W2 (W2tab: 64 rows) is read from
// memory in two halves of eight k-steps, requested BEFORE the tile's output stores (the vector-memory counter retires in order).
template <bool WIDE>
__global__ __launch_bounds__(kThreadsW, 2) void k_rr_fwd(const float *__restrict__ x, const float *__restrict__ feat, const float *__restrict__ dydx,
                                                          const uint16_t *__restrict__ W0f, const uint16_t *__restrict__ W1f, const uint16_t *__restrict__ W2f,
                                                          const float *__restrict__ biasg, const float *__restrict__ W2tab,
                                                          const uint16_t *__restrict__ W1Tf, const uint16_t *__restrict__ W0Tf, int d_out,
                                                          uint16_t *__restrict__ H0t, uint16_t *__restrict__ H1t, uint16_t *__restrict__ Xp,
                                                          float *__restrict__ sdf_raw, float *__restrict__ sdf, int64_t *__restrict__ idx,
                                                          uint16_t *__restrict__ onehot, uint16_t *__restrict__ U0t, uint16_t *__restrict__ V1t,
                                                          uint16_t *__restrict__ V0t, float *__restrict__ grad, float *__restrict__ uxh, float jac_scale,
                                                          int64_t n, int64_t ld, const uint16_t *__restrict__ W2f_b = nullptr, const float *__restrict__ bias2 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char ldsf[];
    float *bias = reinterpret_cast<float *>(ldsf + 2 * kFBuf);
    char *W2lol = ldsf + 2 * kFBuf + kBias * sizeof(float);      // resident: the low plane of W2's fragments (16 KB, wave_tile.h), behind the bias block
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), row = lane & 31, h = lane >> 5;
    int par = 0;
    dma_fill(W2f + kW2F, W2lol, kW2F * 2, wave, lane);             // (lands before the first chunk barrier: that one waits for vmcnt(0))
    // chunk J of a super-tile: 0 W0 (8 tiles x 5 k-steps), 1-2 W1 (4 tiles each), 3 W2, 4-5 W1^T, 6 W0^T (3 slot tiles)
    auto request = [&](int J, int buf) {
        char *dst = ldsf + buf * kFBuf;
        if (J == 0) dma_tiles(W0f, dst, K0S, NT, 0, NT, wave, lane);
        else if (J == 1 || J == 2) dma_tiles(W1f, dst, HS, NT, 4 * (J - 1), 4, wave, lane);
        else if (J == 3) {       // W2 (16 KB) and, behind it, the fp32 rows of W2 the gradient chain starts from (32 rows at a pitch of 1 KB + 16 B)
            dma_tiles(W2f, dst, HS, 1, 0, 1, wave, lane);
            uint32_t lo = (uint32_t)lane * 16u;
            asm volatile("" : "+v"(lo));
            for (int p = wave; p < 32; p += kWaves)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(W2tab) + (size_t)p * 1024 + lo),
                                                 (__attribute__((address_space(3))) void *)(dst + kW2Rows + (size_t)p * kW2Pitch), 16, 0, 0);
        }
        else if (J == 4 || J == 5) dma_tiles(W1Tf, dst, HS, NT, 4 * (J - 4), 4, wave, lane);
        else dma_tiles(W0Tf, dst, HS, XS, 0, XS, wave, lane);
    };
    request(0, 0);
    for (int i = threadIdx.x; i < kBias; i += kThreadsW) bias[i] = biasg[i];
    __syncthreads();        // the bias block (the chunk barriers below are bare barrier instructions)
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const int64_t wt0 = tile_begin(ntiles), wt1 = tile_end(ntiles);
    for (int64_t r0 = wt0; r0 < wt1; r0 += kWaves) {
        const int64_t tile = r0 + wave;
        const bool live = tile < wt1;
        const bool more = r0 + kWaves < wt1;
        const int64_t gp = tile * kRows + row;
        const bool ok = live && gp < n;
        // Chunk J must have landed.  The vector-memory counter retires in issue order and counts this wave's activation stores too: waiting
        // for zero would drain every store of the phase before (a round trip to L2 per chunk); kAfter[J] is a lower bound of the vector-memory
        // instructions a LIVE wave issues between the requests of chunk J and this point, so that many may still be in flight.
        auto chunk_begin = [&](auto jc, auto livec) -> char * {
            constexpr int J = decltype(jc)::value;
            constexpr bool counted = decltype(livec)::value && kCountedWait;
            if constexpr (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (J == 0) { if (r0 == wt0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); }
            else if constexpr (J == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (J == 2 || J == 3 || J == 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (J == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();       // (bare: __syncthreads() brings a workgroup fence, for which the compiler waits for vmcnt(0) anyway)
            if constexpr (J + 1 < 7) request(J + 1, par ^ 1);
            else if (more) request(0, par ^ 1);
            asm volatile("" ::: "memory");      // no store of the coming phase is scheduled ahead of the requests
            char *base = ldsf + par * kFBuf;
            par ^= 1;
            return base;
        };
        if (!live) {
            static_for<7>([&](auto jc) { (void)chunk_begin(jc, std::false_type{}); });
            continue;
        }
        // ---- inputs (k_rr_fwd_value)
        uint32_t hin[4 * K0S];
        {
            float v[40];
            const float x0 = ok ? x[gp * 3] : 0.f, x1 = ok ? x[gp * 3 + 1] : 0.f, x2 = ok ? x[gp * 3 + 2] : 0.f;
            const float xs[3] = {x0, x1, x2};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float f = h ? (float)(8 << k) : (float)(1 << k);
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    float sn, cs;
                    __sincosf(xs[d] * f, &sn, &cs);
                    v[6 * k + d] = sn;
                    v[6 * k + 3 + d] = cs;
                }
            }
            const float4 *fp = reinterpret_cast<const float4 *>(feat + (ok ? gp : 0) * 32 + 16 * h);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 t = ok ? fp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                v[18 + 4 * i] = t.x; v[19 + 4 * i] = t.y; v[20 + 4 * i] = t.z; v[21 + 4 * i] = t.w;
            }
            v[34] = h ? 0.f : x0; v[35] = h ? 0.f : x1; v[36] = h ? 0.f : x2;
            v[37] = v[38] = v[39] = 0.f;
#pragma unroll
            for (int j = 0; j < 40; j += 2) hin[j >> 1] = ok ? pack2(v[j], v[j + 1]) : 0u;
            if (ok) {
                uint4 *xp = reinterpret_cast<uint4 *>(Xp + gp * 80 + 40 * h);
#pragma unroll
                for (int i = 0; i < 5; i++) xp[i] = make_uint4(hin[4 * i], hin[4 * i + 1], hin[4 * i + 2], hin[4 * i + 3]);
            }
        }
        const uint32_t bias_b = lds_base(bias, 16 * h);
        uint32_t h0p[64], h1p[64];          // h0 -> (in place) v0;  h1 -> (in place) v1
        f32x16 acc[2][2];
        // epilogue of a finished quarter qd of a value layer: slices 0..15 Softplus + pack of one register pair, 16..19 its four k-steps leave
        auto epi = [&](auto slc, const f32x16 (&src)[2], uint32_t *hp, auto qdc, uint16_t *T) {
            constexpr int sl = decltype(slc)::value, qd = decltype(qdc)::value;
            if constexpr (sl < 16) {
                constexpr int j = sl >> 3, r = sl & 7, nd = 2 * qd + j;
                hp[8 * nd + r] = anchor(pack2(softplus100(src[j][2 * r]), softplus100(src[j][2 * r + 1])));
            } else {
                constexpr int ks = 4 * qd + (sl - 16);
                tp_store<HS_NT_FWD>(T, here(tile), ks, lane, hp + 4 * ks, ok);
            }
        };
        // ---- layer 0 (chunk 0)
        {
            char *cb = chunk_begin(std::integral_constant<int, 0>{}, std::true_type{});
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                init_acc_b(acc[q & 1][0], relaunder(bias_b), 32 * (2 * q));
                init_acc_b(acc[q & 1][1], relaunder(bias_b), 32 * (2 * q + 1));
                const uint32_t ab = lds_base(cb + 2 * q * K0S * 1024, lane * 16);
                auto f0 = [&](int s_, int j) { return lds_at<bf16x8>(ab, (j * K0S + s_) * 1024); };
                if constexpr (q == 0) phase2<K0S, kFA, K0S, 20, false>(acc[0], hin, f0, [](auto) {});
                else phase2<K0S, kFA, K0S, 20, true>(acc[q & 1], hin, f0, [&](auto slc) { epi(slc, acc[(q & 1) ^ 1], h0p, std::integral_constant<int, q - 1>{}, H0t); });
            });
        }
        // ---- layer 1 (chunks 1, 2); quarter 0 finishes layer 0's quarter 3 within its first 10 k-steps
        {
            char *cb = nullptr;
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q % 2 == 0) cb = chunk_begin(std::integral_constant<int, 1 + q / 2>{}, std::true_type{});
                init_acc_b(acc[q & 1][0], relaunder(bias_b), 256 + 32 * (2 * q));
                init_acc_b(acc[q & 1][1], relaunder(bias_b), 256 + 32 * (2 * q + 1));
                const uint32_t ab = lds_base(cb + 2 * (q % 2) * HS * 1024, lane * 16);
                auto f1 = [&](int s_, int j) { return lds_at<bf16x8>(ab, (j * HS + s_) * 1024); };
                if constexpr (q == 0) phase2<HS, kFA, 10, 20, true>(acc[0], h0p, f1, [&](auto slc) { epi(slc, acc[1], h0p, std::integral_constant<int, 3>{}, H0t); });
                else phase2<HS, kFA, HS, 20, true>(acc[q & 1], h0p, f1, [&](auto slc) { epi(slc, acc[(q & 1) ^ 1], h1p, std::integral_constant<int, q - 1>{}, H1t); });
            });
        }
        // ---- layer 2 (chunk 3): two partial accumulators, layer 1's quarter 3 in the shadow of k-steps 0..9; then the K SDFs, minimum, arg-min
        int bi = 0x7fffffff;
        char *cb3;
        [[maybe_unused]] f32x4 u1g[WIDE ? 8 : 1][2];       // WIDE: eight k-steps of the arg-min row of W2, from memory
        {
            char *cb = cb3 = chunk_begin(std::integral_constant<int, 3>{}, std::true_type{});
            const uint32_t ab = lds_base(cb, lane * 16);
            f32x16 y0, y1, y;
            bf16x8 ring[3][2];
            auto f2 = [&](int s_, int j) { return lds_at<bf16x8>(ab, (2 * s_ + j) * 1024); };
            static_for<2>([&](auto sc) { constexpr int s_ = decltype(sc)::value; ring[s_][0] = f2(s_, 0); ring[s_][1] = f2(s_, 1); });
            static_for<HS / 2>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                if constexpr (s_ + 2 < HS / 2) { ring[(s_ + 2) % 3][0] = f2(s_ + 2, 0); ring[(s_ + 2) % 3][1] = f2(s_ + 2, 1); }
                if constexpr (s_ < 5) static_for<4>([&](auto jc) { epi(std::integral_constant<int, 4 * s_ + decltype(jc)::value>{}, acc[1], h1p, std::integral_constant<int, 3>{}, H1t); });
                if constexpr (s_ == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][0], frag_of(h1p), zero, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][1], frag_of(h1p + 4), zero, 0, 0, 0);
                } else {
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s_ % 3][0], frag_of(h1p + 4 * (2 * s_)), y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s_ % 3][1], frag_of(h1p + 4 * (2 * s_ + 1)), y1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            {   // + W2lo h1 (wave_tile.h: the low plane)
                const uint32_t lb = lds_base(W2lol, lane * 16);
                static_for<HS / 2>([&](auto sc) {
                    constexpr int s_ = decltype(sc)::value;
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_at<bf16x8>(lb, (2 * s_) * 1024), frag_of(h1p + 4 * (2 * s_)), y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_at<bf16x8>(lb, (2 * s_ + 1) * 1024), frag_of(h1p + 4 * (2 * s_ + 1)), y1, 0, 0, 0);
                });
            }
            float best = INFINITY;
            int hq2 = h;        // opaque: sixteen "nn < d_out" lane masks and as many LDS addresses are loop invariants otherwise (kept, and spilled)
            asm volatile("" : "+v"(hq2));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 bq = lds_at<f32x4>(relaunder(bias_b), (uint32_t)(512 + 8 * q) * 4u);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = 4 * q + j, nn = 8 * q + 4 * hq2 + j;
                    y[i] = y0[i] + y1[i] + bq[j];
                    if (nn < d_out && y[i] < best) { best = y[i]; bi = nn; }
                }
            }
            // this lane's sixteen raw outputs of the tile that starts at object `base`
            auto store_raw = [&](int base) {
                const int64_t gq = here(gp);
                int hq = h;         // (opaque, as in k_rr_fwd_value)
                asm volatile("" : "+v"(hq));
                float *dst = sdf_raw + gq * d_out + base;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n0 = 8 * q + 4 * hq;
                    if ((d_out & 3) == 0) {
                        if (base + n0 < d_out) *reinterpret_cast<float4 *>(dst + n0) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (base + n0 + j < d_out) dst[n0 + j] = y[4 * q + j];
                    }
                }
            };
            if constexpr (WIDE) {
                if (ok) store_raw(0);
                // ---- second output tile: objects 32..63 (fragments and biases from memory)
                uint32_t zb = 0;
                asm volatile("" : "+v"(zb));
                const bf16x8 *W2q = reinterpret_cast<const bf16x8 *>(W2f_b) + lane + zb, *W2r = W2q + (size_t)HS * 64;
                static_for<HS / 2>([&](auto sc) {
                    constexpr int s_ = decltype(sc)::value;
                    if constexpr (s_ == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[0], frag_of(h1p), zero, 0, 0, 0);
                        y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[64], frag_of(h1p + 4), zero, 0, 0, 0);
                    } else {
                        y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s_) * 64], frag_of(h1p + 4 * (2 * s_)), y0, 0, 0, 0);
                        y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s_ + 1) * 64], frag_of(h1p + 4 * (2 * s_ + 1)), y1, 0, 0, 0);
                    }
                });
                static_for<HS / 2>([&](auto sc) {
                    constexpr int s_ = decltype(sc)::value;
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2r[(size_t)(2 * s_) * 64], frag_of(h1p + 4 * (2 * s_)), y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2r[(size_t)(2 * s_ + 1) * 64], frag_of(h1p + 4 * (2 * s_ + 1)), y1, 0, 0, 0);
                });
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int nn = 8 * (i >> 2) + 4 * hq2 + (i & 3);
                    y[i] = y0[i] + y1[i] + bias2[512 + nn];
                    if (32 + nn < d_out && y[i] < best) { best = y[i]; bi = 32 + nn; }
                }
            }
            {
                const float ob = __shfl_xor(best, 32);
                const int oi = __shfl_xor(bi, 32);
                if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if constexpr (WIDE) {       // the arg-min row's first half (k-steps 0..7), requested in front of the stores below
                const float *grow = W2tab + (size_t)(ok ? bi : 0) * 256 + 4 * h;
#pragma unroll
                for (int s_ = 0; s_ < 8; s_++) {
                    u1g[s_][0] = *reinterpret_cast<const f32x4 *>(grow + 16 * s_);
                    u1g[s_][1] = *reinterpret_cast<const f32x4 *>(grow + 16 * s_ + 8);
                }
            }
            if (ok) {
                store_raw(WIDE ? 32 : 0);
                const int64_t gq = here(gp);
                int hq = h;
                asm volatile("" : "+v"(hq));
#pragma unroll
                for (int pl = 0; pl < (WIDE ? 2 : 1); pl++) {
                    uint16_t *oh = onehot + (size_t)pl * (size_t)n * 32 + gq * 32;
                    const int bl = bi - 32 * pl;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int n0 = 8 * q + 4 * hq;
                        uint2 o;      // bf16 1.0 = 0x3f80
                        o.x = (bl == n0 ? 0x3f80u : 0u) | (bl == n0 + 1 ? 0x3f800000u : 0u);
                        o.y = (bl == n0 + 2 ? 0x3f80u : 0u) | (bl == n0 + 3 ? 0x3f800000u : 0u);
                        *reinterpret_cast<uint2 *>(oh + n0) = o;
                    }
                }
                if (h == 0) { sdf[gq] = best; idx[gq] = bi; }
            }
        }
        // ---- v1 = W2[k*] . s1 over h1, in place (k-step s: neurons 16 s + 4 h + 0..3 and 16 s + 8 + 4 h + 0..3).  The sample's row of W2 is read from
        //      LDS (it came with chunk 3), not from global memory: the vector-memory counter retires in order, so a global load issued behind
        //      the activation stores of the k-step before is not "there" until those stores have been acknowledged -- sixteen store round
        //      trips per tile in the two-kernel form's k_rr_fwd_grad
        {
            const uint32_t wl = lds_base(cb3 + kW2Rows + (size_t)((ok && !WIDE) ? bi : 0) * kW2Pitch, 16 * h);
            static_for<HS>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                f32x4 ua, ub;
                if constexpr (WIDE) {
                    if constexpr (s_ == 8) {       // second half of the row (one wait behind the first eight k-steps' stores)
                        const float *grow = W2tab + (size_t)(ok ? bi : 0) * 256 + 4 * h + 128;
#pragma unroll
                        for (int t_ = 0; t_ < 8; t_++) {
                            u1g[t_][0] = *reinterpret_cast<const f32x4 *>(grow + 16 * t_);
                            u1g[t_][1] = *reinterpret_cast<const f32x4 *>(grow + 16 * t_ + 8);
                        }
                    }
                    ua = u1g[s_ & 7][0];
                    ub = u1g[s_ & 7][1];
                } else {
                    ua = lds_at<f32x4>(wl, 64 * s_);
                    ub = lds_at<f32x4>(wl, 64 * s_ + 32);
                }
                const uint32_t w0 = h1p[4 * s_], w1 = h1p[4 * s_ + 1], w2 = h1p[4 * s_ + 2], w3 = h1p[4 * s_ + 3];
                h1p[4 * s_] = pack2(ua[0] * sig_of_h(lo_bf(w0)), ua[1] * sig_of_h(hi_bf(w0)));
                h1p[4 * s_ + 1] = pack2(ua[2] * sig_of_h(lo_bf(w1)), ua[3] * sig_of_h(hi_bf(w1)));
                h1p[4 * s_ + 2] = pack2(ub[0] * sig_of_h(lo_bf(w2)), ub[1] * sig_of_h(hi_bf(w2)));
                h1p[4 * s_ + 3] = pack2(ub[2] * sig_of_h(lo_bf(w3)), ub[3] * sig_of_h(hi_bf(w3)));
                tp_store<HS_NT_FWD>(V1t, here(tile), s_, lane, h1p + 4 * s_, ok);
            });
        }
        // ---- u0 = W1^T v1 (chunks 4, 5) in quarter phases; a finished quarter leaves as u0 (TP) and, times s0 of h0, as v0 IN PLACE of h0 (TP)
        // 24 slices of a finished quarter qd, per k-step t of it: four packs (u0 word, v0 word in place of h0's), the u0 store, the v0 store
        uint32_t uw[4];
        auto epu = [&](auto slc, const f32x16 (&src)[2], auto qdc) {
            constexpr int sl = decltype(slc)::value, qd = decltype(qdc)::value, t = sl / 6, w = sl % 6;
            if constexpr (w < 4) {
                constexpr int j = t >> 1, r = 4 * (t & 1) + w, nd = 2 * qd + j;
                const uint32_t hwd = h0p[8 * nd + r];
                const float ua = src[j][2 * r], ub = src[j][2 * r + 1];
                uw[w] = anchor(pack2(ua, ub));
                h0p[8 * nd + r] = anchor(pack2(ua * sig_of_h(lo_bf(hwd)), ub * sig_of_h(hi_bf(hwd))));
            } else if constexpr (w == 4) {
                tp_store<HS_NT_FWD>(U0t, here(tile), 4 * qd + t, lane, uw, ok);
            } else {
                tp_store<HS_NT_FWD>(V0t, here(tile), 4 * qd + t, lane, h0p + 4 * (4 * qd + t), ok);
            }
        };
        {
            char *cb = nullptr;
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q % 2 == 0) cb = chunk_begin(std::integral_constant<int, 4 + q / 2>{}, std::true_type{});
                const uint32_t ab = lds_base(cb + 2 * (q % 2) * HS * 1024, lane * 16);
                auto f1 = [&](int s_, int j) { return lds_at<bf16x8>(ab, (j * HS + s_) * 1024); };
                if constexpr (q == 0) phase2<HS, kFA, HS, 24, false, true>(acc[0], h1p, f1, [](auto) {});
                else phase2<HS, kFA, HS, 24, true, true>(acc[q & 1], h1p, f1, [&](auto slc) { epu(slc, acc[(q & 1) ^ 1], std::integral_constant<int, q - 1>{}); });
            });
        }
        // ---- ux = W0^T v0 (chunk 6: three slot tiles, three accumulator chains); u0's quarter 3 (v0 k-steps 12..15) in the shadow of k-steps 0..11
        f32x16 o0, o1, o2;
        {
            char *cb = chunk_begin(std::integral_constant<int, 6>{}, std::true_type{});
            const uint32_t ab = lds_base(cb, lane * 16);
            bf16x8 ring[3][XS];
            static_for<2>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                static_for<XS>([&](auto tc) { constexpr int t = decltype(tc)::value; ring[s_][t] = lds_at<bf16x8>(ab, (t * HS + s_) * 1024); });
            });
            static_for<HS>([&](auto sc) {
                constexpr int s_ = decltype(sc)::value;
                if constexpr (s_ + 2 < HS)
                    static_for<XS>([&](auto tc) { constexpr int t = decltype(tc)::value; ring[(s_ + 2) % 3][t] = lds_at<bf16x8>(ab, (t * HS + s_ + 2) * 1024); });
                if constexpr (s_ < 12) static_for<2>([&](auto jc) { epu(std::integral_constant<int, 2 * s_ + decltype(jc)::value>{}, acc[1], std::integral_constant<int, 3>{}); });
                const bf16x8 bfrag = frag_of(h0p + 4 * s_);
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (s_ == 0) {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][0], bfrag, zero, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][1], bfrag, zero, 0, 0, 0);
                    o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][2], bfrag, zero, 0, 0, 0);
                } else {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s_ % 3][0], bfrag, o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s_ % 3][1], bfrag, o1, 0, 0, 0);
                    o2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s_ % 3][2], bfrag, o2, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- d min / dx = E^T ux (k_rr_fwd_grad's tail)
        {
            int64_t gpo = ok ? gp : 0;
            asm volatile("" : "+v"(gpo));
            const EFac E = load_efac(x, dydx, gpo, ld, h, jac_scale, ok);
            float gd[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 18; j++) gd[j % 3] += E.pe[j] * HS_SLOT(o0, o1, o2, j);
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int d = 0; d < 3; d++) gd[d] += E.dy[i][d].x * HS_SLOT(o0, o1, o2, 18 + 2 * i) + E.dy[i][d].y * HS_SLOT(o0, o1, o2, 19 + 2 * i);
            if (h == 0) { gd[0] += HS_SLOT(o0, o1, o2, 34); gd[1] += HS_SLOT(o0, o1, o2, 35); gd[2] += HS_SLOT(o0, o1, o2, 36); }
#pragma unroll
            for (int d = 0; d < 3; d++) gd[d] += __shfl_xor(gd[d], 32);
            if (ok) {
                if (h == 0) { grad[gpo * 3] = gd[0]; grad[gpo * 3 + 1] = gd[1]; grad[gpo * 3 + 2] = gd[2]; }
                float4 *up = reinterpret_cast<float4 *>(uxh + gpo * 32 + 16 * h);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    up[i] = make_float4(HS_SLOT(o0, o1, o2, 18 + 4 * i), HS_SLOT(o0, o1, o2, 19 + 4 * i), HS_SLOT(o0, o1, o2, 20 + 4 * i), HS_SLOT(o0, o1, o2, 21 + 4 * i));
            }
        }
    }
}

}  // namespace

extern "C" {

int64_t hs_trunk_rr_pack_bytes(int32_t which) {
    switch (which) {
        case 0: return (int64_t)kW1F * 2;        /* W1^T image */
        case 1: return (int64_t)kW0TF * 2;       /* W0^T image */
        case 2: return (int64_t)kW2TF * 2;       /* W2^T image */
        case 3: return (int64_t)32 * 256 * 4;    /* W2 gather table */
        default: return -1;
    }
}

int hs_trunk_rr_pack(const float *W0, int32_t ld0, const float *W1, const float *W2, int32_t d_out, void *W1Tf, void *W0Tf, void *W2Tf, float *W2tab,
                     void *stream) {
    if (d_out < 1 || d_out > 32 || ld0 < 71) return HS_ERR_ARG;
    if (!W0 || !W1 || !W2 || !W1Tf || !W0Tf || !W2Tf || !W2tab) return HS_ERR_NULL;
    const int slots = kRrPackSlots;
    k_rr_pack<<<(slots + 255) / 256, 256, 0, (hipStream_t)stream>>>(W0, ld0, W1, W2, d_out, (uint16_t *)W1Tf, (uint16_t *)W0Tf, (uint16_t *)W2Tf, W2tab);
    return wt_check_launch();
}

int hs_trunk_pack_all(const float *W0, int32_t ld0, int32_t f_in, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2, int32_t d_out,
                      void *W0f, void *W1f, void *W2f, float *bias, void *W1Tf, void *W0Tf, void *W2Tf, float *W2tab, void *w1t, void *w2t, void *w0t,
                      void *stream) {
    if (d_out < 1 || d_out > 32 || ld0 < 71 || f_in < 1 || f_in > 256 || f_in > ld0) return HS_ERR_ARG;
    if (!W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !W0f || !W1f || !W2f || !bias || !W1Tf || !W0Tf || !W2Tf || !W2tab) return HS_ERR_NULL;
    if ((w1t != nullptr) != (w2t != nullptr) || (w1t != nullptr) != (w0t != nullptr)) return HS_ERR_NULL;
    const int slots = kSdfPackSlots + kRrPackSlots + (w1t ? kTransSlots : 0);
    k_trunk_pack_all<<<(slots + 255) / 256, 256, 0, (hipStream_t)stream>>>(W0, ld0, f_in, b0, W1, b1, W2, b2, d_out, (uint16_t *)W0f, (uint16_t *)W1f, (uint16_t *)W2f,
                                                                         bias, (uint16_t *)W1Tf, (uint16_t *)W0Tf, (uint16_t *)W2Tf, W2tab, (uint16_t *)w1t,
                                                                         (uint16_t *)w2t, (uint16_t *)w0t);
    return wt_check_launch();
}

static int rr_grid(int64_t n) {
    const int64_t ntiles = (n + kRows - 1) / kRows, want = (ntiles + kWaves - 1) / kWaves;
    return (int)(want < 256 ? want : 256);
}

int hs_trunk_rr_gy(const float *g_raw, const float *g_sdf, const int64_t *idx, int32_t K, void *gy, float *gb2_part, int64_t n, void *stream) {
    if (K < 1 || K > 64) return HS_ERR_ARG;
    if (n == 0) return HS_OK;
    if (!gy || (g_sdf && !idx)) return HS_ERR_NULL;
    if (K <= 32) k_rr_gy<4><<<HS_RR_GY_BLOCKS, 256, 0, (hipStream_t)stream>>>(g_raw, g_sdf, idx, K, (uint16_t *)gy, gb2_part, n);
    else k_rr_gy<8><<<HS_RR_GY_BLOCKS, 256, 0, (hipStream_t)stream>>>(g_raw, g_sdf, idx, K, (uint16_t *)gy, gb2_part, n);       /* gy [2][n][32], gb2_part [blocks, 64] */
    return wt_check_launch();
}

int hs_trunk_rr_gy_split(const float *g_raw, const float *g_sdf, const int64_t *idx, int32_t K, void *gy, float *gb2_part, int64_t n, const int64_t *idx_e,
                         const float *g_y_eik, const float *g_min_eik, const float *g_grad_theta, int64_t Be, int32_t KPe, void *g_img, void *stream) {
    if (K < 1 || K > 64 || K > KPe || (KPe & 31) || n < 1 || Be < 1) return HS_ERR_ARG;
    if (!gy || (g_sdf && !idx) || !idx_e || !g_img) return HS_ERR_NULL;
    const int64_t want = (Be * 4 * (KPe / 8) + 255) / 256;
    const int split_blocks = (int)(want < 2048 ? want : 2048);
    if (K <= 32)
        k_rr_gy_split<4><<<HS_RR_GY_BLOCKS + split_blocks, 256, 0, (hipStream_t)stream>>>(g_raw, g_sdf, idx, K, (uint16_t *)gy, gb2_part, n, idx_e, g_y_eik, g_min_eik,
                                                                                         g_grad_theta, Be, KPe, (__hip_bfloat16 *)g_img);
    else
        k_rr_gy_split<8><<<HS_RR_GY_BLOCKS + split_blocks, 256, 0, (hipStream_t)stream>>>(g_raw, g_sdf, idx, K, (uint16_t *)gy, gb2_part, n, idx_e, g_y_eik, g_min_eik,
                                                                                         g_grad_theta, Be, KPe, (__hip_bfloat16 *)g_img);
    return wt_check_launch();
}

int hs_trunk_rr_fwd_value(const float *x, const float *feat, const void *W0f, const void *W1f, const void *W2f, const float *bias, int32_t d_out,
                          void *H0t, void *H1t, void *Xp, float *sdf_raw, float *sdf, int64_t *idx, void *onehot, int64_t n, void *stream) {
    if (d_out < 1 || d_out > 32) return HS_ERR_ARG;
    if (n == 0) return HS_OK;
    if (!x || !feat || !W0f || !W1f || !W2f || !bias || !H0t || !H1t || !Xp || !sdf_raw || !sdf || !idx || !onehot) return HS_ERR_NULL;
    if ((const char *)W2f != (const char *)W1f + (size_t)kW1F * 2) return HS_ERR_ARG;
    const size_t lds = (size_t)(kW1F + kW2F) * sizeof(uint16_t) + kBias * sizeof(float);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_rr_fwd_value, (int)lds);
    k_rr_fwd_value<<<rr_grid(n), kThreadsW, lds, (hipStream_t)stream>>>(x, feat, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias, d_out,
                                                                    (uint16_t *)H0t, (uint16_t *)H1t, (uint16_t *)Xp, sdf_raw, sdf, idx, (uint16_t *)onehot, n);
    return wt_check_launch();
}

int hs_trunk_rr_fwd(const float *x, const float *feat, const float *dydx, const void *W0f, const void *W1f, const void *W2f, const float *bias,
                    const float *W2tab, const void *W1Tf, const void *W0Tf, int32_t d_out, void *H0t, void *H1t, void *Xp, float *sdf_raw, float *sdf,
                    int64_t *idx, void *onehot, void *U0t, void *V1t, void *V0t, float *grad, float *uxh, float jac_scale, int64_t n, int64_t ld,
                    void *stream) {
    if (d_out < 1 || d_out > 32) return HS_ERR_ARG;
    if (n == 0) return HS_OK;
    if (ld == 0) ld = n;
    if (ld < n) return HS_ERR_ARG;
    if (!x || !feat || !dydx || !W0f || !W1f || !W2f || !bias || !W2tab || !W1Tf || !W0Tf || !H0t || !H1t || !Xp || !sdf_raw || !sdf || !idx || !onehot || !U0t ||
        !V1t || !V0t || !grad || !uxh)
        return HS_ERR_NULL;
    const size_t lds = 2 * (size_t)kFBuf + kBias * sizeof(float) + (size_t)kW2F * sizeof(uint16_t);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_rr_fwd<false>, (int)lds);
    k_rr_fwd<false><<<rr_grid(n), kThreadsW, lds, (hipStream_t)stream>>>(x, feat, dydx, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias, W2tab,
                                                                      (const uint16_t *)W1Tf, (const uint16_t *)W0Tf, d_out, (uint16_t *)H0t, (uint16_t *)H1t,
                                                                      (uint16_t *)Xp, sdf_raw, sdf, idx, (uint16_t *)onehot, (uint16_t *)U0t, (uint16_t *)V1t,
                                                                      (uint16_t *)V0t, grad, uxh, jac_scale, n, ld);
    return wt_check_launch();
}

/* 33..64 objects: W0f .. bias = the pack of the last layer's rows 0..31 (hs_trunk_pack_all / hs_sdf_mlp2_pack(log2_domain = 0)), W2f_b / bias_b = W2f / bias of the
 * pack of rows 32..63; W2tab fp32 [64, 256] (rows >= d_out unused); sdf_raw [n, d_out]; onehot two planes [2][n][32] */
int hs_trunk_rr_fwd_wide(const float *x, const float *feat, const float *dydx, const void *W0f, const void *W1f, const void *W2f, const float *bias,
                         const void *W2f_b, const float *bias_b, const float *W2tab, const void *W1Tf, const void *W0Tf, int32_t d_out, void *H0t, void *H1t,
                         void *Xp, float *sdf_raw, float *sdf, int64_t *idx, void *onehot, void *U0t, void *V1t, void *V0t, float *grad, float *uxh,
                         float jac_scale, int64_t n, int64_t ld, void *stream) {
    if (d_out < 33 || d_out > 64) return HS_ERR_ARG;
    if (n == 0) return HS_OK;
    if (ld == 0) ld = n;
    if (ld < n) return HS_ERR_ARG;
    if (!x || !feat || !dydx || !W0f || !W1f || !W2f || !bias || !W2f_b || !bias_b || !W2tab || !W1Tf || !W0Tf || !H0t || !H1t || !Xp || !sdf_raw || !sdf || !idx ||
        !onehot || !U0t || !V1t || !V0t || !grad || !uxh)
        return HS_ERR_NULL;
    const size_t lds = 2 * (size_t)kFBuf + kBias * sizeof(float) + (size_t)kW2F * sizeof(uint16_t);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_rr_fwd<true>, (int)lds);
    k_rr_fwd<true><<<rr_grid(n), kThreadsW, lds, (hipStream_t)stream>>>(x, feat, dydx, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias, W2tab,
                                                                     (const uint16_t *)W1Tf, (const uint16_t *)W0Tf, d_out, (uint16_t *)H0t, (uint16_t *)H1t,
                                                                     (uint16_t *)Xp, sdf_raw, sdf, idx, (uint16_t *)onehot, (uint16_t *)U0t, (uint16_t *)V1t,
                                                                     (uint16_t *)V0t, grad, uxh, jac_scale, n, ld, (const uint16_t *)W2f_b, bias_b);
    return wt_check_launch();
}

int hs_trunk_rr_fwd_grad(const float *x, const float *dydx, const int64_t *idx, const float *W2tab, const void *W1Tf, const void *W0Tf, const void *H0t,
                         const void *H1t, void *U0t, void *V1t, void *V0t, float *grad, float *uxh, float jac_scale, int64_t n, int64_t ld, void *stream) {
    if (n == 0) return HS_OK;
    if (ld == 0) ld = n;
    if (ld < n) return HS_ERR_ARG;
    if (!x || !dydx || !idx || !W2tab || !W1Tf || !W0Tf || !H0t || !H1t || !U0t || !V1t || !V0t || !grad || !uxh) return HS_ERR_NULL;
    const size_t lds = (size_t)kW1F * sizeof(uint16_t);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_rr_fwd_grad, (int)lds);
    k_rr_fwd_grad<<<rr_grid(n), kThreadsW, lds, (hipStream_t)stream>>>(x, dydx, idx, W2tab, (const uint16_t *)W1Tf, (const uint16_t *)W0Tf, (const uint16_t *)H0t,
                                                                   (const uint16_t *)H1t, (uint16_t *)U0t, (uint16_t *)V1t, (uint16_t *)V0t, grad, uxh, jac_scale, n, ld);
    return wt_check_launch();
}

int hs_trunk_rr_bwd_grad(const float *x, const float *dydx, const float *g_grad, const float *uxh, const int64_t *idx, const float *W2tab, const void *W0f,
                         const void *W1f, const void *H0t, const void *H1t, const void *U0t, void *U0bt, void *A0pt, void *A1pt, void *U1bt, void *UXb,
                         float *g_dydx, float jac_scale, int64_t n, int64_t ld, void *stream) {
    if (n == 0) return HS_OK;
    if (ld == 0) ld = n;
    if (ld < n) return HS_ERR_ARG;
    if (!x || !dydx || !g_grad || !uxh || !idx || !W2tab || !W0f || !W1f || !H0t || !H1t || !U0t || !U0bt || !A0pt || !A1pt || !U1bt || !UXb)       /* g_dydx may be NULL */
        return HS_ERR_NULL;
    const size_t lds = (size_t)kW1F * sizeof(uint16_t);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_rr_bwd_grad, (int)lds);
    k_rr_bwd_grad<<<rr_grid(n), kThreadsW, lds, (hipStream_t)stream>>>(x, dydx, g_grad, uxh, idx, W2tab, (const uint16_t *)W0f, (const uint16_t *)W1f,
                                                                   (const uint16_t *)H0t, (const uint16_t *)H1t, (const uint16_t *)U0t, (uint16_t *)U0bt,
                                                                   (uint16_t *)A0pt, (uint16_t *)A1pt, (uint16_t *)U1bt, (uint16_t *)UXb, g_dydx, jac_scale, n, ld);
    return wt_check_launch();
}

int hs_trunk_rr_bwd_value(const void *gy, const void *W2Tf, const void *W1Tf, const void *W0Tf, const void *H0t, const void *H1t, const void *A0pt,
                          const void *A1pt, void *A0t, void *A1t, float *g_feat, int64_t n, int64_t ld, void *stream) {
    if (n == 0) return HS_OK;
    if (ld == 0) ld = n;
    if (ld < n) return HS_ERR_ARG;
    if (!gy || !W2Tf || !W1Tf || !W0Tf || !H0t || !H1t || !A0t || !A1t || !g_feat || (!A0pt) != (!A1pt)) return HS_ERR_NULL;
    const size_t lds = (size_t)(kW1F + kW2TF) * sizeof(uint16_t);
    static hsLdsAttrOnce attr_a, attr_b;
    attr_a.set((const void *)k_rr_bwd_value<true>, (int)lds);
    attr_b.set((const void *)k_rr_bwd_value<false>, (int)lds);
    if (A0pt)
        k_rr_bwd_value<true><<<rr_grid(n), kThreadsW, lds, (hipStream_t)stream>>>((const uint16_t *)gy, (const uint16_t *)W2Tf, (const uint16_t *)W1Tf, (const uint16_t *)W0Tf,
                                                                              (const uint16_t *)H0t, (const uint16_t *)H1t, (const uint16_t *)A0pt, (const uint16_t *)A1pt,
                                                                              (uint16_t *)A0t, (uint16_t *)A1t, g_feat, n, ld);
    else
        k_rr_bwd_value<false><<<rr_grid(n), kThreadsW, lds, (hipStream_t)stream>>>((const uint16_t *)gy, (const uint16_t *)W2Tf, (const uint16_t *)W1Tf, (const uint16_t *)W0Tf,
                                                                               (const uint16_t *)H0t, (const uint16_t *)H1t, nullptr, nullptr, (uint16_t *)A0t, (uint16_t *)A1t,
                                                                               g_feat, n, ld);
    return wt_check_launch();
}

/* 33..64 objects: gy = two planes [2][n][32] (hs_trunk_rr_gy with K > 32), W2Tf / W2Tf_b = the W2^T images of rows 0..31 / 32..63 (hs_trunk_rr_pack of each half) */
int hs_trunk_rr_bwd_value_wide(const void *gy, const void *W2Tf, const void *W2Tf_b, const void *W1Tf, const void *W0Tf, const void *H0t, const void *H1t,
                               const void *A0pt, const void *A1pt, void *A0t, void *A1t, float *g_feat, int64_t n, int64_t ld, void *stream) {
    if (n == 0) return HS_OK;
    if (ld == 0) ld = n;
    if (ld < n) return HS_ERR_ARG;
    if (!gy || !W2Tf || !W2Tf_b || !W1Tf || !W0Tf || !H0t || !H1t || !A0t || !A1t || !g_feat || (!A0pt) != (!A1pt)) return HS_ERR_NULL;
    const size_t lds = (size_t)(kW1F + kW2TF) * sizeof(uint16_t);
    static hsLdsAttrOnce attr_a, attr_b;
    attr_a.set((const void *)k_rr_bwd_value<true, true>, (int)lds);
    attr_b.set((const void *)k_rr_bwd_value<false, true>, (int)lds);
    if (A0pt)
        k_rr_bwd_value<true, true><<<rr_grid(n), kThreadsW, lds, (hipStream_t)stream>>>((const uint16_t *)gy, (const uint16_t *)W2Tf, (const uint16_t *)W1Tf, (const uint16_t *)W0Tf,
                                                                                    (const uint16_t *)H0t, (const uint16_t *)H1t, (const uint16_t *)A0pt, (const uint16_t *)A1pt,
                                                                                    (uint16_t *)A0t, (uint16_t *)A1t, g_feat, n, ld, (const uint16_t *)W2Tf_b);
    else
        k_rr_bwd_value<false, true><<<rr_grid(n), kThreadsW, lds, (hipStream_t)stream>>>((const uint16_t *)gy, (const uint16_t *)W2Tf, (const uint16_t *)W1Tf, (const uint16_t *)W0Tf,
                                                                                     (const uint16_t *)H0t, (const uint16_t *)H1t, nullptr, nullptr, (uint16_t *)A0t, (uint16_t *)A1t,
                                                                                     g_feat, n, ld, (const uint16_t *)W2Tf_b);
    return wt_check_launch();
}

}  // extern "C"
