// holoscene_amd/csrc/wave_tile.h -- shared device machinery of the "wave tile" matrix-core kernels (sdf_mlp2.hip, trunk_mlp2.hip).
//
// A wave owns 32 rows end to end.  The product is formed as D[neuron][row] = W . H^T (v_mfma_f32_32x32x16_bf16): lane (row = lane & 31,
// half h = lane >> 5) receives, per 32-neuron tile, accumulator register r <-> neuron 8 (r >> 2) + 4 h + (r & 3).  After the
// activation and v_cvt_pk_bf16_f32 a tile is 8 packed words per lane that ARE the next layer's B-operand fragments of k-steps 2 nt and
// 2 nt + 1 once that layer's reduction index is permuted to k(s, h, e) = 16 s + 8 (e >> 2) + 4 h + (e & 3) -- done to the weights when
// they are packed into FRAGMENT ORDER ([k-step][neuron tile][lane] x 16 B).  W1 and W2 live in LDS in that order for the whole kernel,
// W0 streams from L2; activations never leave the registers.  See sdf_mlp2.hip for the measurements behind the structure.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>
#include <utility>

#include "holoscene_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

constexpr int kRows = 32;            // points per wave tile
constexpr int kWaves = 8;
constexpr int kThreadsW = 64 * kWaves;
constexpr int K0S = 5;               // k-steps of layer 0: 80 padded inputs
constexpr int HS = 16;               // k-steps of a 256-deep layer
constexpr int NT = 8;                // 32-neuron tiles of a 256-wide layer
constexpr int kW0F = K0S * NT * 64 * 8;   // bf16 elements of the packed matrices
constexpr int kW1F = HS * NT * 64 * 8;
constexpr int kW2F = HS * 64 * 8;
constexpr int kBias = 256 + 256 + 32;     // b0 (scaled) | b1 (scaled) | b2
constexpr float kAct = 100.f * 1.44269504f;   // log2-domain softplus of the inference kernel (sdf_mlp2.hip)

__device__ __forceinline__ uint32_t pack2(float a, float b) {   // one v_cvt_pk_bf16_f32 (round-to-nearest-even)
    const float2_t v = {a, b};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<const uint32_t *>(&r);
}

__device__ __forceinline__ bf16x8 frag_of(const uint32_t *p) {
    union { uint32_t u[4]; bf16x8 v; } c;
    c.u[0] = p[0]; c.u[1] = p[1]; c.u[2] = p[2]; c.u[3] = p[3];
    return c.v;
}

// input column (reference order: x, then per octave k sin(2^k x) cos(2^k x), then the 32 hash features) held by lane half h at
// position j of its 40-value list; -1 = zero padding
__host__ __device__ inline int input_column(int h, int j) {
    if (j < 18) return 3 + 18 * h + j;                 // octaves 3h .. 3h+2: [sin x3, cos x3] each
    if (j < 34) return 39 + 16 * h + (j - 18);         // hash levels 8h .. 8h+7, two channels each
    if (j < 37) return h == 0 ? j - 34 : -1;           // the raw coordinates ride in half 0
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------- weight packing
// fp32 effective (weight-normalised) matrices, row-major [out][in] -> bf16 fragment images + the bias block
// W2f is TWO bf16 planes, [W2 high | W2 low] (kW2F elements each; low = W2 - bf16(W2) in the same fragment order): the last layer's rows are a
// large common value plus small learned structure (geometric initialisation: 0.11 +- 1e-4 where bf16's grid is 4.9e-4), which a single plane
// loses -- measured: bf16 training then settles at another balance of its regularisers.  Every forward kernel forms
// y = W2hi h1 + W2lo h1; the resident LDS image stays [W1 | W2 high], the low plane is read from memory (16 KB, cache-resident) or kept in LDS
// where there is room (k_rr_fwd).
#ifndef HS_W2_LOW_PLANE
#define HS_W2_LOW_PLANE 1
#endif
constexpr bool kW2LowPlane = HS_W2_LOW_PLANE;     // (0: ablation -- the single-plane products of rounds 1-4)
constexpr int kSdfPackSlots = K0S * NT * 64 + HS * NT * 64 + 2 * HS * 64 + kBias;      // one thread per 16-byte fragment slot / bias entry
__device__ __forceinline__ void sdf_pack2_slot(int idx, const float *__restrict__ W0, int ld0, const float *__restrict__ b0, const float *__restrict__ W1,
                                               const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2, int d_out,
                                               uint16_t *__restrict__ W0f, uint16_t *__restrict__ W1f, uint16_t *__restrict__ W2f,
                                               float *__restrict__ bias, float act) {
    // act: factor folded into W0 and the hidden biases (its inverse into W2): 100 log2(e) for the inference kernel's log2-domain
    // softplus, 1 for the training kernel (which stores plain-domain activations for the backward pass)
    constexpr int n0 = K0S * NT * 64, n1 = HS * NT * 64, n2 = 2 * HS * 64;
    float v[8];
    uint16_t *dst;
    if (idx < n0) {
        const int s = idx / (NT * 64), nt = (idx / 64) % NT, lane = idx & 63, n = 32 * nt + (lane & 31), h = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int c = input_column(h, 8 * s + e);
            v[e] = c >= 0 ? act * W0[(size_t)n * ld0 + c] : 0.f;
        }
        dst = W0f + (size_t)idx * 8;
    } else if (idx < n0 + n1) {
        const int i = idx - n0, s = i / (NT * 64), nt = (i / 64) % NT, lane = i & 63, n = 32 * nt + (lane & 31), h = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = W1[(size_t)n * 256 + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3)];
        dst = W1f + (size_t)i * 8;
    } else if (idx < n0 + n1 + n2) {
        const int i = idx - n0 - n1, plane = i / (HS * 64), s = (i / 64) % HS, lane = i & 63, n = lane & 31, h = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float w = n < d_out ? W2[(size_t)n * 256 + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3)] * (1.f / act) : 0.f;
            v[e] = plane == 0 ? w : w - __uint_as_float(pack2(w, 0.f) << 16);      // low plane: what the high plane's rounding dropped
        }
        dst = W2f + (size_t)i * 8;
    } else {
        const int i = idx - n0 - n1 - n2;
        if (i < 256) bias[i] = b0[i] * act;
        else if (i < 512) bias[i] = b1[i - 256] * act;
        else if (i < kBias) bias[i] = (i - 512) < d_out ? b2[i - 512] : 0.f;
        return;
    }
    uint4 pk;
    pk.x = pack2(v[0], v[1]); pk.y = pack2(v[2], v[3]); pk.z = pack2(v[4], v[5]); pk.w = pack2(v[6], v[7]);
    *reinterpret_cast<uint4 *>(dst) = pk;
}

__global__ __launch_bounds__(256) void k_sdf_pack2(const float *__restrict__ W0, int ld0, const float *__restrict__ b0, const float *__restrict__ W1,
                                                   const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2, int d_out,
                                                   uint16_t *__restrict__ W0f, uint16_t *__restrict__ W1f, uint16_t *__restrict__ W2f,
                                                   float *__restrict__ bias, float act) {
    sdf_pack2_slot(blockIdx.x * 256 + threadIdx.x, W0, ld0, b0, W1, b1, W2, b2, d_out, W0f, W1f, W2f, bias, act);
}

// ---------------------------------------------------------------------------------------------------------------- tile machinery
// accumulator register r of a 32-neuron tile <-> neuron 8 (r >> 2) + 4 h + (r & 3)
__device__ __forceinline__ void init_acc(f32x16 &acc, const float *bias_tile, int h) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 b = *reinterpret_cast<const float4 *>(bias_tile + 8 * q + 4 * h);
        acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
    }
}

// One PHASE of a wave tile = the MFMAs of one neuron quarter (two 32-neuron tiles, KS k-steps) with, in their shadow, the softplus
// epilogue of the accumulator set the previous phase filled (two sets alternate).  Phases chain ACROSS layers: the first quarter of
// layer 1 only needs layer 0's quarters 0-2 for its k-steps 0..11, so layer 0's last epilogue is spread over its first E = 8 k-steps
// (likewise layer 2 under layer 1's last) -- no epilogue runs un-overlapped, and the matrix pipe and the VALU of ONE wave overlap
// without relying on the SIMD's other wave being in the opposite phase.
//   frag(s, j): A fragment of k-step s, tile j (0/1) of this quarter;  hin: B fragments of this layer (4 words per k-step);
//   epi(sl): slice sl of NSL of the previous phase's epilogue (16 packed pairs for the softplus kernel, 8 quads for the 4-row one);
//   E: k-steps over which those slices are spread;  AHEAD: how many k-steps the A fragments travel in front of their MFMAs
//   (explicit ring: at its register limit the scheduler otherwise issues every ds_read right before the MFMA that needs it).
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>).  The neuron-quarter loops index the accumulator sets and the
// packed-activation arrays by the quarter; when those loops are left to `#pragma unroll` the unrolling of a body this large happens after
// the last scalar-replacement pass and the arrays stay in scratch memory (measured on k_trunk_fwd2: 832 bytes per lane, 367 scratch
// instructions) -- with the index a template constant they are registers from the start.
template <int... Q, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Q...>, F &&f) { (f(std::integral_constant<int, Q>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

//   ZERO_C: the first k-step takes the inline constant 0 as its C operand instead of reading cur (no accumulator initialisation at all)
template <int KS, int AHEAD, int E, int NSL, bool EPI, bool ZERO_C = false, class FragFn, class EpiFn>
__device__ __forceinline__ void phase2(f32x16 (&cur)[2], const uint32_t *hin, FragFn frag, EpiFn epi) {
    bf16x8 ring[AHEAD + 1][2];
    static_for<(AHEAD < KS ? AHEAD : KS)>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        ring[s][0] = frag(s, 0);
        ring[s][1] = frag(s, 1);
    });
    static_for<KS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AHEAD < KS) {
            ring[(s + AHEAD) % (AHEAD + 1)][0] = frag(s + AHEAD, 0);
            ring[(s + AHEAD) % (AHEAD + 1)][1] = frag(s + AHEAD, 1);
        }
        const bf16x8 b = frag_of(hin + 4 * s);
        if constexpr (ZERO_C && s == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)][0], b, zero, 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)][1], b, zero, 0, 0, 0);
        } else {
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)][0], b, cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)][1], b, cur[1], 0, 0, 0);
        }
        if constexpr (EPI && s < E) {      // slices [s NSL / E, (s + 1) NSL / E) of the previous phase's epilogue
            constexpr int lo = (s * NSL) / E, hi = ((s + 1) * NSL) / E;
            static_for<hi - lo>([&](auto jc) { epi(std::integral_constant<int, lo + decltype(jc)::value>{}); });
        }
        __builtin_amdgcn_sched_barrier(0);      // pin the k-step order: loads of s + AHEAD | MFMAs of s | epilogue slice
    });
}

// The same for ONE 32-neuron tile per phase (layer 0 of the training kernel: half the accumulator and weight-fragment registers of a
// quarter phase; its five MFMAs form a dependent chain, which a VALU-bound layer can afford -- the SIMD's other wave fills the pipe).
template <int KS, int E, int NSL, bool EPI, bool ZERO_C = false, class FragFn, class EpiFn>
__device__ __forceinline__ void phase1(f32x16 &cur, const uint32_t *hin, FragFn frag, EpiFn epi) {
    static_for<KS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (ZERO_C && s == 0) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(s), frag_of(hin + 4 * s), zero, 0, 0, 0);
        } else {
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(s), frag_of(hin + 4 * s), cur, 0, 0, 0);
        }
        if constexpr (EPI && s < E) {
            constexpr int lo = (s * NSL) / E, hi = ((s + 1) * NSL) / E;
            static_for<hi - lo>([&](auto jc) { epi(std::integral_constant<int, lo + decltype(jc)::value>{}); });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// LDS addressing with an opaque per-lane base.  Written as `lds_array[const + lane_part]` every distinct constant becomes its own
// loop-invariant address VGPR once the resident image passes 64 KB (the 16-bit ds offset field cannot hold base + constant), and a
// kernel with a hundred such sites spills them all (k_trunk_fwd2: 60 spilled registers, 28 scratch reloads inside layer 0).  A base the
// compiler cannot see through plus a constant below 64 KB folds into `ds_read v, base offset:imm`: one VGPR per 64 KB window.
__device__ __forceinline__ uint32_t lds_base(const void *p, uint32_t lane_bytes) {
    uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p + lane_bytes;
    asm volatile("" : "+v"(a));
    return a;
}
__device__ __forceinline__ uint32_t relaunder(uint32_t a) {     // per loop trip: keeps base + constant from being hoisted as a VGPR each
    asm volatile("" : "+v"(a));
    return a;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ T lds_at(uint32_t base, uint32_t byte_off) {
    return *(__attribute__((address_space(3))) const T *)(uintptr_t)(base + byte_off);
}

// a workgroup-uniform 64-bit value into scalar registers (the 64-bit division behind a tile range runs on the vector ALU: without this
// the loop bounds live in -- and are spilled from -- vector registers)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// accumulator initialisation from the bias block through an opaque LDS base (wave_tile.h: lds_base -- written as bias[const + lane part] every
// one of the 128 distinct addresses of a tile becomes a loop-invariant VGPR)
__device__ __forceinline__ void init_acc_b(f32x16 &acc, uint32_t bias_base, int float_off) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f32x4 bq = lds_at<f32x4>(bias_base, (uint32_t)(float_off + 8 * q) * 4u);
        acc[4 * q + 0] = bq[0]; acc[4 * q + 1] = bq[1]; acc[4 * q + 2] = bq[2]; acc[4 * q + 3] = bq[3];
    }
}

// an empty volatile asm on a freshly computed value pins its computation HERE: the optimiser otherwise sinks an epilogue to the use of
// its result many MFMAs later, out of the MFMA shadow it was placed in (measured: layer 1's epilogues piled up in front of layer 2)
__device__ __forceinline__ uint32_t anchor(uint32_t w) {
    asm volatile("" : "+v"(w));
    return w;
}

// DPP helpers of the 4-row (value + three tangents) epilogue: the four rows of a point sit in the four lanes of a quad
__device__ __forceinline__ float quad_bcast0(float v) {  // value held by lane (lane & ~3) of each quad
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }

inline int wt_check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

}  // namespace
