// holoscene_amd/csrc/sdf_mlp2.hip -- fused SDF-trunk inference, "wave tile" form (gfx950).
//
// Same function as k_sdf_mlp (sdf_mlp.hip: the SDF branch of ObjectImplicitNetworkGrid.forward, model/network.py:169-210, for the
// sampler's no-grad sweeps :305-326), restructured around what bounded that kernel: it spent two thirds of its time in
// workgroup-wide phases (weight chunks L2 -> registers -> LDS behind a barrier per chunk, activations through an LDS tile, epilogue)
// that its 8 waves had to walk in lockstep, so the matrix pipe (18 % busy) and the VALU (softplus = exp2 + log2 per element: as many
// issue cycles as the MFMAs) took turns instead of overlapping.  Here:
//
//   * every WAVE owns 32 points end to end -- no __syncthreads in the steady state, so the two waves of a SIMD drift into
//     complementary phases (one in its MFMAs, the other in its epilogue) and both pipes stay busy;
//   * activations never leave the registers: the product is formed as D[neuron][point] = W . H^T (v_mfma_f32_32x32x16_bf16), whose
//     accumulator layout hands lane (point, half h) the neurons 8q + 4h + (0..3) -- after softplus and v_cvt_pk_bf16_f32 those are
//     eight bf16 per 16 neurons, i.e. EXACTLY a B-operand fragment of the next layer once that layer's reduction index is permuted
//     to match (a reduction may be summed in any order): k(s, h, e) = 16 s + 8 (e >> 2) + 4 h + (e & 3).  The permutation is applied
//     to the weights once, when they are packed;
//   * weights are packed in FRAGMENT ORDER ([k-step][neuron tile][lane] x 16 B): W1 (128 KB) and W2 (16 KB) live in LDS for the
//     whole kernel (one fill per workgroup; a fragment is one conflict-free ds_read_b128 of consecutive 16-byte slots), W0 (40 KB,
//     read once per 32 points) streams from L2 as fully coalesced 1 KB wave loads;
//   * the 71 network inputs are built by the lanes that consume them: lane (point, h) evaluates the three positional-encoding
//     octaves 3h..3h+2 and converts the hash features of levels 8h..8h+7 (80 padded inputs = 5 k-steps instead of 96 = 6).
//
// Softplus(beta = 100) is evaluated in the scaled domain of sdf_mlp.hip (t = 100 log2(e) v: log2(1 + 2^t)), the biases initialise
// the accumulators.  d_out <= 32 (one output tile); wider heads keep the workgroup-tile kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"

#ifdef HS_SDF2_PROFILE     // tools/exp/sdf2_prof.hip: per-phase s_memtime stamps of a steady-state wave tile
__device__ unsigned long long g_sdf2_prof[256 * 8 * 16];   // [block][wave][stamp]
#endif

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
constexpr float kAct = 100.f * 1.44269504f;

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

// log2(1 + 2^t); above t = 30 that IS t in fp32 (sdf_mlp.hip: softplus_scaled).  No clamp of the exponential's argument: beyond
// t = 128 it overflows to +inf, the logarithm returns +inf, and the select below never looks at it.
__device__ __forceinline__ float softplus_scaled2(float t) {
    const float l = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(t));
    return t > 30.f ? t : l;
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
__global__ __launch_bounds__(256) void k_sdf_pack2(const float *__restrict__ W0, int ld0, const float *__restrict__ b0, const float *__restrict__ W1,
                                                   const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2, int d_out,
                                                   uint16_t *__restrict__ W0f, uint16_t *__restrict__ W1f, uint16_t *__restrict__ W2f,
                                                   float *__restrict__ bias) {
    const int idx = blockIdx.x * 256 + threadIdx.x;      // one 16-byte fragment slot per thread
    constexpr int n0 = K0S * NT * 64, n1 = HS * NT * 64, n2 = HS * 64;
    float v[8];
    uint16_t *dst;
    if (idx < n0) {
        const int s = idx / (NT * 64), nt = (idx / 64) % NT, lane = idx & 63, n = 32 * nt + (lane & 31), h = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int c = input_column(h, 8 * s + e);
            v[e] = c >= 0 ? kAct * W0[(size_t)n * ld0 + c] : 0.f;
        }
        dst = W0f + (size_t)idx * 8;
    } else if (idx < n0 + n1) {
        const int i = idx - n0, s = i / (NT * 64), nt = (i / 64) % NT, lane = i & 63, n = 32 * nt + (lane & 31), h = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = W1[(size_t)n * 256 + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3)];
        dst = W1f + (size_t)i * 8;
    } else if (idx < n0 + n1 + n2) {
        const int i = idx - n0 - n1, s = i / 64, lane = i & 63, n = lane & 31, h = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = n < d_out ? W2[(size_t)n * 256 + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3)] * (1.f / kAct) : 0.f;
        dst = W2f + (size_t)i * 8;
    } else {
        const int i = idx - n0 - n1 - n2;
        if (i < 256) bias[i] = b0[i] * kAct;
        else if (i < 512) bias[i] = b1[i - 256] * kAct;
        else if (i < kBias) bias[i] = (i - 512) < d_out ? b2[i - 512] : 0.f;
        return;
    }
    uint4 pk;
    pk.x = pack2(v[0], v[1]); pk.y = pack2(v[2], v[3]); pk.z = pack2(v[4], v[5]); pk.w = pack2(v[6], v[7]);
    *reinterpret_cast<uint4 *>(dst) = pk;
}

// ---------------------------------------------------------------------------------------------------------------- the kernel
// accumulator register r of a 32-neuron tile <-> neuron 8 (r >> 2) + 4 h + (r & 3)
__device__ __forceinline__ void init_acc(f32x16 &acc, const float *bias_tile, int h) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 b = *reinterpret_cast<const float4 *>(bias_tile + 8 * q + 4 * h);
        acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
    }
}

// softplus + bf16 pack of accumulator registers r, r + 1 (r even) of a tile: half a register pair of the next layer's B fragment.
// Register r of tile nt lands in word (r >> 1) of that tile's 8-word block = k-steps 2 nt (words 0-3) and 2 nt + 1 (words 4-7).
__device__ __forceinline__ uint32_t epilogue_pair(const f32x16 &acc, int r) {
    uint32_t w = pack2(softplus_scaled2(acc[r]), softplus_scaled2(acc[r + 1]));
    // the value is consumed many MFMAs later (as a B operand of the next layer): without an anchor the optimiser SINKS the whole softplus
    // down to that use, out of the MFMA shadow it was placed in (measured: layer 1's epilogues piled up in front of layer 2)
    asm volatile("" : "+v"(w));
    return w;
}

// One PHASE of a wave tile = the MFMAs of one neuron quarter (two 32-neuron tiles, KS k-steps) with, in their shadow, the softplus
// epilogue of the accumulator set the previous phase filled (two sets alternate).  Phases chain ACROSS layers: the first quarter of
// layer 1 only needs layer 0's quarters 0-2 for its k-steps 0..11, so layer 0's last epilogue is spread over its first E = 8 k-steps
// (likewise layer 2 under layer 1's last) -- no epilogue runs un-overlapped, and the matrix pipe and the VALU of ONE wave overlap
// without relying on the SIMD's other wave being in the opposite phase.
//   frag(s, j): A fragment of k-step s, tile j (0/1) of this quarter;  hin: B fragments of this layer (4 words per k-step);
//   prev/hprev: the previous phase's accumulators and the 16 words (2 tiles x 8) their epilogue writes (nullptr: nothing to finish);
//   E: k-steps over which those 16 packed pairs are spread;  AHEAD: how many k-steps the A fragments travel in front of their MFMAs
//   (explicit ring: at its register limit the scheduler otherwise issues every ds_read right before the MFMA that needs it).
template <int KS, int AHEAD, int E, bool EPI, class FragFn>
__device__ __forceinline__ void phase2(f32x16 (&cur)[2], const f32x16 (&prev)[2], const uint32_t *hin, uint32_t *hprev, FragFn frag) {
    bf16x8 ring[AHEAD + 1][2];
#pragma unroll
    for (int s = 0; s < AHEAD && s < KS; s++) { ring[s][0] = frag(s, 0); ring[s][1] = frag(s, 1); }
#pragma unroll
    for (int s = 0; s < KS; s++) {
        if (s + AHEAD < KS) { ring[(s + AHEAD) % (AHEAD + 1)][0] = frag(s + AHEAD, 0); ring[(s + AHEAD) % (AHEAD + 1)][1] = frag(s + AHEAD, 1); }
        const bf16x8 b = frag_of(hin + 4 * s);
        cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)][0], b, cur[0], 0, 0, 0);
        cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)][1], b, cur[1], 0, 0, 0);
        if (EPI && s < E) {
#pragma unroll
            for (int sl = (s * 16) / E; sl < ((s + 1) * 16) / E; sl++) hprev[8 * (sl >> 3) + (sl & 7)] = epilogue_pair(prev[sl >> 3], 2 * (sl & 7));
        }
        __builtin_amdgcn_sched_barrier(0);      // pin the k-step order: loads of s + AHEAD | MFMAs of s | epilogue slice
    }
}

__global__ __launch_bounds__(kThreadsW, 2) void k_sdf_mlp2(const float *__restrict__ x, const float *__restrict__ feat, const uint16_t *__restrict__ W0f,
                                                            const uint16_t *__restrict__ W1f, const uint16_t *__restrict__ W2f,
                                                            const float *__restrict__ biasg, int d_out, int select, uint64_t select_mask,
                                                            float *__restrict__ out_min, float *__restrict__ out_raw, int64_t B, hsGate gate,
                                                            int feat_level_major) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    if (gate.a != nullptr && !(*gate.a > *gate.b)) return;
#ifdef HS_SDF2_PROFILE
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime(), prof_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    uint16_t *W1l = lds;
    uint16_t *W2l = lds + kW1F;
    float *bias = reinterpret_cast<float *>(W2l + kW2F);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, h = lane >> 5;
    // ---- one fill of the resident weights per workgroup (fragment order in global memory already: a straight copy of 144 KB), by
    //      LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no staging registers): the 18 requests of a wave are in flight
    //      while it builds the inputs of its first tile and runs layer 0 (whose weights come from L2); the workgroup meets once,
    //      before the first layer-1 MFMA.  (A load -> ds_write loop cost 16 us per launch, a batched register copy 6 us, measured.)
    {
        constexpr int kChunks = (kW1F + kW2F) * 2 / 1024;          // 144 x 1 KB; W2f follows W1f in the packed buffer
        static_assert(kChunks % kWaves == 0, "resident image must split evenly over the waves");
        const char *src = reinterpret_cast<const char *>(W1f);
        char *dst = reinterpret_cast<char *>(W1l);
#pragma unroll
        for (int i = 0; i < kChunks / kWaves; i++) {
            const int c = wave + i * kWaves;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(dst + (size_t)c * 1024), 16, 0, 0);
        }
        for (int i = threadIdx.x; i < kBias; i += kThreadsW) bias[i] = biasg[i];
    }
    __syncthreads();            // the bias block (layer 0 initialises its accumulators from it); the weight image is still landing
    bool resident = false;      // this wave has passed the rendezvous that makes the LDS weights visible
    const int64_t ntiles = (B + kRows - 1) / kRows;
    const bf16x8 *W0v = reinterpret_cast<const bf16x8 *>(W0f) + lane;
    const bf16x8 *W1v = reinterpret_cast<const bf16x8 *>(W1l) + lane;
    const bf16x8 *W2v = reinterpret_cast<const bf16x8 *>(W2l) + lane;
#ifdef HS_SDF2_PROFILE
#define HS_STAMP(i) do { if (lane == 0) g_sdf2_prof[(blockIdx.x * 8 + wave) * 16 + (tile >= (int64_t)gridDim.x * kWaves ? 0 : 8) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HS_STAMP(i) do { } while (0)
#endif
    for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < ntiles; tile += (int64_t)gridDim.x * kWaves) {
        HS_STAMP(0);
        const int64_t gp = tile * kRows + row;
        const bool ok = gp < B;
        // ---- this lane's 40 inputs (input_column order), as five B fragments
        uint32_t hin[4 * K0S];
        {
            float v[40];
            const float x0 = ok ? x[gp * 3] : 0.f, x1 = ok ? x[gp * 3 + 1] : 0.f, x2 = ok ? x[gp * 3 + 2] : 0.f;
            const float xs[3] = {x0, x1, x2};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float f = h ? (float)(8 << k) : (float)(1 << k);    // octave 3h + k
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    float sn, cs;
                    __sincosf(xs[d] * f, &sn, &cs);
                    v[6 * k + d] = sn;
                    v[6 * k + 3 + d] = cs;
                }
            }
            if (feat_level_major) {      // feat [16, B, 2]
                const float2 *fl = reinterpret_cast<const float2 *>(feat) + (size_t)(8 * h) * B + (ok ? gp : 0);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float2 t = ok ? fl[(size_t)i * B] : make_float2(0.f, 0.f);
                    v[18 + 2 * i] = t.x;
                    v[19 + 2 * i] = t.y;
                }
            } else {                     // feat [B, 32]
                const float4 *fp = reinterpret_cast<const float4 *>(feat + (ok ? gp : 0) * 32 + 16 * h);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 t = ok ? fp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                    v[18 + 4 * i] = t.x; v[19 + 4 * i] = t.y; v[20 + 4 * i] = t.z; v[21 + 4 * i] = t.w;
                }
            }
            v[34] = h ? 0.f : x0; v[35] = h ? 0.f : x1; v[36] = h ? 0.f : x2;
            v[37] = v[38] = v[39] = 0.f;
#pragma unroll
            for (int j = 0; j < 40; j += 2) hin[j >> 1] = pack2(v[j], v[j + 1]);
        }
        HS_STAMP(1);
        // ---- nine phases: layer 0 quarters 0-3 (weights from L2 in fragment order: one coalesced 1 KB load per wave and fragment,
        //      a quarter's ten fragments requested one phase ahead), layer 1 quarters 0-3 (weights resident in LDS), layer 2
        uint32_t h0p[64], h1p[64];
        f32x16 acc[2][2];
        {
            bf16x8 w0[2][2 * K0S];
            uint32_t zoff = 0;
            asm volatile("" : "+v"(zoff));     // opaque zero: keeps the (tile-invariant) fragment loads inside the tile loop, 160 registers otherwise
            const bf16x8 *W0q = W0v + zoff;
#define HS_W0_FETCH(q) do { _Pragma("unroll") for (int s_ = 0; s_ < K0S; s_++) { \
        w0[(q) & 1][2 * s_] = W0q[(size_t)(s_ * NT + 2 * (q)) * 64]; w0[(q) & 1][2 * s_ + 1] = W0q[(size_t)(s_ * NT + 2 * (q) + 1) * 64]; } } while (0)
            HS_W0_FETCH(0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                init_acc(acc[q & 1][0], bias + 32 * (2 * q), h);
                init_acc(acc[q & 1][1], bias + 32 * (2 * q + 1), h);
                if (q < 3) HS_W0_FETCH(q + 1);
                auto f0 = [&](int s, int j) { return w0[q & 1][2 * s + j]; };
                if (q == 0) phase2<K0S, 1, K0S, false>(acc[0], acc[1], hin, nullptr, f0);
                else phase2<K0S, 1, K0S, true>(acc[q & 1], acc[(q & 1) ^ 1], hin, h0p + 16 * (q - 1), f0);
            }
#undef HS_W0_FETCH
        }
        HS_STAMP(2);
        if (!resident) {        // first tile of this wave: its own DMA requests have landed, then everybody's
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            resident = true;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            init_acc(acc[q & 1][0], bias + 256 + 32 * (2 * q), h);
            init_acc(acc[q & 1][1], bias + 256 + 32 * (2 * q + 1), h);
            auto f1 = [&](int s, int j) { return W1v[(size_t)(s * NT + 2 * q + j) * 64]; };
            // q = 0 finishes layer 0's last quarter (needed from k-step 12 on) within its first 8 k-steps
            if (q == 0) phase2<HS, 2, 8, true>(acc[0], acc[1], h0p, h0p + 48, f1);
            else phase2<HS, 2, HS, true>(acc[q & 1], acc[(q & 1) ^ 1], h0p, h1p + 16 * (q - 1), f1);
        }
        HS_STAMP(3);
        // ---- layer 2: 256 -> d_out (<= 32) on two partial accumulators (even / odd k-steps: no dependent-MFMA chain), with layer 1's
        //      last epilogue in its first 8 k-steps; then the minimum the caller asked for
        f32x16 y;
        {
            f32x16 &y0 = acc[0][0], &y1 = acc[0][1];     // set 0 is free: layer 1's last quarter lives in set 1
#pragma unroll
            for (int i = 0; i < 16; i++) { y0[i] = 0.f; y1[i] = 0.f; }
            auto f2 = [&](int s, int j) { return W2v[(size_t)(2 * s + j) * 64]; };
            // "k-step" s of this phase = layer-2 k-steps 2s (accumulator 0) and 2s + 1 (accumulator 1); their B fragments are consecutive
            bf16x8 ring[3][2];
#pragma unroll
            for (int s = 0; s < 2; s++) { ring[s][0] = f2(s, 0); ring[s][1] = f2(s, 1); }
#pragma unroll
            for (int s = 0; s < HS / 2; s++) {
                if (s + 2 < HS / 2) { ring[(s + 2) % 3][0] = f2(s + 2, 0); ring[(s + 2) % 3][1] = f2(s + 2, 1); }
                if (s < 4) {      // k-steps 0..7 only need layer 1's quarters 0-1; quarter 3's epilogue rides here
#pragma unroll
                    for (int sl = 4 * s; sl < 4 * s + 4; sl++) h1p[48 + 8 * (sl >> 3) + (sl & 7)] = epilogue_pair(acc[1][sl >> 3], 2 * (sl & 7));
                }
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % 3][0], frag_of(h1p + 4 * (2 * s)), y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % 3][1], frag_of(h1p + 4 * (2 * s + 1)), y1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) y[i] = y0[i] + y1[i];
        }
        float best = INFINITY;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int n = 8 * (i >> 2) + 4 * h + (i & 3);
            const float val = y[i] + bias[512 + n];
            y[i] = val;
            if (n < d_out) {
                const bool take = select_mask ? ((select_mask >> n) & 1ull) != 0ull : (select < 0 || n == select);
                if (take) best = fminf(best, val);
            }
        }
        best = fminf(best, __shfl_xor(best, 32));
        if (h == 0 && ok) out_min[gp] = best;
        if (out_raw && ok) {
            float *dst = out_raw + gp * d_out;
            if ((d_out & 3) == 0) {      // 16-byte runs: neurons 8 q + 4 h .. + 3
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (8 * q + 4 * h < d_out) *reinterpret_cast<float4 *>(dst + 8 * q + 4 * h) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                    if (n < d_out) dst[n] = y[i];
                }
            }
        }
        HS_STAMP(4);
    }
    if (!resident) {            // a wave without a tile still owes the workgroup its arrival (and its share of the copy)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#ifdef HS_SDF2_PROFILE
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *p = g_sdf2_prof + (blockIdx.x * 8 + (threadIdx.x >> 6)) * 16;
        p[5] = __builtin_amdgcn_s_memtime() - prof_t0;
        p[6] = __builtin_amdgcn_s_memrealtime() - prof_r0;
        p[7] = prof_t0;
    }
#endif
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

}  // namespace

extern "C" {

int64_t hs_sdf_mlp2_pack_bytes(int32_t which) {
    switch (which) {
        case 0: return (int64_t)kW0F * 2;
        case 1: return (int64_t)kW1F * 2;
        case 2: return (int64_t)kW2F * 2;
        case 3: return (int64_t)kBias * 4;
        default: return -1;
    }
}

int hs_sdf_mlp2_pack(const float *W0, int32_t ld0, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2, int32_t d_out,
                     void *W0f, void *W1f, void *W2f, float *bias, void *stream) {
    if (d_out < 1 || d_out > 32 || ld0 < 71) return HS_ERR_ARG;
    if (!W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !W0f || !W1f || !W2f || !bias) return HS_ERR_NULL;
    const int slots = K0S * NT * 64 + HS * NT * 64 + HS * 64 + kBias;
    k_sdf_pack2<<<(slots + 255) / 256, 256, 0, (hipStream_t)stream>>>(W0, ld0, b0, W1, b1, W2, b2, d_out, (uint16_t *)W0f, (uint16_t *)W1f,
                                                                      (uint16_t *)W2f, bias);
    return check_launch();
}

int hs_sdf_mlp2_fwd(const float *x, const float *feat, const void *W0f, const void *W1f, const void *W2f, const float *bias, int32_t d_out,
                    int32_t select, uint64_t select_mask, float *out_min, float *out_raw, int64_t B, const hsGate *gate, int32_t feat_level_major,
                    void *stream) {
    if (d_out < 1 || d_out > 32 || select >= d_out) return HS_ERR_ARG;
    if (select_mask && (select_mask >> d_out)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!x || !feat || !W0f || !W1f || !W2f || !bias || !out_min) return HS_ERR_NULL;
    if ((const char *)W2f != (const char *)W1f + (size_t)kW1F * 2) return HS_ERR_ARG;   // the resident image is copied in one sweep: W2f directly behind W1f
    const size_t lds = (size_t)(kW1F + kW2F) * sizeof(uint16_t) + kBias * sizeof(float);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)k_sdf_mlp2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    const int64_t ntiles = (B + kRows - 1) / kRows;
    const int64_t want = (ntiles + kWaves - 1) / kWaves;
    const int grid = (int)(want < 256 ? want : 256);      // one workgroup per CU (146 KB of LDS), wave tiles strided across the grid
    k_sdf_mlp2<<<grid, kThreadsW, lds, (hipStream_t)stream>>>(x, feat, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias, d_out,
                                                               select, select_mask, out_min, out_raw, B, gate ? *gate : hsGate{nullptr, nullptr},
                                                               feat_level_major);
    return check_launch();
}

}  // extern "C"
