// holoscene_amd/csrc/sdf_mlp32.hip -- fused SDF-trunk inference in the REFERENCE's own arithmetic: fp32 operands on the fp32 matrix cores (gfx950).
//
// The reference's Stage 1 is fp32 end to end (training/holoscene_train.py:45, no autocast), and its sampler evaluates the SDF branch of
// ObjectImplicitNetworkGrid.forward (model/network.py:169-210, via get_sdf_vals :305-311) on 5 x 128 x R points per iteration -- 85 % of all
// network evaluations.  In this package's fp32 configuration those sweeps were three library GEMMs + two Softplus launches + the input
// concatenation each (3.1 ms + 1.4 ms of an 11 ms iteration).  This is sdf_mlp2.hip's function -- positional encoding + hash features ->
// 71 -> 256 -> 256 -> d_out with Softplus(beta = 100), then the minimum the caller asked for -- on v_mfma_f32_32x32x2_f32 (157 TFLOP/s
// dense): fp32 products, fp32 accumulation, activations kept in fp32 registers from layer to layer.
//
//   * a WAVE owns 32 points end to end.  D[neuron][point] = W . H^T: lane (point = lane & 31, g = lane >> 5) receives accumulator register r
//     of a 32-neuron tile <-> neuron 8 (r >> 2) + 4 g + (r & 3).  One MFMA consumes k = 2 reduction indices -- lanes g = 0 supply one,
//     lanes g = 1 the other -- so REGISTER r OF TILE nt IS the B operand of reduction step s = 16 nt + r of the next layer, once that
//     layer's reduction index is permuted to k(s, g) = 32 (s >> 4) + 8 ((s & 15) >> 2) + 4 g + (s & 3); the permutation goes into the
//     packed weights.  No shuffle, no conversion, no LDS round trip between layers.
//   * W1 alone is 256 KB in fp32 -- it cannot be LDS-resident as in the bf16 kernel.  The WORKGROUP shares a weight pipeline instead
//     (appearance2.hip's scheme): the images are cut into one block per 32-neuron tile (W0: 10 KB, W1: 32 KB) that cycle through two LDS
//     buffers by LDS-DMA; the four waves of a workgroup -- each on its own 32 points -- meet once per block; W2's slice for a layer-1 tile rides
//     behind that tile's block, the biases stay resident.  Two workgroups share a compute unit (2 x 74 KB of LDS).
//   * layer 2 rides inside layer 1's tile loop: a finished tile's 16 activations feed 16 reduction steps of the output accumulator at
//     once, so layer 1's output is never held (the wave keeps vin[40] + h0[128] + two accumulators: ~210 registers, two waves per SIMD).
//   * Softplus as torch.nn.Softplus(beta = 100): log1p(exp(100 v)) / 100 below the threshold 100 v = 20, v above it, at fp32 accuracy
//     (softplus100_exact below: this kernel serves the configuration whose results must be the reference's).
#include "launch_util.h"
#include "wave_tile.h"

namespace {

constexpr int K0 = 40;                       // reduction steps of layer 0: the 40-entry input list of each lane half (wave_tile.h: input_column)
constexpr int HK = 128;                      // reduction steps of a 256-deep layer
constexpr int kW0Blk = K0 * 64;              // floats of one W0 block   [K0 / 4][64 lanes][4]
constexpr int kW2Slice = 16 * 64;            // floats of the 16 layer-2 reduction steps that consume ONE layer-1 tile   [4][64 lanes][4]
constexpr int kW1Blk = HK * 64 + kW2Slice;   //              W1 block   [HK / 4][64 lanes][4] | that tile's W2 slice  (36 KB)
constexpr int kW0I = NT * kW0Blk, kW1I = NT * kW1Blk;      // floats of the two images (W2 rides in W1's blocks)
constexpr int kBias32 = 256 + 256 + 32;
constexpr int kBufFloats = kW1Blk;           // one LDS buffer = the largest block
constexpr int kW32 = 4, kThreads32 = 64 * kW32;      // FOUR waves per workgroup, two workgroups per compute unit: the two waves of a SIMD then belong to
                                                     // different workgroups and drift into complementary phases (one in its MFMAs, the other in its
                                                     // Softplus epilogue) instead of meeting at the same block barrier (measured: 410 -> see DESIGN)

// reduction index of step s supplied by lane half g (see the header)
__host__ __device__ inline int k_of(int s, int g) { return 32 * (s >> 4) + 8 * ((s & 15) >> 2) + 4 * g + (s & 3); }

// ---------------------------------------------------------------------------------------------------------------- packing
// one thread per float4 slot of the three images (+ the bias block)
constexpr int kPack32Slots = (kW0I + kW1I) / 4 + kBias32;
__global__ __launch_bounds__(256) void k_sdf_pack32(const float *__restrict__ W0, int ld0, const float *__restrict__ b0, const float *__restrict__ W1,
                                                    const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2, int d_out,
                                                    float *__restrict__ W0i, float *__restrict__ W1i, float *__restrict__ bias) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    float4 v;
    float *dst;
    if (idx < kW0I / 4) {                    // [mt][j4][lane] x 4
        const int mt = idx / (K0 / 4 * 64), j4 = (idx / 64) % (K0 / 4), lane = idx & 63, m = 32 * mt + (lane & 31), g = lane >> 5;
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = input_column(g, 4 * j4 + e);
            t[e] = c >= 0 ? W0[(size_t)m * ld0 + c] : 0.f;
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
        dst = W0i + (size_t)idx * 4;
    } else if ((idx -= kW0I / 4) < kW1I / 4) {
        const int mt = idx / (kW1Blk / 4), in_blk = idx % (kW1Blk / 4), lane = idx & 63, g = lane >> 5;
        float t[4];
        if (in_blk < HK / 4 * 64) {          // W1 tile mt
            const int s4 = in_blk / 64, m = 32 * mt + (lane & 31);
#pragma unroll
            for (int e = 0; e < 4; e++) t[e] = W1[(size_t)m * 256 + k_of(4 * s4 + e, g)];
        } else {                             // W2's reduction steps 16 mt .. 16 mt + 15
            const int r4 = (in_blk - HK / 4 * 64) / 64, m = lane & 31;
#pragma unroll
            for (int e = 0; e < 4; e++) t[e] = m < d_out ? W2[(size_t)m * 256 + k_of(16 * mt + 4 * r4 + e, g)] : 0.f;
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
        dst = W1i + (size_t)idx * 4;
    } else if ((idx -= kW1I / 4) < kBias32) {
        bias[idx] = idx < 256 ? b0[idx] : (idx < 512 ? b1[idx - 256] : (idx - 512 < d_out ? b2[idx - 512] : 0.f));
        return;
    } else {
        return;
    }
    *reinterpret_cast<float4 *>(dst) = v;
}

// torch.nn.Softplus(beta = 100, threshold = 20): log1p(exp(t)) / 100 with t = 100 v, v itself above the threshold.  libm's expf + log1pf are
// ~150 instructions per activation -- 256 activations per lane and tile made the first version of this kernel VALU-bound at 2.5x its MFMA
// time (410 us per 131 072-point sweep).  The same real function in ~14 instructions at fp32 accuracy: log1p(exp(t)) = max(t, 0) +
// log1p(exp(-|t|)); e = exp(-|t|) in (0, 1] by the hardware exponential (its argument rounding, <= |t| 2^-24, is an ABSOLUTE error of e
// 2^-24 |t| <= 4e-8 on a result >= e / 2); log1p(e) = log(1 + e) with Kahan's correction for the rounding of 1 + e.  Agrees with the libm form
// to 2 ulp of the result (tests/test_model_gpu.py::test_fp32_fused_sdf_sweep_vs_library_gemms pins the kernel to 2e-5 of the library path
// end to end; measured 1e-6).
__device__ __forceinline__ float softplus100_exact(float v) {
    const float t = 100.f * v;
    const float e = __builtin_amdgcn_exp2f(-fabsf(t) * 1.44269504f);
    const float u = 1.f + e;
    const float l = __builtin_amdgcn_logf(u) * 0.69314718f - ((u - 1.f) - e) * __builtin_amdgcn_rcpf(u);
    return t > 20.f ? v : (fmaxf(t, 0.f) + l) * 0.01f;
}

// `bytes` (a multiple of 1 KB) from global to LDS by LDS-DMA, the 1 KB pieces dealt round-robin to the eight waves
__device__ __forceinline__ void dma_block(const float *__restrict__ src, float *dst, int bytes, int wave, int lane) {
    const int pieces = bytes / 1024;
    for (int c = wave; c < pieces; c += kW32)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)src + (size_t)c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)((char *)dst + (size_t)c * 1024), 16, 0, 0);
}

// acc += W_block . B over STEPS reduction steps; wl = this lane's float4 column of the block (LDS), bin[s] = this lane's B operand of step s
template <int STEPS>
__device__ __forceinline__ void block_mma(f32x16 &acc, const float *wl, const float (&bin)[STEPS]) {
    static_for<STEPS / 4>([&](auto sc) {
        constexpr int s4 = decltype(sc)::value;
        const float4 w = *reinterpret_cast<const float4 *>(wl + s4 * 256);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, bin[4 * s4 + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, bin[4 * s4 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, bin[4 * s4 + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, bin[4 * s4 + 3], acc, 0, 0, 0);
    });
}

__global__ __launch_bounds__(kThreads32, 2) void k_sdf_mlp32(const float *__restrict__ x, const float *__restrict__ feat, const float *__restrict__ W0i,
                                                              const float *__restrict__ W1i, const float *__restrict__ biasg,
                                                             int d_out, int select, uint64_t select_mask, float *__restrict__ out_min,
                                                             float *__restrict__ out_raw, int64_t B, hsGate gate, int feat_level_major) {
    extern __shared__ __attribute__((aligned(16))) float lds32[];
    if (gate.a != nullptr && !(*gate.a > *gate.b)) return;
    auto buf_at = [&](int par) { return lds32 + (par & 1) * kBufFloats; };
    float *bias = lds32 + 2 * kBufFloats;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), row = lane & 31, g = lane >> 5;
    for (int i = threadIdx.x; i < kBias32; i += kThreads32) bias[i] = biasg[i];
    const int64_t ntiles = (B + kRows - 1) / kRows;
    const int64_t rounds = (ntiles + (int64_t)gridDim.x * kW32 - 1) / ((int64_t)gridDim.x * kW32);      // the same for every wave: they meet per block
    // block sequence of a round: W0 tiles 0..7, W1 tiles 0..7; block q of the launch lives in buf[q & 1]
    auto request = [&](int b16, int par) {
        if (b16 < NT) dma_block(W0i + (size_t)b16 * kW0Blk, buf_at(par), kW0Blk * 4, wave, lane);
        else dma_block(W1i + (size_t)(b16 - NT) * kW1Blk, buf_at(par), kW1Blk * 4, wave, lane);
    };
    request(0, 0);
    int q = 0;          // blocks consumed so far (parity = buffer)
    for (int64_t rd = 0; rd < rounds; rd++) {
        const int64_t tile = (rd * gridDim.x + blockIdx.x) * kW32 + wave;
        const int64_t gp = tile * kRows + row;
        const bool ok = gp < B;
        // ---- this lane's 40 inputs (input_column order)
        float vin[K0];
        {
            const float x0 = ok ? x[gp * 3] : 0.f, x1 = ok ? x[gp * 3 + 1] : 0.f, x2 = ok ? x[gp * 3 + 2] : 0.f;
            const float xs[3] = {x0, x1, x2};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float f = g ? (float)(8 << k) : (float)(1 << k);    // octave 3 g + k
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    vin[6 * k + d] = sinf(xs[d] * f);
                    vin[6 * k + 3 + d] = cosf(xs[d] * f);
                }
            }
            if (feat_level_major) {      // feat [16, B, 2]
                const float2 *fl = reinterpret_cast<const float2 *>(feat) + (size_t)(8 * g) * B + (ok ? gp : 0);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float2 t = ok ? fl[(size_t)i * B] : make_float2(0.f, 0.f);
                    vin[18 + 2 * i] = t.x;
                    vin[19 + 2 * i] = t.y;
                }
            } else {                     // feat [B, 32]
                const float4 *fp = reinterpret_cast<const float4 *>(feat + (ok ? gp : 0) * 32 + 16 * g);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 t = ok ? fp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                    vin[18 + 4 * i] = t.x; vin[19 + 4 * i] = t.y; vin[20 + 4 * i] = t.z; vin[21 + 4 * i] = t.w;
                }
            }
            vin[34] = g ? 0.f : x0; vin[35] = g ? 0.f : x1; vin[36] = g ? 0.f : x2;
            vin[37] = vin[38] = vin[39] = 0.f;
        }
        // one block of the pipeline: block q has landed for every wave, every wave is done with block q - 1 (whose buffer the next request reuses)
        auto next_block = [&](int b16_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const bool more = b16_next < 2 * NT || rd + 1 < rounds;
            if (more) request(b16_next % (2 * NT), (q + 1) & 1);
            return buf_at(q) + lane * 4;
        };
        // ---- layer 0: 71 -> 256, Softplus; tile mt's 16 activations are layer 1's reduction steps 16 mt .. 16 mt + 15
        float h0[HK];
        static_for<NT>([&](auto mc) {
            constexpr int mt = decltype(mc)::value;
            const float *wl = next_block(mt + 1);
            f32x16 acc;
            init_acc(acc, bias + 32 * mt, g);
            block_mma<K0>(acc, wl, vin);
#pragma unroll
            for (int r = 0; r < 16; r++) h0[16 * mt + r] = softplus100_exact(acc[r]);
            q++;
        });
        // ---- layer 1: 256 -> 256, Softplus, with layer 2 (256 -> d_out) riding in its tile loop
        f32x16 y;
#pragma unroll
        for (int i = 0; i < 16; i++) y[i] = 0.f;
#pragma unroll 1
        for (int mt = 0; mt < NT; mt++) {
            const float *wl = next_block(NT + mt + 1);
            f32x16 acc;
            init_acc(acc, bias + 256 + 32 * mt, g);
            block_mma<HK>(acc, wl, h0);
            const float *w2 = wl + HK * 64;         // this tile's W2 slice behind its W1 block: [4][lane][4] = layer-2 steps 16 mt .. 16 mt + 15
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const float4 w = *reinterpret_cast<const float4 *>(w2 + r4 * 256);
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, softplus100_exact(acc[4 * r4 + 0]), y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, softplus100_exact(acc[4 * r4 + 1]), y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, softplus100_exact(acc[4 * r4 + 2]), y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, softplus100_exact(acc[4 * r4 + 3]), y, 0, 0, 0);
            }
            q++;
        }
        // ---- the K SDFs of the point and the minimum the caller asked for (sdf_mlp2.hip)
        float best = INFINITY;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int n = 8 * (i >> 2) + 4 * g + (i & 3);
            const float val = y[i] + bias[512 + n];
            y[i] = val;
            if (n < d_out) {
                const bool take = select_mask ? ((select_mask >> n) & 1ull) != 0ull : (select < 0 || n == select);
                if (take) best = fminf(best, val);
            }
        }
        best = fminf(best, __shfl_xor(best, 32));
        if (g == 0 && ok) out_min[gp] = best;
        if (out_raw && ok) {
            float *dst = out_raw + gp * d_out;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int n = 8 * (i >> 2) + 4 * g + (i & 3);
                if (n < d_out) dst[n] = y[i];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (nothing may still be landing in LDS when the workgroup retires)
}

}  // namespace

extern "C" {

int64_t hs_sdf_mlp32_pack_bytes(int32_t which) {
    switch (which) {
        case 0: return (int64_t)kW0I * 4;
        case 1: return (int64_t)kW1I * 4;       /* W1's tiles, each followed by the W2 slice its activations feed */
        case 2: return (int64_t)kBias32 * 4;
        default: return -1;
    }
}

int hs_sdf_mlp32_pack(const float *W0, int32_t ld0, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2, int32_t d_out,
                      float *W0i, float *W1i, float *bias, void *stream) {
    if (d_out < 1 || d_out > 32 || ld0 < 71) return HS_ERR_ARG;
    if (!W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !W0i || !W1i || !bias) return HS_ERR_NULL;
    k_sdf_pack32<<<(kPack32Slots + 255) / 256, 256, 0, (hipStream_t)stream>>>(W0, ld0, b0, W1, b1, W2, b2, d_out, W0i, W1i, bias);
    return wt_check_launch();
}

int hs_sdf_mlp32_fwd(const float *x, const float *feat, const float *W0i, const float *W1i, const float *bias, int32_t d_out,
                     int32_t select, uint64_t select_mask, float *out_min, float *out_raw, int64_t B, const hsGate *gate, int32_t feat_level_major,
                     void *stream) {
    if (d_out < 1 || d_out > 32 || select >= d_out || (feat_level_major != 0 && feat_level_major != 1)) return HS_ERR_ARG;
    if (select_mask && (select_mask >> d_out)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!x || !feat || !W0i || !W1i || !bias || !out_min) return HS_ERR_NULL;
    const size_t lds = (size_t)(2 * kBufFloats + kBias32) * sizeof(float);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_sdf_mlp32, (int)lds);
    const int64_t ntiles = (B + kRows - 1) / kRows, want = (ntiles + kW32 - 1) / kW32;
    k_sdf_mlp32<<<(int)(want < 512 ? want : 512), kThreads32, lds, (hipStream_t)stream>>>(x, feat, W0i, W1i, bias, d_out, select, select_mask, out_min,
                                                                                        out_raw, B, gate ? *gate : hsGate{nullptr, nullptr}, feat_level_major);
    return wt_check_launch();
}

}  // extern "C"
