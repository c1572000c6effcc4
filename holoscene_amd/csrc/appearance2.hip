// holoscene_amd/csrc/appearance2.hip -- the colour branch of a rendered sample in "wave tile" form (gfx950).
//
// Same function as appearance_mlp.hip (k_appear_fwd / k_appear_bwd: color_grid_feature_map_mlp, model/network.py:99-101, 186-188; the
// positional encodings + concatenation and the three weight-normalised layers of RenderingNetwork.forward, :586-612), restructured the
// way sdf_mlp2.hip restructured the SDF trunk: a WAVE owns 32 samples through all five layers and its activations stay in registers as
// the next product's B fragments (wave_tile.h).  The trunk's one 256 x 256 matrix is LDS-resident there; here five matrices (448 KB of
// fragments) do not fit, and streaming them per wave from L2 was measured latency-bound (a register ring cannot travel far enough ahead
// of a single-accumulator MFMA chain: 119 us, slower than the kernel this replaces).  So the WORKGROUP shares a weight pipeline: the
// fragment image is cut into eight chunks of whole neuron tiles (16-66 KB) that cycle through two LDS buffers by LDS-DMA; the eight waves
// -- each on its own 32-sample tile -- meet once per chunk (the barrier that publishes chunk j also frees the buffer chunk j + 1 is then
// requested into), i.e. eight barriers per 256 samples instead of the ~40 of the workgroup-tile kernel, and no activation ever touches LDS.
// What crosses kernels is TILE-PACKED (trunk_rr.hip): the four layer outputs the weight gradients need (hc, fv, r0, r1), the assembled
// inputs [colour features | encodings] as an 8-k-step image, and the ReLU signs as 128 bits per lane and layer -- the backward pass reads
// the signs only.
//
//   forward   hc = relu(Wc0 f + bc0);  fv = Wc1 hc + bc1;  r0 = relu(Wr0 [enc | fv] + br0);  r1 = relu(Wr1 r0 + br1);  rgb = sigmoid(Wr2 r1 + br2)
#include "launch_util.h"
#include <stdlib.h>
#include "wave_tile.h"
#include "trunk_pack.h"

#ifndef HS_A2_NOSTORE
#define HS_A2_NOSTORE 0      // 1 (a variant build): k_appear2_bwd without its cotangent stores -- the ablation of DESIGN 14.3
#endif

namespace {

#ifndef HS_A2_LA
#define HS_A2_LA 1
#endif
constexpr int kLA = HS_A2_LA;          // k-steps an A fragment is read from LDS ahead of its MFMA (1: backward 66.4 -> 64.4 us, forward 69.9 -> 69.2;
                                       // bench median -3 .. -9 us in six alternating pairs on two boxes; 0 the same as 1, 3 slower than 2)
constexpr int XAS = 8;                 // k-steps of the assembled-input image: 2 (colour features) + 6 (encodings, 81 -> 96)
constexpr int kA2Bias = 4 * 256 + 32;  // bc0 | bc1 | br0 | br1 | br2
// the streamed fragment image: per layer, per 32-neuron tile, [k-steps][64 lanes] x 16 B
constexpr int kC0Tile = 2 * 1024, kBigTile = 16 * 1024, kR0Tile = 22 * 1024;           // bytes per neuron tile (R0: 6 encoding + 16 feature k-steps)
constexpr int kOffC0 = 0, kOffC1 = kOffC0 + 8 * kC0Tile, kOffR0 = kOffC1 + 8 * kBigTile, kOffR1 = kOffR0 + 8 * kR0Tile, kStreamBytes = kOffR1 + 8 * kBigTile;
constexpr int kBufBytes = 4 * kBigTile;  // 64 KB: the largest chunk
// chunks of whole neuron-tile PAIRS (a phase works on two tiles); consecutive chunks alternate between the two LDS buffers
constexpr int kChunks = 9;
constexpr int kChunkOff[kChunks] = {kOffC0, kOffC1, kOffC1 + 4 * kBigTile, kOffR0, kOffR0 + 2 * kR0Tile, kOffR0 + 4 * kR0Tile, kOffR0 + 6 * kR0Tile, kOffR1,
                                    kOffR1 + 4 * kBigTile};
constexpr int kChunkBytes[kChunks] = {8 * kC0Tile, 4 * kBigTile, 4 * kBigTile, 2 * kR0Tile, 2 * kR0Tile, 2 * kR0Tile, 2 * kR0Tile, 4 * kBigTile, 4 * kBigTile};

// the backward pass streams the TRANSPOSED matrices the same way: W_R2^T (8 tiles x 1 k-step: only outputs 0..2 exist), W_R1^T, W_R0^T's
// feature rows (8 tiles) + its 27 encoded-normal rows (1 tile), W_C1^T, W_C0^T (1 tile: the 32 colour features)
constexpr int kT2Tile = 1 * 1024;
constexpr int kOffT2 = 0, kOffT1 = kOffT2 + 8 * kT2Tile, kOffT0 = kOffT1 + 8 * kBigTile, kOffTN = kOffT0 + 8 * kBigTile, kOffTC1 = kOffTN + kBigTile,
              kOffTC0 = kOffTC1 + 8 * kBigTile, kStreamTBytes = kOffTC0 + kBigTile;
constexpr int kChunksT = 9;
constexpr int kChunkTOff[kChunksT] = {kOffT2, kOffT1, kOffT1 + 4 * kBigTile, kOffT0, kOffT0 + 4 * kBigTile, kOffTN, kOffTC1, kOffTC1 + 4 * kBigTile, kOffTC0};
constexpr int kChunkTBytes[kChunksT] = {8 * kT2Tile, 4 * kBigTile, 4 * kBigTile, 4 * kBigTile, 4 * kBigTile, kBigTile, 4 * kBigTile, 4 * kBigTile, kBigTile};
// row i of W_C0^T's output tile <-> colour feature: lane half h then receives features 16 h .. 16 h + 15 (levels 8 h .. 8 h + 7) in registers 0..15
__host__ __device__ inline int featc_of_row(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

// column of the rendering network's 81 encoded inputs ([point | view dir | normal], each [v, sin 2^k v, cos 2^k v (k = 0..3)]: embedder.py:22-36)
// held by lane half h at slot j (0..47) of its encoding list; -1 = padding.  Half h evaluates frequencies 2h, 2h + 1 of all three vectors,
// half 0 also carries the nine raw coordinates.
__host__ __device__ inline int enc_column(int h, int j) {
    if (j < 36) {
        const int t = j / 12, kk = (j % 12) / 6, comp = j % 6;
        return 27 * t + 3 + 6 * (2 * h + kk) + comp;
    }
    if (j < 45) return h == 0 ? 27 * ((j - 36) / 3) + (j - 36) % 3 : -1;
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------- weight images
// fp32 effective matrices -> the streamed image (kStreamBytes), W_R2's fragments (one tile, resident) and the bias block
constexpr int kPackSlots = kStreamBytes / 16 + HS * 64 + kA2Bias;      // slots of the forward images (one thread each)
__device__ __forceinline__ void pack_slot(int idx, const float *__restrict__ Wc0, const float *__restrict__ Wc1, const float *__restrict__ Wr0, int ldr0,
                                          const float *__restrict__ Wr1, const float *__restrict__ Wr2, const float *__restrict__ bc0,
                                          const float *__restrict__ bc1, const float *__restrict__ br0, const float *__restrict__ br1,
                                          const float *__restrict__ br2, uint16_t *__restrict__ stream, uint16_t *__restrict__ R2f,
                                          float *__restrict__ bias) {
    constexpr int nstream = kStreamBytes / 16, nr2 = HS * 64;
    float v[8];
    uint16_t *dst;
    if (idx < nstream) {
        const int byte = idx * 16, lane = idx & 63, m = lane & 31, h = lane >> 5;
        auto kperm = [&](int s, int e) { return 16 * s + 8 * (e >> 2) + 4 * h + (e & 3); };
        if (byte < kOffC1) {                       // colour features: slot 8 s + e of half h <-> feature 16 h + 8 s + e (levels 8 h .. 8 h + 7)
            const int nt = byte / kC0Tile, s = (byte % kC0Tile) / 1024, n = 32 * nt + m;
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = Wc0[(size_t)n * 32 + 16 * h + 8 * s + e];
        } else if (byte < kOffR0) {
            const int b2 = byte - kOffC1, nt = b2 / kBigTile, s = (b2 % kBigTile) / 1024, n = 32 * nt + m;
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = Wc1[(size_t)n * 256 + kperm(s, e)];
        } else if (byte < kOffR1) {
            const int b2 = byte - kOffR0, nt = b2 / kR0Tile, s = (b2 % kR0Tile) / 1024, n = 32 * nt + m;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (s < 6) {
                    const int c = enc_column(h, 8 * s + e);
                    v[e] = c >= 0 ? Wr0[(size_t)n * ldr0 + c] : 0.f;
                } else {
                    v[e] = Wr0[(size_t)n * ldr0 + 81 + kperm(s - 6, e)];
                }
            }
        } else {
            const int b2 = byte - kOffR1, nt = b2 / kBigTile, s = (b2 % kBigTile) / 1024, n = 32 * nt + m;
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = Wr1[(size_t)n * 256 + kperm(s, e)];
        }
        dst = stream + (size_t)idx * 8;
    } else if (idx < nstream + nr2) {
        const int i = idx - nstream, s = i / 64, lane = i & 63, n = lane & 31, h = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = n < 3 ? Wr2[(size_t)n * 256 + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3)] : 0.f;
        dst = R2f + (size_t)i * 8;
    } else if (idx < nstream + nr2 + kA2Bias) {
        const int i = idx - nstream - nr2;
        bias[i] = i < 256 ? bc0[i] : i < 512 ? bc1[i - 256] : i < 768 ? br0[i - 512] : i < 1024 ? br1[i - 768] : (i - 1024 < 3 ? br2[i - 1024] : 0.f);
        return;
    } else {
        return;
    }
    uint4 pk;
    pk.x = pack2(v[0], v[1]); pk.y = pack2(v[2], v[3]); pk.z = pack2(v[4], v[5]); pk.w = pack2(v[6], v[7]);
    *reinterpret_cast<uint4 *>(dst) = pk;
}

// the transposed image for the backward pass (kStreamTBytes): every product is D[input unit][sample] = sum_k W[k][input unit] g[k][sample], i.e. the A
// operand is W^T with the OUTPUT unit k of the layer as the reduction index, permuted like every 256-deep reduction (wave_tile.h)
__device__ __forceinline__ void pack_t_slot(int idx, const float *__restrict__ Wc0, const float *__restrict__ Wc1, const float *__restrict__ Wr0, int ldr0,
                                            const float *__restrict__ Wr1, const float *__restrict__ Wr2, uint16_t *__restrict__ streamT) {
    if (idx >= kStreamTBytes / 16) return;
    const int byte = idx * 16, lane = idx & 63, m = lane & 31, h = lane >> 5;
    auto kperm = [&](int s, int e) { return 16 * s + 8 * (e >> 2) + 4 * h + (e & 3); };
    float v[8];
    if (byte < kOffT1) {                            // W_R2^T: rows = 256 units of r1, k = the 3 outputs (k-step 0 only)
        const int nt = byte / kT2Tile, u = 32 * nt + m;
#pragma unroll
        for (int e = 0; e < 8; e++) { const int k = kperm(0, e); v[e] = k < 3 ? Wr2[(size_t)k * 256 + u] : 0.f; }
    } else if (byte < kOffT0) {                     // W_R1^T
        const int b2 = byte - kOffT1, nt = b2 / kBigTile, s = (b2 % kBigTile) / 1024, u = 32 * nt + m;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = Wr1[(size_t)kperm(s, e) * 256 + u];
    } else if (byte < kOffTN) {                     // W_R0^T, feature-vector rows (columns 81.. of W_R0)
        const int b2 = byte - kOffT0, nt = b2 / kBigTile, s = (b2 % kBigTile) / 1024, u = 32 * nt + m;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = Wr0[(size_t)kperm(s, e) * ldr0 + 81 + u];
    } else if (byte < kOffTC1) {                    // W_R0^T, the 27 encoded-normal rows (columns 54..80), padded to one tile
        const int s = (byte - kOffTN) / 1024;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = m < 27 ? Wr0[(size_t)kperm(s, e) * ldr0 + 54 + m] : 0.f;
    } else if (byte < kOffTC0) {                    // W_C1^T
        const int b2 = byte - kOffTC1, nt = b2 / kBigTile, s = (b2 % kBigTile) / 1024, u = 32 * nt + m;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = Wc1[(size_t)kperm(s, e) * 256 + u];
    } else {                                        // W_C0^T: row i <-> colour feature featc_of_row(i)
        const int s = (byte - kOffTC0) / 1024;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = Wc0[(size_t)kperm(s, e) * 32 + featc_of_row(m)];
    }
    uint4 pk;
    pk.x = pack2(v[0], v[1]); pk.y = pack2(v[2], v[3]); pk.z = pack2(v[4], v[5]); pk.w = pack2(v[6], v[7]);
    *reinterpret_cast<uint4 *>(streamT + (size_t)idx * 8) = pk;
}

// both images in one launch (streamT may be NULL: forward only)
__global__ __launch_bounds__(256) void k_appear2_pack(const float *__restrict__ Wc0, const float *__restrict__ Wc1, const float *__restrict__ Wr0, int ldr0,
                                                      const float *__restrict__ Wr1, const float *__restrict__ Wr2, const float *__restrict__ bc0,
                                                      const float *__restrict__ bc1, const float *__restrict__ br0, const float *__restrict__ br1,
                                                      const float *__restrict__ br2, uint16_t *__restrict__ stream, uint16_t *__restrict__ R2f,
                                                      float *__restrict__ bias, uint16_t *__restrict__ streamT) {
    const int idx = blockIdx.x * 256 + threadIdx.x;       // one 16-byte fragment slot per thread
    if (idx < kPackSlots) pack_slot(idx, Wc0, Wc1, Wr0, ldr0, Wr1, Wr2, bc0, bc1, br0, br1, br2, stream, R2f, bias);
    else if (streamT != nullptr) pack_t_slot(idx - kPackSlots, Wc0, Wc1, Wr0, ldr0, Wr1, Wr2, streamT);
}

// Every weight image of a Stage-1 iteration -- the sampler sweeps' trunk images (log2-domain softplus, sdf_mlp2.hip), the training trunk's
// (trunk_pack.h), the colour branch's (above) -- in ONE launch: they are functions of the same ~0.3 M weights, and the three separate
// pack launches of an iteration cost ~5 us each whatever they do.
struct PackIterArgs {
    // trunk
    const float *W0, *b0, *W1, *b1, *W2, *b2;
    int32_t ld0, f_in, d_out;
    uint16_t *sW0f, *sW1f, *sW2f; float *sbias;                                         // sampler images (NULL: none)
    uint16_t *W0f, *W1f, *W2f; float *bias; uint16_t *W1Tf, *W0Tf, *W2Tf; float *W2tab;  // training images (NULL: none)
    uint16_t *w1t, *w2t, *w0t;                                                          // + row-major transposes (NULL: none)
    // colour branch (stream NULL: none)
    const float *Wc0, *Wc1, *Wr0, *Wr1, *Wr2, *bc0, *bc1, *br0, *br1, *br2;
    int32_t ldr0;
    uint16_t *stream, *R2f; float *abias; uint16_t *streamT;
};
constexpr int kIterPackSlots = kSdfPackSlots + kTrunkPackSlots + kPackSlots + kStreamTBytes / 16;

__global__ __launch_bounds__(256) void k_pack_iteration(PackIterArgs a) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < kSdfPackSlots) {
        if (a.sW0f) sdf_pack2_slot(idx, a.W0, a.ld0, a.b0, a.W1, a.b1, a.W2, a.b2, a.d_out, a.sW0f, a.sW1f, a.sW2f, a.sbias, kAct);
        return;
    }
    idx -= kSdfPackSlots;
    if (idx < kTrunkPackSlots) {
        if (a.W0f) trunk_pack_all_slot(idx, a.W0, a.ld0, a.f_in, a.b0, a.W1, a.b1, a.W2, a.b2, a.d_out, a.W0f, a.W1f, a.W2f, a.bias, a.W1Tf, a.W0Tf, a.W2Tf, a.W2tab,
                                       a.w1t, a.w2t, a.w0t);
        return;
    }
    idx -= kTrunkPackSlots;
    if (!a.stream) return;
    if (idx < kPackSlots) pack_slot(idx, a.Wc0, a.Wc1, a.Wr0, a.ldr0, a.Wr1, a.Wr2, a.bc0, a.bc1, a.br0, a.br1, a.br2, a.stream, a.R2f, a.abias);
    else if (a.streamT != nullptr) pack_t_slot(idx - kPackSlots, a.Wc0, a.Wc1, a.Wr0, a.ldr0, a.Wr1, a.Wr2, a.streamT);
}

// ---------------------------------------------------------------------------------------------------------------- helpers
typedef short short2_t __attribute__((ext_vector_type(2)));

// (the lane part of the address passes through an opaque register produced HERE: the scheduler otherwise computes the ~90 store addresses
// of a tile at its top and spills them -- 80 of the first version's 120 spilled registers)
#ifndef HS_NT_A2_FWD
#define HS_NT_A2_FWD 1
#endif
#ifndef HS_NT_A2_BWD
#define HS_NT_A2_BWD 0
#endif
template <bool NT = false>      // NT: a non-temporal store (trunk_rr.hip: tp_store)
__device__ __forceinline__ void tp_store_n(uint16_t *__restrict__ T, int64_t tile, int ksteps, int s, int lane, const uint32_t *w4) {
    uint32_t lo = (uint32_t)lane * 8u;
    asm volatile("" : "+v"(lo));
    if constexpr (NT) {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const u32x4_t vv = {w4[0], w4[1], w4[2], w4[3]};
        __builtin_nontemporal_store(vv, reinterpret_cast<u32x4_t *>(T + ((size_t)tile * ksteps + s) * 512 + lo));
    } else {
        *reinterpret_cast<uint4 *>(T + ((size_t)tile * ksteps + s) * 512 + lo) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
}

// resident image by LDS-DMA (sdf_mlp2.hip), `bytes` a multiple of 1 KB
__device__ __forceinline__ void dma_fill2(const void *src, void *dst, int bytes, int wave, int lane) {
    const int chunks = bytes / 1024;
    for (int c = wave; c < chunks; c += kWaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)src + (size_t)c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)((char *)dst + (size_t)c * 1024), 16, 0, 0);
}

// One phase = the KS k-steps of a neuron QUARTER (two 32-neuron tiles: two independent accumulator chains, so that a wave's MFMAs issue back
// to back instead of at the latency of a dependent chain -- measured on the single-tile form: 145 cycles per MFMA) with slices of the
// previous quarter's epilogue in their shadow (wave_tile.h: phase2); both operands come from callables: afn(s, j) the A fragment of tile j,
// bfn(s) the B fragment (a layer may read two arrays).
template <int KS, int AHEAD, int E, int NSL, bool EPI, class AFn, class BFn, class EpiFn>
__device__ __forceinline__ void phase2g(f32x16 (&cur)[2], AFn afn, BFn bfn, EpiFn epi) {
    bf16x8 ring[AHEAD + 1][2];
    static_for<(AHEAD < KS ? AHEAD : KS)>([&](auto sc) { constexpr int s = decltype(sc)::value; ring[s][0] = afn(s, 0); ring[s][1] = afn(s, 1); });
    static_for<KS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + AHEAD < KS) { ring[(s + AHEAD) % (AHEAD + 1)][0] = afn(s + AHEAD, 0); ring[(s + AHEAD) % (AHEAD + 1)][1] = afn(s + AHEAD, 1); }
        const bf16x8 bq = bfn(s);
        cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)][0], bq, cur[0], 0, 0, 0);
        cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (AHEAD + 1)][1], bq, cur[1], 0, 0, 0);
        if constexpr (EPI && s < E) {
            constexpr int lo = (s * NSL) / E, hi = ((s + 1) * NSL) / E;
            static_for<hi - lo>([&](auto jc) { epi(std::integral_constant<int, lo + decltype(jc)::value>{}); });
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// ReLU of an accumulator register pair on the PACKED bf16 word: a bf16 bit pattern read as int16 is negative exactly when the value is, so
// one v_pk_max_i16 against zero is the ReLU of both halves; the two sign bits (bit 15, bit 31) are the mask the backward pass needs (set =
// unit off) and go to bits `bit`, `bit + 16` of m.  (-0 counts as off, +0 as on: a pre-activation that is exactly zero.)
__device__ __forceinline__ uint32_t relu_pair(float a, float b, uint32_t &m, int bit) {
    const uint32_t p = pack2(a, b);
    m |= ((p >> 15) & 0x00010001u) << bit;
    const short2_t r = __builtin_elementwise_max(*reinterpret_cast<const short2_t *>(&p), (short2_t){0, 0});
    return anchor(*reinterpret_cast<const uint32_t *>(&r));
}

// ================================================================================================================ forward
// chunk `j` of the streamed image -> LDS buffer `par`, 1 KB per wave instruction, the pieces dealt round-robin to the eight waves
__device__ __forceinline__ void dma_chunk(const char *__restrict__ stream, char *lds, int off, int bytes, int par, int wave, int lane) {
    const char *src = stream + off;
    char *dst = lds + par * kBufBytes;
    const int pieces = bytes / 1024;
    uint32_t lo = (uint32_t)lane * 16u;       // opaque: the request addresses of a super-tile are loop invariants the compiler would keep (and spill)
    asm volatile("" : "+v"(lo));
    for (int c = wave; c < pieces; c += kWaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)c * 1024 + lo),
                                         (__attribute__((address_space(3))) void *)(dst + (size_t)c * 1024), 16, 0, 0);
}

__global__ __launch_bounds__(kThreadsW, 2) void k_appear2_fwd(const float *__restrict__ featc, const float *__restrict__ points, const float *__restrict__ dirs,
                                                               const float *__restrict__ normals, const char *__restrict__ stream,
                                                               const uint16_t *__restrict__ R2f, const float *__restrict__ biasg,
                                                               uint16_t *__restrict__ XAt, uint16_t *__restrict__ HCt, uint16_t *__restrict__ FVt,
                                                               uint16_t *__restrict__ R0t, uint16_t *__restrict__ R1t, uint32_t *__restrict__ masks,
                                                               float *__restrict__ rgb, int64_t n, int featc_words) {
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    uint16_t *R2l = reinterpret_cast<uint16_t *>(lds2 + 2 * kBufBytes);
    float *bias = reinterpret_cast<float *>(lds2 + 2 * kBufBytes + kW2F * 2);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), row = lane & 31, h = lane >> 5;
    int par = 0;                                 // LDS buffer of the chunk about to be consumed (workgroup-uniform, toggles per chunk)
    dma_chunk(stream, lds2, kChunkOff[0], kChunkBytes[0], 0, wave, lane);
    dma_fill2(R2f, R2l, kW2F * 2, wave, lane);
    for (int i = threadIdx.x; i < kA2Bias; i += kThreadsW) bias[i] = biasg[i];
    __syncthreads();        // the bias block, the resident last layer (the chunk barriers below are bare barrier instructions)
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const bf16x8 *R2v = reinterpret_cast<const bf16x8 *>(R2l) + lane;
    // workgroup b owns the contiguous tiles [b T / G, (b + 1) T / G) (trunk_rr.hip: tile_begin): at the stock size 12 or 13 -- a round of
    // eight waves and a round of four or five, each alone on its SIMD
    const int64_t wt0 = uniform64(ntiles * (int64_t)blockIdx.x / (int64_t)gridDim.x), wt1 = uniform64(ntiles * ((int64_t)blockIdx.x + 1) / (int64_t)gridDim.x);
    for (int64_t r0 = wt0; r0 < wt1; r0 += kWaves) {
        const int64_t tile = r0 + wave;
        const bool live = tile < wt1;                     // wave-uniform: a wave without a tile still serves the pipeline (DMA share, barriers)
        const bool more = r0 + kWaves < wt1;              // workgroup-uniform
        const int64_t gp = tile * kRows + row;
        const bool ok = gp < n;
        const int64_t b = ok ? gp : 0;
        // the barrier that publishes chunk J (everybody's DMA share has landed) is also the point after which nobody reads the OTHER buffer
        // any more: the next chunk is requested into it at once and lands under this chunk's products.  Returns the LDS base of chunk J.
        // The wait: the vector-memory counter retires in issue order and counts this wave's activation stores too, so waiting for zero would
        // also wait out every store of the phase just finished; kAfterF[J] = the vector-memory instructions a LIVE wave issues for certain
        // between the request of chunk J and this point (the tile-packed stores of the epilogues in between), which may stay in flight.
        // The barrier is the bare instruction: __syncthreads() carries a workgroup fence, for which the compiler drains the counter anyway.
        auto chunk_begin = [&](auto jc, auto livec) -> char * {
            constexpr int J = decltype(jc)::value;
            if constexpr (!decltype(livec)::value) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (J == 0) { if (r0 == wt0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); }
            else if constexpr (J == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (J == 2 || J == 8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if constexpr (J == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (J + 1 < kChunks) dma_chunk(stream, lds2, kChunkOff[J + 1], kChunkBytes[J + 1], par ^ 1, wave, lane);
            else if (more) dma_chunk(stream, lds2, kChunkOff[0], kChunkBytes[0], par ^ 1, wave, lane);
            asm volatile("" ::: "memory");      // no store of the coming phase is scheduled ahead of the requests
            char *base = lds2 + par * kBufBytes;
            par ^= 1;
            return base;
        };
        if (!live) {      // nothing to compute: keep the chunk pipeline turning for the others
            static_for<kChunks>([&](auto jc) { (void)chunk_begin(jc, std::false_type{}); });
            continue;
        }
        // ---- this lane's inputs: 16 colour features (levels 8 h .. 8 h + 7) and its 48 encoding slots (rows past the end: zeros)
        uint32_t fcw[8], pew[24];
        {
            if (featc_words) {        // featc: uint32 [16, n], a level's two channels as bf16 (hs_hash_fwd, hsHashLayout::out_bf16): the words themselves
                const uint32_t *fl = reinterpret_cast<const uint32_t *>(featc) + (size_t)(8 * h) * n + b;
#pragma unroll
                for (int i = 0; i < 8; i++) fcw[i] = ok ? fl[(size_t)i * n] : 0u;
            } else {
                const float2 *fl = reinterpret_cast<const float2 *>(featc) + (size_t)(8 * h) * n + b;     // featc [16, n, 2]
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float2 t = fl[(size_t)i * n];
                    fcw[i] = ok ? pack2(t.x, t.y) : 0u;
                }
            }
            const float *src[3] = {points, dirs, normals};
            float raw[9];
#pragma unroll
            for (int t = 0; t < 3; t++) {          // one vector at a time, packed at once
                float v[12];
#pragma unroll
                for (int d = 0; d < 3; d++) raw[3 * t + d] = src[t][b * 3 + d];
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const float f = (float)(1 << (2 * h + kk));
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        float sn, cs;
                        __sincosf(raw[3 * t + d] * f, &sn, &cs);
                        v[6 * kk + d] = sn;
                        v[6 * kk + 3 + d] = cs;
                    }
                }
#pragma unroll
                for (int j = 0; j < 12; j += 2) pew[6 * t + (j >> 1)] = ok ? anchor(pack2(v[j], v[j + 1])) : 0u;
            }
#pragma unroll
            for (int j = 0; j < 12; j += 2) {      // slots 36..47: the nine raw coordinates (half 0), padding
                const float a0 = (j < 9 && h == 0) ? raw[j] : 0.f, a1 = (j + 1 < 9 && h == 0) ? raw[j + 1] : 0.f;
                pew[18 + (j >> 1)] = ok ? pack2(a0, a1) : 0u;
            }
            if (live) {
                static_for<2>([&](auto sc) { constexpr int s = decltype(sc)::value; tp_store_n<HS_NT_A2_FWD>(XAt, tile, XAS, s, lane, fcw + 4 * s); });
                static_for<6>([&](auto sc) { constexpr int s = decltype(sc)::value; tp_store_n<HS_NT_A2_FWD>(XAt, tile, XAS, 2 + s, lane, pew + 4 * s); });
            }
        }
        const uint32_t bias_b = lds_base(bias, 16 * h);
        const uint32_t okm = ok ? 0xffffffffu : 0u;      // rows past the end store zeros: the weight gradients sum whole tiles
        uint32_t ha[64], hb[64];            // hc in ha -> fv in hb -> r0 in ha -> r1 in hb
        uint32_t mk[4];
        f32x16 acc[2][2];
        // epilogue of a finished QUARTER qd (tiles 2 qd, 2 qd + 1): slices 0..15 activation + pack of one register pair, 16..19 its four k-steps leave
        auto epi = [&](auto slc, auto reluc, const f32x16 (&src)[2], uint32_t *hp, auto qdc, uint16_t *T) {
            constexpr int sl = decltype(slc)::value, qd = decltype(qdc)::value;
            constexpr bool relu = decltype(reluc)::value;
            if constexpr (sl < 16) {
                constexpr int j = sl >> 3, r = sl & 7, nd = 2 * qd + j;
                if constexpr (relu) hp[8 * nd + r] = relu_pair(src[j][2 * r], src[j][2 * r + 1], mk[nd >> 1], 8 * (nd & 1) + r) & okm;
                else hp[8 * nd + r] = anchor(pack2(src[j][2 * r], src[j][2 * r + 1])) & okm;
            } else {
                constexpr int ks = 4 * qd + (sl - 16);
                if (live) tp_store_n<HS_NT_A2_FWD>(T, tile, HS, ks, lane, hp + 4 * ks);
            }
        };
        auto store_mask = [&](int layer) {
            uint32_t lo = (uint32_t)lane * 4u;
            asm volatile("" : "+v"(lo));
            if (live) *reinterpret_cast<uint4 *>(masks + ((size_t)tile * 3 + layer) * 256 + lo) = make_uint4(mk[0], mk[1], mk[2], mk[3]);
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        // ---- colour MLP layer 0: 32 -> 256, ReLU (chunk 0: all eight tiles, two k-steps each)
        mk[0] = mk[1] = mk[2] = mk[3] = 0u;
        {
            char *cb = chunk_begin(std::integral_constant<int, 0>{}, std::true_type{});
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                init_acc_b(acc[q & 1][0], relaunder(bias_b), 32 * (2 * q));
                init_acc_b(acc[q & 1][1], relaunder(bias_b), 32 * (2 * q + 1));
                const uint32_t ab = lds_base(cb + 2 * q * kC0Tile, lane * 16);
                auto af = [&](int s, int j) { return lds_at<bf16x8>(ab, j * kC0Tile + s * 1024); };
                auto bf = [&](int s) { return frag_of(fcw + 4 * s); };
                if constexpr (q == 0) phase2g<2, 2, 2, 20, false>(acc[0], af, bf, [](auto) {});
                else phase2g<2, 2, 2, 20, true>(acc[q & 1], af, bf, [&](auto slc) { epi(slc, T_{}, acc[(q & 1) ^ 1], ha, std::integral_constant<int, q - 1>{}, HCt); });
            });
        }
        // ---- colour MLP layer 1: 256 -> 256, linear (chunks 1, 2); its quarter 0 finishes layer 0's quarter 3 (k-steps 12..15 of this product)
        {
            char *cb = nullptr;
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q % 2 == 0) cb = chunk_begin(std::integral_constant<int, 1 + q / 2>{}, std::true_type{});
                init_acc_b(acc[q & 1][0], relaunder(bias_b), 256 + 32 * (2 * q));
                init_acc_b(acc[q & 1][1], relaunder(bias_b), 256 + 32 * (2 * q + 1));
                const uint32_t ab = lds_base(cb + 2 * (q % 2) * kBigTile, lane * 16);
                auto af = [&](int s, int j) { return lds_at<bf16x8>(ab, j * kBigTile + s * 1024); };
                auto bf = [&](int s) { return frag_of(ha + 4 * s); };
                if constexpr (q == 0) {
                    phase2g<HS, kLA, 10, 20, true>(acc[0], af, bf, [&](auto slc) { epi(slc, T_{}, acc[1], ha, std::integral_constant<int, 3>{}, HCt); });
                    store_mask(0);
                } else {
                    phase2g<HS, kLA, HS, 20, true>(acc[q & 1], af, bf, [&](auto slc) { epi(slc, F_{}, acc[(q & 1) ^ 1], hb, std::integral_constant<int, q - 1>{}, FVt); });
                }
            });
        }
        // ---- rendering layer 0: [encodings (6 k-steps) | fv (16 k-steps)] -> 256, ReLU (chunks 3..6: one quarter each); the encodings first, so
        //      that quarter 0 can finish fv's quarter 3 under them
        mk[0] = mk[1] = mk[2] = mk[3] = 0u;
        static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            char *cb = chunk_begin(std::integral_constant<int, 3 + q>{}, std::true_type{});
            init_acc_b(acc[q & 1][0], relaunder(bias_b), 512 + 32 * (2 * q));
            init_acc_b(acc[q & 1][1], relaunder(bias_b), 512 + 32 * (2 * q + 1));
            const uint32_t ab = lds_base(cb, lane * 16);
            auto af = [&](int s, int j) { return lds_at<bf16x8>(ab, j * kR0Tile + s * 1024); };
            auto bf = [&](int s) { return s < 6 ? frag_of(pew + 4 * s) : frag_of(hb + 4 * (s - 6)); };
            if constexpr (q == 0) phase2g<22, kLA, 16, 20, true>(acc[0], af, bf, [&](auto slc) { epi(slc, F_{}, acc[1], hb, std::integral_constant<int, 3>{}, FVt); });
            else phase2g<22, kLA, 22, 20, true>(acc[q & 1], af, bf, [&](auto slc) { epi(slc, T_{}, acc[(q & 1) ^ 1], ha, std::integral_constant<int, q - 1>{}, R0t); });
        });
        // ---- rendering layer 1: 256 -> 256, ReLU (chunks 7, 8)
        {
            char *cb = nullptr;
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q % 2 == 0) cb = chunk_begin(std::integral_constant<int, 7 + q / 2>{}, std::true_type{});
                init_acc_b(acc[q & 1][0], relaunder(bias_b), 768 + 32 * (2 * q));
                init_acc_b(acc[q & 1][1], relaunder(bias_b), 768 + 32 * (2 * q + 1));
                const uint32_t ab = lds_base(cb + 2 * (q % 2) * kBigTile, lane * 16);
                auto af = [&](int s, int j) { return lds_at<bf16x8>(ab, j * kBigTile + s * 1024); };
                auto bf = [&](int s) { return frag_of(ha + 4 * s); };
                if constexpr (q == 0) {
                    phase2g<HS, kLA, 10, 20, true>(acc[0], af, bf, [&](auto slc) { epi(slc, T_{}, acc[1], ha, std::integral_constant<int, 3>{}, R0t); });
                    store_mask(1);
                    mk[0] = mk[1] = mk[2] = mk[3] = 0u;
                } else {
                    phase2g<HS, kLA, HS, 20, true>(acc[q & 1], af, bf, [&](auto slc) { epi(slc, T_{}, acc[(q & 1) ^ 1], hb, std::integral_constant<int, q - 1>{}, R1t); });
                }
            });
        }
        // ---- rendering layer 2: 256 -> 3 on two partial accumulators (weights resident), layer 1's quarter 3 in the shadow of its first k-steps; sigmoid
        f32x16 y0, y1;
        {
            auto f2 = [&](int s, int j) { return R2v[(size_t)(2 * s + j) * 64]; };
            bf16x8 ring[3][2];
            static_for<2>([&](auto sc) { constexpr int s = decltype(sc)::value; ring[s][0] = f2(s, 0); ring[s][1] = f2(s, 1); });
            static_for<HS / 2>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 2 < HS / 2) { ring[(s + 2) % 3][0] = f2(s + 2, 0); ring[(s + 2) % 3][1] = f2(s + 2, 1); }
                if constexpr (s < 5) static_for<4>([&](auto jc) { epi(std::integral_constant<int, 4 * s + decltype(jc)::value>{}, T_{}, acc[1], hb, std::integral_constant<int, 3>{}, R1t); });
                // (k-steps 2 s, 2 s + 1 <= 11 read quarters 0..2 only while quarter 3 is being finished)
                if constexpr (s == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][0], frag_of(hb), zero, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0][1], frag_of(hb + 4), zero, 0, 0, 0);
                } else {
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % 3][0], frag_of(hb + 4 * (2 * s)), y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % 3][1], frag_of(hb + 4 * (2 * s + 1)), y1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        store_mask(2);
        if (h == 0 && ok) {       // neurons 0..2 = registers 0..2 of lane half 0
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float yv = y0[c] + y1[c] + bias[1024 + c];
                rgb[gp * 3 + c] = 1.f / (1.f + __expf(-yv));
            }
        }
    }
}

// ================================================================================================================ backward
//   y~ = rgb~ rgb (1 - rgb);  r1~ = (Wr2^T y~) . [r1 > 0];  r0~ = (Wr1^T r1~) . [r0 > 0];  fv~ = Wr0f^T r0~,  enc~(normal) = Wr0n^T r0~;
//   hc~ = (Wc1^T fv~) . [hc > 0];  featc~ = Wc0^T hc~;   d normals = (d enc / d normal)^T enc~(normal)
// The four 256-wide cotangents leave tile-packed (they are the A operands of the weight gradients), y~ row-major [n, 32]; the ReLU signs
// come from the forward pass's masks; nothing else of the forward pass is read.
// STORE = false: ablation for DESIGN 14.1 only (the four tile-packed cotangents are not written: what the kernel would cost if its weight
// gradients never left the chip); results are then incomplete on purpose (-DHS_A2_NOSTORE=1 in a variant build, tools/exp/dw_fusion_bounds.py)
template <bool STORE>
__global__ __launch_bounds__(kThreadsW, 2) void k_appear2_bwd(const float *__restrict__ g_rgb, const float *__restrict__ rgb, const float *__restrict__ normals,
                                                               const uint32_t *__restrict__ masks, const char *__restrict__ streamT,
                                                               uint16_t *__restrict__ gy_out, uint16_t *__restrict__ GR1t, uint16_t *__restrict__ GR0t,
                                                               uint16_t *__restrict__ GFVt, uint16_t *__restrict__ GHCt, float *__restrict__ d_normals,
                                                               float *__restrict__ g_featc, float *__restrict__ gb2, int64_t n, int normals_add) {
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), row = lane & 31, h = lane >> 5;
    int par = 0;
    dma_chunk(streamT, lds2, kChunkTOff[0], kChunkTBytes[0], 0, wave, lane);
    const int64_t ntiles = (n + kRows - 1) / kRows;
    const int64_t wt0 = uniform64(ntiles * (int64_t)blockIdx.x / (int64_t)gridDim.x), wt1 = uniform64(ntiles * ((int64_t)blockIdx.x + 1) / (int64_t)gridDim.x);
    for (int64_t r0 = wt0; r0 < wt1; r0 += kWaves) {
        const int64_t tile = r0 + wave;
        const bool live = tile < wt1;
        const bool more = r0 + kWaves < wt1;
        const int64_t gp = tile * kRows + row;
        const bool ok = gp < n;
        const int64_t b = ok ? gp : 0;
        // (counted wait + bare barrier, as in the forward kernel: the tile-packed stores issued since chunk J's request may stay in flight)
        auto chunk_begin = [&](auto jc, auto livec) -> char * {
            constexpr int J = decltype(jc)::value;
            if constexpr (!decltype(livec)::value || !STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (J == 0) { if (r0 == wt0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); }
            else if constexpr (J == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (J == 6 || J == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (J + 1 < kChunksT) dma_chunk(streamT, lds2, kChunkTOff[J + 1], kChunkTBytes[J + 1], par ^ 1, wave, lane);
            else if (more) dma_chunk(streamT, lds2, kChunkTOff[0], kChunkTBytes[0], par ^ 1, wave, lane);
            asm volatile("" ::: "memory");
            char *base = lds2 + par * kBufBytes;
            par ^= 1;
            return base;
        };
        if (!live) {
            static_for<kChunksT>([&](auto jc) { (void)chunk_begin(jc, std::false_type{}); });
            continue;
        }
        // (the normal itself, for d normals below: requested here, ahead of the tile's stores -- a load issued behind them waits for them)
        const float nx[3] = {normals[b * 3], normals[b * 3 + 1], normals[b * 3 + 2]};
        // ---- the ReLU signs of this lane's 3 x 128 activations, and the cotangent of the three pre-sigmoid outputs (lane half 0 holds k = 0..3)
        uint32_t mk[3][4];
        {
            uint32_t lo = (uint32_t)lane * 4u;
            asm volatile("" : "+v"(lo));
#pragma unroll
            for (int l = 0; l < 3; l++) {
                uint4 m4 = make_uint4(0u, 0u, 0u, 0u);
                if (live) m4 = *reinterpret_cast<const uint4 *>(masks + ((size_t)tile * 3 + l) * 256 + lo);
                mk[l][0] = m4.x; mk[l][1] = m4.y; mk[l][2] = m4.z; mk[l][3] = m4.w;
            }
        }
        uint32_t gyw[4] = {0u, 0u, 0u, 0u};
        {
            float gy3[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float o = rgb[b * 3 + c];
                gy3[c] = (ok && h == 0) ? g_rgb[b * 3 + c] * o * (1.f - o) : 0.f;
            }
            gyw[0] = pack2(gy3[0], gy3[1]);
            gyw[1] = pack2(gy3[2], 0.f);
            if (ok) {      // y~ row-major [n, 32] bf16 (the A operand of dW_R2): half 0 writes columns 0..15, half 1 the zeros of 16..31
                uint4 *dst = reinterpret_cast<uint4 *>(gy_out + gp * 32 + 16 * h);
                dst[0] = make_uint4(gyw[0], gyw[1], 0u, 0u);
                dst[1] = make_uint4(0u, 0u, 0u, 0u);
            }
            if (gb2 != nullptr) {      // d b_R2: the tile's three row sums as a per-tile partial [tiles, 4] (summed by the caller: 9 408 atomics on three
                                       // addresses queue in one L2 channel, and every wave would wait for its own at the next chunk boundary)
                float sums[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float v = gy3[c];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                    sums[c] = v;
                }
                if (lane == 0 && live) *reinterpret_cast<float4 *>(gb2 + tile * 4) = make_float4(sums[0], sums[1], sums[2], 0.f);
            }
        }
        uint32_t ha[64], hb[64];            // r1~ in ha -> r0~ in hb -> fv~ in ha -> hc~ in hb
        f32x16 acc[2][2];
        auto zero2 = [&](f32x16 (&a)[2]) {
#pragma unroll
            for (int i = 0; i < 16; i++) { a[0][i] = 0.f; a[1][i] = 0.f; }
        };
        // epilogue of a finished quarter qd: slices 0..15 mask + pack of one register pair, 16..19 its four k-steps leave tile-packed.
        // LAYER >= 0: the cotangent passes where the forward's ReLU output was positive (mask bit SET = unit off, appearance2 forward)
        auto epi = [&](auto slc, auto layerc, const f32x16 (&src)[2], uint32_t *hp, auto qdc, uint16_t *T) {
            constexpr int sl = decltype(slc)::value, qd = decltype(qdc)::value, LAYER = decltype(layerc)::value;
            if constexpr (sl < 16) {
                constexpr int j = sl >> 3, r = sl & 7, nd = 2 * qd + j;
                uint32_t p = pack2(src[j][2 * r], src[j][2 * r + 1]);
                if constexpr (LAYER >= 0) {
                    const uint32_t off = (mk[LAYER][nd >> 1] >> (8 * (nd & 1) + r)) & 0x00010001u;
                    p &= (off ^ 0x00010001u) * 0xffffu;
                }
                hp[8 * nd + r] = anchor(p);
            } else {
                constexpr int ks = 4 * qd + (sl - 16);
                if (STORE && live) tp_store_n<HS_NT_A2_BWD>(T, tile, HS, ks, lane, hp + 4 * ks);
            }
        };
        using L2_ = std::integral_constant<int, 2>;
        using L1_ = std::integral_constant<int, 1>;
        using L0_ = std::integral_constant<int, 0>;
        using LN_ = std::integral_constant<int, -1>;
        // ---- r1~ = (Wr2^T y~) . [r1 > 0]: one k-step per quarter (chunk 0)
        {
            char *cb = chunk_begin(std::integral_constant<int, 0>{}, std::true_type{});
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                zero2(acc[q & 1]);
                const uint32_t ab = lds_base(cb + 2 * q * kT2Tile, lane * 16);
                auto af = [&](int s, int j) { return lds_at<bf16x8>(ab, j * kT2Tile + s * 1024); };
                auto bf = [&](int) { return frag_of(gyw); };
                if constexpr (q == 0) phase2g<1, 1, 1, 20, false>(acc[0], af, bf, [](auto) {});
                else phase2g<1, 1, 1, 20, true>(acc[q & 1], af, bf, [&](auto slc) { epi(slc, L2_{}, acc[(q & 1) ^ 1], ha, std::integral_constant<int, q - 1>{}, GR1t); });
            });
        }
        // ---- r0~ = (Wr1^T r1~) . [r0 > 0] (chunks 1, 2)
        {
            char *cb = nullptr;
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q % 2 == 0) cb = chunk_begin(std::integral_constant<int, 1 + q / 2>{}, std::true_type{});
                zero2(acc[q & 1]);
                const uint32_t ab = lds_base(cb + 2 * (q % 2) * kBigTile, lane * 16);
                auto af = [&](int s, int j) { return lds_at<bf16x8>(ab, j * kBigTile + s * 1024); };
                auto bf = [&](int s) { return frag_of(ha + 4 * s); };
                if constexpr (q == 0) phase2g<HS, kLA, 10, 20, true>(acc[0], af, bf, [&](auto slc) { epi(slc, L2_{}, acc[1], ha, std::integral_constant<int, 3>{}, GR1t); });
                else phase2g<HS, kLA, HS, 20, true>(acc[q & 1], af, bf, [&](auto slc) { epi(slc, L1_{}, acc[(q & 1) ^ 1], hb, std::integral_constant<int, q - 1>{}, GR0t); });
            });
        }
        // ---- fv~ = Wr0f^T r0~ (chunks 3, 4), then the encoded-normal rows (chunk 5, one tile) with fv~'s last quarter in their shadow
        {
            char *cb = nullptr;
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q % 2 == 0) cb = chunk_begin(std::integral_constant<int, 3 + q / 2>{}, std::true_type{});
                zero2(acc[q & 1]);
                const uint32_t ab = lds_base(cb + 2 * (q % 2) * kBigTile, lane * 16);
                auto af = [&](int s, int j) { return lds_at<bf16x8>(ab, j * kBigTile + s * 1024); };
                auto bf = [&](int s) { return frag_of(hb + 4 * s); };
                if constexpr (q == 0) phase2g<HS, kLA, 10, 20, true>(acc[0], af, bf, [&](auto slc) { epi(slc, L1_{}, acc[1], hb, std::integral_constant<int, 3>{}, GR0t); });
                else phase2g<HS, kLA, HS, 20, true>(acc[q & 1], af, bf, [&](auto slc) { epi(slc, LN_{}, acc[(q & 1) ^ 1], ha, std::integral_constant<int, q - 1>{}, GFVt); });
            });
        }
        f32x16 en;       // enc~ of the 27 encoded-normal inputs: register r <-> input 8 (r >> 2) + 4 h + (r & 3)
        {
            char *cb = chunk_begin(std::integral_constant<int, 5>{}, std::true_type{});
#pragma unroll
            for (int i = 0; i < 16; i++) en[i] = 0.f;
            const uint32_t ab = lds_base(cb, lane * 16);
            bf16x8 ring[kLA + 1];
            static_for<kLA>([&](auto sc) { constexpr int s = decltype(sc)::value; ring[s] = lds_at<bf16x8>(ab, s * 1024); });
            static_for<HS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + kLA < HS) ring[(s + kLA) % (kLA + 1)] = lds_at<bf16x8>(ab, (s + kLA) * 1024);
                en = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (kLA + 1)], frag_of(hb + 4 * s), en, 0, 0, 0);
                // fv~'s quarter 3 (acc[1]) leaves in the shadow: 20 slices over the 16 k-steps
                constexpr int lo = (s * 20) / HS, hi = ((s + 1) * 20) / HS;
                static_for<hi - lo>([&](auto jc) { epi(std::integral_constant<int, lo + decltype(jc)::value>{}, LN_{}, acc[1], ha, std::integral_constant<int, 3>{}, GFVt); });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- hc~ = (Wc1^T fv~) . [hc > 0] (chunks 6, 7); d normals from enc~ in the shadow of the first quarter
        {
            char *cb = nullptr;
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q % 2 == 0) cb = chunk_begin(std::integral_constant<int, 6 + q / 2>{}, std::true_type{});
                zero2(acc[q & 1]);
                const uint32_t ab = lds_base(cb + 2 * (q % 2) * kBigTile, lane * 16);
                auto af = [&](int s, int j) { return lds_at<bf16x8>(ab, j * kBigTile + s * 1024); };
                auto bf = [&](int s) { return frag_of(ha + 4 * s); };
                if constexpr (q == 0) phase2g<HS, kLA, HS, 20, false>(acc[0], af, bf, [](auto) {});
                else phase2g<HS, kLA, HS, 20, true>(acc[q & 1], af, bf, [&](auto slc) { epi(slc, L0_{}, acc[(q & 1) ^ 1], hb, std::integral_constant<int, q - 1>{}, GHCt); });
            });
        }
        {   // d normals[d] = sum over the normal's 27 encoded inputs c of (d enc_c / d n_d) enc~_c: c < 3 the coordinate itself, c = 3 + 6 k + comp:
            // sin (comp < 3) / cos (comp >= 3) of 2^k n_(comp % 3)
            float dn[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r++) {
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    const int c = 8 * (r >> 2) + 4 * hh + (r & 3);
                    if (c >= 27) continue;
                    float w;
                    int d;
                    if (c < 3) { d = c; w = 1.f; }
                    else {
                        const int k = (c - 3) / 6, comp = (c - 3) % 6;
                        d = comp % 3;
                        const float f = (float)(1 << k);
                        w = comp < 3 ? f * __cosf(f * nx[d]) : -f * __sinf(f * nx[d]);
                    }
                    if (h == hh) dn[d] += w * en[r];
                }
            }
#pragma unroll
            for (int d = 0; d < 3; d++) dn[d] += __shfl_xor(dn[d], 32);
            if (h == 0 && ok) {
                if (normals_add) { dn[0] += d_normals[gp * 3]; dn[1] += d_normals[gp * 3 + 1]; dn[2] += d_normals[gp * 3 + 2]; }    // the other consumer's cotangent is there already
                d_normals[gp * 3] = dn[0]; d_normals[gp * 3 + 1] = dn[1]; d_normals[gp * 3 + 2] = dn[2];
            }
        }
        // ---- featc~ = Wc0^T hc~ (chunk 8, one tile): hc~'s last quarter in the shadow of its first k-steps; register r <-> colour feature 16 h + r
        {
            char *cb = chunk_begin(std::integral_constant<int, 8>{}, std::true_type{});
            f32x16 fc;
#pragma unroll
            for (int i = 0; i < 16; i++) fc[i] = 0.f;
            const uint32_t ab = lds_base(cb, lane * 16);
            bf16x8 ring[kLA + 1];
            static_for<kLA>([&](auto sc) { constexpr int s = decltype(sc)::value; ring[s] = lds_at<bf16x8>(ab, s * 1024); });
            static_for<HS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + kLA < HS) ring[(s + kLA) % (kLA + 1)] = lds_at<bf16x8>(ab, (s + kLA) * 1024);
                if constexpr (s < 10) static_for<2>([&](auto jc) { epi(std::integral_constant<int, 2 * s + decltype(jc)::value>{}, L0_{}, acc[1], hb, std::integral_constant<int, 3>{}, GHCt); });
                fc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % (kLA + 1)], frag_of(hb + 4 * s), fc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (ok) {       // g_featc [16, n, 2]: level 8 h + i, channels (2 i, 2 i + 1)
                float2 *dst = reinterpret_cast<float2 *>(g_featc) + (size_t)(8 * h) * n + gp;
#pragma unroll
                for (int i = 0; i < 8; i++) dst[(size_t)i * n] = make_float2(fc[2 * i], fc[2 * i + 1]);
            }
        }
    }
}

}  // namespace

extern "C" {

int64_t hs_appearance2_pack_bytes(int32_t which) {
    switch (which) {
        case 0: return (int64_t)kStreamBytes;          /* the streamed fragment image (W_C0, W_C1, W_R0, W_R1) */
        case 1: return (int64_t)kW2F * 2;              /* W_R2's fragments */
        case 2: return (int64_t)kA2Bias * 4;           /* bias block */
        default: return -1;
    }
}

int hs_appearance2_enc_column(int32_t h, int32_t j) { return (h < 0 || h > 1 || j < 0 || j >= 48) ? -1 : enc_column(h, j); }

int hs_appearance2_pack(const float *Wc0, const float *Wc1, const float *Wr0, int32_t ldr0, const float *Wr1, const float *Wr2, const float *bc0,
                        const float *bc1, const float *br0, const float *br1, const float *br2, void *stream_image, void *R2f, float *bias,
                        void *streamT_image, void *stream) {
    if (ldr0 < 337) return HS_ERR_ARG;
    if (!Wc0 || !Wc1 || !Wr0 || !Wr1 || !Wr2 || !bc0 || !bc1 || !br0 || !br1 || !br2 || !stream_image || !R2f || !bias) return HS_ERR_NULL;
    const int slots = kPackSlots + (streamT_image ? kStreamTBytes / 16 : 0);
    k_appear2_pack<<<(slots + 255) / 256, 256, 0, (hipStream_t)stream>>>(Wc0, Wc1, Wr0, ldr0, Wr1, Wr2, bc0, bc1, br0, br1, br2, (uint16_t *)stream_image,
                                                                       (uint16_t *)R2f, bias, (uint16_t *)streamT_image);
    return wt_check_launch();
}

int hs_pack_iteration(const float *W0, int32_t ld0, int32_t f_in, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2,
                      int32_t d_out, void *sW0f, void *sW1f, void *sW2f, float *sbias, void *W0f, void *W1f, void *W2f, float *bias, void *W1Tf, void *W0Tf,
                      void *W2Tf, float *W2tab, void *w1t, void *w2t, void *w0t, const float *Wc0, const float *Wc1, const float *Wr0, int32_t ldr0,
                      const float *Wr1, const float *Wr2, const float *bc0, const float *bc1, const float *br0, const float *br1, const float *br2,
                      void *stream_image, void *R2f, float *abias, void *streamT_image, void *stream) {
    const bool trunk = sW0f || W0f, colour = stream_image != nullptr;
    if (!trunk && !colour) return HS_OK;
    if (trunk) {
        if (d_out < 1 || d_out > 32 || ld0 < 71) return HS_ERR_ARG;
        if (!W0 || !b0 || !W1 || !b1 || !W2 || !b2) return HS_ERR_NULL;
        if (sW0f && (!sW1f || !sW2f || !sbias)) return HS_ERR_NULL;
        if (W0f && (!W1f || !W2f || !bias || !W1Tf || !W0Tf || !W2Tf || !W2tab)) return HS_ERR_NULL;
        if (w1t && (!W0f || !w2t || !w0t)) return HS_ERR_NULL;
    }
    if (colour) {
        if (ldr0 < 337) return HS_ERR_ARG;
        if (!Wc0 || !Wc1 || !Wr0 || !Wr1 || !Wr2 || !bc0 || !bc1 || !br0 || !br1 || !br2 || !R2f || !abias) return HS_ERR_NULL;
    }
    PackIterArgs a;
    a.W0 = W0; a.b0 = b0; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.ld0 = ld0; a.f_in = f_in; a.d_out = d_out;
    a.sW0f = (uint16_t *)sW0f; a.sW1f = (uint16_t *)sW1f; a.sW2f = (uint16_t *)sW2f; a.sbias = sbias;
    a.W0f = (uint16_t *)W0f; a.W1f = (uint16_t *)W1f; a.W2f = (uint16_t *)W2f; a.bias = bias;
    a.W1Tf = (uint16_t *)W1Tf; a.W0Tf = (uint16_t *)W0Tf; a.W2Tf = (uint16_t *)W2Tf; a.W2tab = W2tab;
    a.w1t = (uint16_t *)w1t; a.w2t = (uint16_t *)w2t; a.w0t = (uint16_t *)w0t;
    a.Wc0 = Wc0; a.Wc1 = Wc1; a.Wr0 = Wr0; a.Wr1 = Wr1; a.Wr2 = Wr2; a.bc0 = bc0; a.bc1 = bc1; a.br0 = br0; a.br1 = br1; a.br2 = br2; a.ldr0 = ldr0;
    a.stream = (uint16_t *)stream_image; a.R2f = (uint16_t *)R2f; a.abias = abias; a.streamT = (uint16_t *)streamT_image;
    k_pack_iteration<<<(kIterPackSlots + 255) / 256, 256, 0, (hipStream_t)stream>>>(a);
    return wt_check_launch();
}

int hs_appearance2_fwd(const float *featc, const float *points, const float *dirs, const float *normals, const void *stream_image, const void *R2f,
                       const float *bias, void *XAt, void *HCt, void *FVt, void *R0t, void *R1t, uint32_t *masks, float *rgb, int64_t n, int32_t featc_words,
                       void *stream) {
    if (n < 0) return HS_ERR_ARG;
    if (n == 0) return HS_OK;
    if (!featc || !points || !dirs || !normals || !stream_image || !R2f || !bias || !XAt || !HCt || !FVt || !R0t || !R1t || !masks || !rgb) return HS_ERR_NULL;
    const size_t lds = 2 * (size_t)kBufBytes + (size_t)kW2F * 2 + kA2Bias * sizeof(float);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_appear2_fwd, (int)lds);
    const int64_t ntiles = (n + kRows - 1) / kRows, want = (ntiles + kWaves - 1) / kWaves;
    k_appear2_fwd<<<(int)(want < 256 ? want : 256), kThreadsW, lds, (hipStream_t)stream>>>(
        featc, points, dirs, normals, (const char *)stream_image, (const uint16_t *)R2f, bias, (uint16_t *)XAt, (uint16_t *)HCt, (uint16_t *)FVt, (uint16_t *)R0t,
        (uint16_t *)R1t, masks, rgb, n, featc_words);
    return wt_check_launch();
}

int64_t hs_appearance2_pack_t_bytes(void) { return (int64_t)kStreamTBytes; }

int hs_appearance2_bwd(const float *g_rgb, const float *rgb, const float *normals, const uint32_t *masks, const void *streamT_image, void *gy, void *GR1t,
                       void *GR0t, void *GFVt, void *GHCt, float *d_normals, float *g_featc, float *gb2, int64_t n, int32_t normals_add, void *stream) {
    if (n < 0) return HS_ERR_ARG;
    if (n == 0) return HS_OK;
    if (!g_rgb || !rgb || !normals || !masks || !streamT_image || !gy || !GR1t || !GR0t || !GFVt || !GHCt || !d_normals || !g_featc) return HS_ERR_NULL;
    const size_t lds = 2 * (size_t)kBufBytes;
    static hsLdsAttrOnce attr;
    constexpr bool nostore = HS_A2_NOSTORE;      /* ablation (-DHS_A2_NOSTORE=1 in a variant build): the backward without its cotangent stores */
    const int64_t ntiles = (n + kRows - 1) / kRows, want = (ntiles + kWaves - 1) / kWaves;
    if (nostore) {
        static hsLdsAttrOnce attr0;
        attr0.set((const void *)k_appear2_bwd<false>, (int)lds);
        k_appear2_bwd<false><<<(int)(want < 256 ? want : 256), kThreadsW, lds, (hipStream_t)stream>>>(
            g_rgb, rgb, normals, masks, (const char *)streamT_image, (uint16_t *)gy, (uint16_t *)GR1t, (uint16_t *)GR0t, (uint16_t *)GFVt, (uint16_t *)GHCt,
            d_normals, g_featc, gb2, n, normals_add);
        return wt_check_launch();
    }
    attr.set((const void *)k_appear2_bwd<true>, (int)lds);
    k_appear2_bwd<true><<<(int)(want < 256 ? want : 256), kThreadsW, lds, (hipStream_t)stream>>>(
        g_rgb, rgb, normals, masks, (const char *)streamT_image, (uint16_t *)gy, (uint16_t *)GR1t, (uint16_t *)GR0t, (uint16_t *)GFVt, (uint16_t *)GHCt, d_normals,
        g_featc, gb2, n, normals_add);
    return wt_check_launch();
}

}  // extern "C"
