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
#include "launch_util.h"
#include <stdlib.h>
#include "wave_tile.h"

#ifdef HS_SDF2_PROFILE     // tools/exp/sdf2_prof.hip: per-phase s_memtime stamps of a steady-state wave tile
__device__ unsigned long long g_sdf2_prof[256 * 8 * 16];   // [block][wave][stamp]
#endif

#ifdef HS_SWEEP_PROFILE    // tools/exp/sweep_prof.hip: s_memtime stamps inside the gather of k_sdf_mlp2<., true> (with HS_SDF2_PROFILE)
__device__ unsigned long long g_sweep_prof[256 * 8 * 8];    // [block][wave][first / second tile][reads issued, posenc done, blended, -]
#define HS_GSTAMP(i) do { if ((threadIdx.x & 63) == 0) g_sweep_prof[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + prof_slot + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HS_GSTAMP(i) do { } while (0)
#endif

namespace {

// log2(1 + 2^t); above t = 30 that IS t in fp32 (sdf_mlp.hip: softplus_scaled).  No clamp of the exponential's argument: beyond
// t = 128 it overflows to +inf, the logarithm returns +inf, and the median below never returns it.
__device__ __forceinline__ float softplus_scaled2(float t) {
    const float l = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(t));
    // t > 30 ? t : l as ONE instruction: l >= t always, and l <= 30 exactly when t <= 30, so the wanted value is the median of
    // (l, t, 30) -- also when l overflowed to +inf.  (The kernel is bound by its VALU issue slots: 5.5 -> 4.5 per activation.)
    return __builtin_amdgcn_fmed3f(l, t, 30.f);
}

// softplus + bf16 pack of accumulator registers r, r + 1 (r even) of a tile: half a register pair of the next layer's B fragment.
// Register r of tile nt lands in word (r >> 1) of that tile's 8-word block = k-steps 2 nt (words 0-3) and 2 nt + 1 (words 4-7).
__device__ __forceinline__ uint32_t epilogue_pair(const f32x16 &acc, int r) {
    uint32_t w = pack2(softplus_scaled2(acc[r]), softplus_scaled2(acc[r + 1]));
    return anchor(w);      // keeps the softplus in the MFMA shadow it was placed in (wave_tile.h)
}

// ---- the in-kernel gather of the GATHER instantiations (see GatherLevel below): eight levels of one point on one lane
struct GatherLevel;
__device__ __forceinline__ float gw_smoothstep(float t) { return t * t * (3.0f - 2.0f * t); }     // hash_encode.hip: smoothstep
// `between()` -- the trunk's own input arithmetic (eighteen sines and cosines) -- runs while the 64 reads are in flight; it is written into both arms of
// the inside / outside branch so that the reads' destination registers never cross a join (as two functions around the call they went to scratch).
template <class GL, class F>
__device__ __forceinline__ void gather_words(uint32_t (&fw)[8], const float *__restrict__ x01, const char *__restrict__ table, const GL *lv, int64_t gp, bool ok,
                                             F &&between, [[maybe_unused]] int prof_slot = 0) {
    const float p0 = ok ? x01[gp * 3] : 2.f, p1 = ok ? x01[gp * 3 + 1] : 2.f, p2 = ok ? x01[gp * 3 + 2] : 2.f;
    const bool inside = !(p0 < 0.f || p0 > 1.f) && !(p1 < 0.f || p1 > 1.f) && !(p2 < 0.f || p2 > 1.f);     // hash_encode.hip: locate
#pragma unroll
    for (int i = 0; i < 8; i++) fw[i] = 0u;
    if (!inside) {                               // (points outside the unit cube: zero features, hashencoder.cu:124-149 -- and no reads)
        between();
        return;
    }
    float2 e[8][8];
    float w[8][3];
    // phase A: locate, index, request -- all eight levels' reads are in flight before the first is used
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint4 la = *reinterpret_cast<const uint4 *>(&lv[i]), lb = *(reinterpret_cast<const uint4 *>(&lv[i]) + 1);
        const float scale = __uint_as_float(la.x);
        const uint32_t A = la.y, Bm = la.z, mask = la.w, limit = lb.x, byte_off = lb.y, flags = lb.z;
        uint32_t g[3];
        const float ps[3] = {p0, p1, p2};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float pos = ps[d] * scale;
            const float fl = floorf(pos);
            g[d] = (uint32_t)fl;
            pos -= (float)g[d];
            w[i][d] = gw_smoothstep(pos);
        }
        const uint32_t a0 = g[1] * A, a1 = a0 + A, b0 = g[2] * Bm, b1 = b0 + Bm, x0 = g[0], x1 = g[0] + 1u;
        uint32_t idx[8];
        if (flags & 1u) {
#pragma unroll
            for (int c = 0; c < 8; c++) idx[c] = ((c & 1) ? x1 : x0) ^ ((c & 2) ? a1 : a0) ^ ((c & 4) ? b1 : b0);
        } else {
#pragma unroll
            for (int c = 0; c < 8; c++) idx[c] = ((c & 1) ? x1 : x0) + ((c & 2) ? a1 : a0) + ((c & 4) ? b1 : b0);
        }
        uint32_t top = 0u;
#pragma unroll
        for (int c = 0; c < 8; c++) { idx[c] &= mask; top = max(top, idx[c]); }
        if (top >= limit) {       // rare: a table that is not a power of two, and a corner past its end (x = 1 on an integer scale): the reference's modulo
            const uint32_t t = lb.w;
#pragma unroll
            for (int c = 0; c < 8; c++) idx[c] = idx[c] >= t ? idx[c] % t : idx[c];
        }
        const char *base = table + byte_off;
#pragma unroll
        for (int c = 0; c < 8; c++) e[i][c] = *reinterpret_cast<const float2 *>(base + ((size_t)idx[c] << 3));
    }
    HS_GSTAMP(0);
    between();
    HS_GSTAMP(1);
    // phase B: blend in the reference's order (k_hash_fwd / k_hash_fwd_pair: corners 0..7, x bit fastest; weight = ((1 * wx) * wy) * wz)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            float wt = 1.f;
            wt *= (c & 1) ? w[i][0] : 1 - w[i][0];
            wt *= (c & 2) ? w[i][1] : 1 - w[i][1];
            wt *= (c & 4) ? w[i][2] : 1 - w[i][2];
            const float q0 = wt * e[i][c].x, q1 = wt * e[i][c].y;
            acc0 += q0;
            acc1 += q1;
        }
        const uint32_t flags = reinterpret_cast<const uint32_t *>(&lv[i])[6];
        fw[i] = (flags & 2u) ? 0u : pack2(acc0, acc1);
    }
    HS_GSTAMP(2);
}

// The same gather with TWO LANES PER POINT, as k_hash_fwd_pair has them: lanes (2 p, 2 p + 1) hold the two x-corners of point p's cell, every lane
// walks all sixteen levels with four reads each.  The single-lane form above asks the L1 for 64 different lines per instruction (two points' worth of
// nothing in common) and is bound by its tag rate: tools/exp/sweep_prof.hip measured 62 clocks per gather instruction, 31 us per 131 072 points, all of
// it in front of the trunk.  Here an instruction covers 32 points x 2 corners that share a line 15 times in 16 -- half the lookups.  The sum keeps the
// reference's order: the lane holding x = 0 adds its own product, then its neighbour's (DPP); for levels 0..7 that is the even lane, for levels 8..15
// the ODD one, so that lane 2 p + h ends up with the eight words lane (p, h) of the trunk's tile feeds to the first layer -- one ds_bpermute per word.
template <class GL, class F>
__device__ __forceinline__ void gather_words_pair(uint32_t (&fw)[8], const float *__restrict__ x01, const char *__restrict__ table, const GL *lv, int64_t tile0,
                                                  int64_t B, int lane, F &&between, [[maybe_unused]] int prof_slot = 0) {
    const int xb0 = lane & 1;
    const int64_t gq = tile0 + (lane >> 1);
    const bool okq = gq < B;
    const float p0 = okq ? x01[gq * 3] : 2.f, p1 = okq ? x01[gq * 3 + 1] : 2.f, p2 = okq ? x01[gq * 3 + 2] : 2.f;
    const bool inside = !(p0 < 0.f || p0 > 1.f) && !(p1 < 0.f || p1 > 1.f) && !(p2 < 0.f || p2 > 1.f);     // both lanes of a pair alike
    uint32_t fwp[8];
#pragma unroll
    for (int i = 0; i < 8; i++) fwp[i] = 0u;
    if (inside) {
        auto half = [&](auto hc) {
            constexpr int HB = decltype(hc)::value;          // levels 8 HB .. 8 HB + 7
            float2 e[8][4];
            float w[8][3];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const GL *L = lv + 8 * HB + i;
                const uint4 la = *reinterpret_cast<const uint4 *>(L), lb = *(reinterpret_cast<const uint4 *>(L) + 1);
                const float scale = __uint_as_float(la.x);
                const uint32_t A = la.y, Bm = la.z, mask = la.w, limit = lb.x, byte_off = lb.y, flags = lb.z;
                uint32_t g[3];
                const float ps[3] = {p0, p1, p2};
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    float pos = ps[d] * scale;
                    const float fl = floorf(pos);
                    g[d] = (uint32_t)fl;
                    pos -= (float)g[d];
                    w[i][d] = gw_smoothstep(pos);
                }
                const uint32_t xb = (uint32_t)(xb0 ^ HB);
                const uint32_t a0 = g[1] * A, a1 = a0 + A, b0 = g[2] * Bm, b1 = b0 + Bm, xs = g[0] + xb;
                uint32_t idx[4];
                if (flags & 1u) {          // (wave-uniform: one level at a time)
#pragma unroll
                    for (int c = 0; c < 4; c++) idx[c] = xs ^ ((c & 1) ? a1 : a0) ^ ((c & 2) ? b1 : b0);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) idx[c] = xs + ((c & 1) ? a1 : a0) + ((c & 2) ? b1 : b0);
                }
                uint32_t top = 0u;
#pragma unroll
                for (int c = 0; c < 4; c++) { idx[c] &= mask; top = max(top, idx[c]); }
                if (top >= limit) {
                    const uint32_t t = lb.w;
#pragma unroll
                    for (int c = 0; c < 4; c++) idx[c] = idx[c] >= t ? idx[c] % t : idx[c];
                }
                const char *base = table + byte_off;
#pragma unroll
                for (int c = 0; c < 4; c++) e[i][c] = *reinterpret_cast<const float2 *>(base + ((size_t)idx[c] << 3));
            }
            if constexpr (HB == 0) { HS_GSTAMP(0); between(); HS_GSTAMP(1); }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t xb = (uint32_t)(xb0 ^ HB);
                float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float wt = 1.f;
                    wt *= xb ? w[i][0] : 1 - w[i][0];
                    wt *= (c & 1) ? w[i][1] : 1 - w[i][1];
                    wt *= (c & 2) ? w[i][2] : 1 - w[i][2];
                    const float q0 = wt * e[i][c].x, q1 = wt * e[i][c].y;
                    acc0 += q0;                                                                                              // corner (0, yz) on the lane that holds x = 0
                    acc0 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q0), 0xB1, 0xf, 0xf, true));        // corner (1, yz): quad_perm [1,0,3,2]
                    acc1 += q1;
                    acc1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q1), 0xB1, 0xf, 0xf, true));
                }
                const uint32_t flags = reinterpret_cast<const uint32_t *>(lv + 8 * HB + i)[6];
                const uint32_t word = (flags & 2u) ? 0u : pack2(acc0, acc1);
                fwp[i] = xb == 0u ? word : fwp[i];       // the x = 0 lane of this half's levels: even lanes keep levels 0..7, odd lanes 8..15
            }
        };
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
    } else {
        between();
    }
    HS_GSTAMP(2);
    // lane (row, h) of the trunk's tile takes its eight words from lane 2 row + h
    const int src = (2 * (lane & 31) + (lane >> 5)) * 4;
#pragma unroll
    for (int i = 0; i < 8; i++) fw[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)fwp[i]);
}

#ifndef HS_SWEEP_PAIR
#define HS_SWEEP_PAIR 1      // 0: the one-lane-per-point gather (A/B, python -m holoscene_amd.csrc.build --variant lane -DHS_SWEEP_PAIR=0)
#endif

// WIDE: 33..64 outputs -- a second 32-neuron tile of the last layer after the first, its fragments (W2b: the [high | low] planes of rows 32..63,
// packed like W2f) and biases (bias2: a pack's bias block whose b2 slots hold rows 32..63) read from memory (16 KB, cache-resident; LDS is full).
// The 32-output instantiation is the kernel the benchmark runs, untouched by the other.
template <bool WIDE> struct WideArgs {};
template <> struct WideArgs<true> { const uint16_t *W2b; const float *bias2; };

// GATHER: the sweep gathers its own hash features (hs_sdf_sweep_fwd) -- `feat` is the TABLE (16 levels x 2 channels, hsHashLayout-free: one grid),
// x01 the points' grid coordinates.  Lane (point, h) needs the features of levels 8 h .. 8 h + 7 of ITS point and nobody else does: it locates the
// point in each of its eight levels, issues the 64 corner reads (global_load_dwordx2, the eight levels' reads in flight together) and blends them in
// the reference's order -- corner 0, 1, ..., 7 with the x bit alternating, a product then a sum per corner and channel, no contraction (this file is
// built with -ffp-contract=off like hash_encode.hip) -- so the word it packs is bit for bit the word k_hash_fwd_pair<2, false> writes with
// hsHashLayout::out_bf16, and every result of the sweep is bit-identical to gather launch + k_sdf_mlp2.  What the fusion buys: the gather is bound by the
// texture addresser and the L1 tag rate, the trunk by the matrix pipe and the VALU; as two launches they run one after the other (31 + 31 us per
// 131 072 points), as one kernel a compute unit's eight waves drift into different phases and the units overlap; the 8 MB feature round trip and one
// launch boundary per round go away.  What it costs: every compute unit now reads all sixteen levels, so an XCD's 4 MB L2 sees the whole 48.8 MB
// table instead of the two levels the XCD-affine gather schedule deals it (measured on the gather alone: 31.1 -> 37.1 us without the affinity).
struct GatherLevel {        // 32 bytes per level, in LDS behind the bias block (two ds_read_b128 per level and tile)
    float scale;            // hashencoder.cu:152
    uint32_t A, B;          // index = g0 (op) g1 A (op) g2 B: the hash primes with xor, or (res, res^2) with +
    uint32_t mask;          // table - 1 for power-of-two tables, else ~0 (then `limit` catches what the reference's modulo wraps)
    uint32_t limit;         // an index >= limit takes the modulo: the table size when it is not a power of two, else ~0 (never)
    uint32_t byte_off;      // of the level's first entry from the table's base (8-byte entries)
    uint32_t flags;         // 1: hashed (xor), 2: empty level (HashEncoder.fused_offsets pads a grid of fewer levels: zero features)
    uint32_t table;
};
template <bool GATHER> struct GatherArgs {};
template <> struct GatherArgs<true> { const float *x01; const int32_t *offsets; float scale[16]; int stagger; };

template <bool WIDE, bool GATHER = false>
__global__ __launch_bounds__(kThreadsW, 2) void k_sdf_mlp2(const float *__restrict__ x, const float *__restrict__ feat, const uint16_t *__restrict__ W0f,
                                                            const uint16_t *__restrict__ W1f, const uint16_t *__restrict__ W2f,
                                                            const float *__restrict__ biasg, int d_out, int select, uint64_t select_mask,
                                                            float *__restrict__ out_min, float *__restrict__ out_raw, int64_t B, hsGate gate,
                                                            int feat_level_major, int lo_plane, WideArgs<WIDE> wide, GatherArgs<GATHER> ga = {}) {
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
    //      LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no staging registers); the workgroup meets once, before the
    //      first layer-1 MFMA.  (A load -> ds_write loop cost 16 us per launch, a batched register copy 6 us, measured.)  The 18
    //      requests of a wave are NOT hidden behind its first tile: vector-memory loads return in order, so the bias loads below,
    //      the first tile's inputs and layer 0's weight fragments from L2 -- all issued after them -- wait for the whole image
    //      (tools/exp/sdf2_prof.hip: kernel entry -> first tile 5.1 k cycles = 2.7 us; a launch is ~9 us of fixed cost + ~11 us
    //      per 65 536 points).  Interleaving the requests with layer 0's fragment loads would overlap the two; not built.
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
        if constexpr (GATHER) {     // the sixteen levels' geometry (hash_encode.hip: level_info / cell_index), once per workgroup
            if (threadIdx.x < 16) {
                GatherLevel gl;
                const uint32_t off = (uint32_t)ga.offsets[threadIdx.x], table = (uint32_t)ga.offsets[threadIdx.x + 1] - off;
                gl.scale = ga.scale[threadIdx.x];
                const uint32_t res = (uint32_t)ceilf(gl.scale) + 1u;
                uint32_t stride = 1;
#pragma unroll
                for (int d = 0; d < 3; d++)
                    if (stride <= table) stride *= res;
                const bool hashed = stride > table, pow2 = (table & (table - 1u)) == 0u, empty = table == 0u;
                gl.A = hashed ? 2654435761u : res;
                gl.B = hashed ? 805459861u : res * res;
                gl.mask = empty ? 0u : (pow2 ? table - 1u : 0xffffffffu);
                gl.limit = (pow2 || empty) ? 0xffffffffu : table;
                gl.byte_off = empty ? 0u : off * 8u;
                gl.flags = (hashed ? 1u : 0u) | (empty ? 2u : 0u);
                gl.table = table;
                reinterpret_cast<GatherLevel *>(bias + kBias)[threadIdx.x] = gl;
            }
        }
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
            auto posenc = [&]() {
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
            };
            if constexpr (!GATHER) posenc();
            uint32_t fw[8];
            if constexpr (GATHER) {
                // Left alone, a compute unit's eight waves issue their first tiles' reads together, the addresser serves them interleaved, they all
                // return together and the waves walk gather | trunk | gather | trunk in lockstep: the units take turns (measured: 69 us = the
                // gather's 37 + the trunk's 32).  Wave w starts its first gather w x stagger later, so the reads reach the (in-order) addresser
                // wave after wave and wave 0 is in its matrix products while wave 1's reads are served.
                if (!resident && ga.stagger > 0)
                    for (int i = 0; i < wave * ga.stagger; i++) __builtin_amdgcn_s_sleep(8);       // 512 clocks per unit
                if constexpr (HS_SWEEP_PAIR)
                    gather_words_pair(fw, ga.x01, reinterpret_cast<const char *>(feat), reinterpret_cast<const GatherLevel *>(bias + kBias), tile * kRows, B, lane,
                                      posenc, tile >= (int64_t)gridDim.x * kWaves ? 4 : 0);
                else
                    gather_words(fw, ga.x01, reinterpret_cast<const char *>(feat), reinterpret_cast<const GatherLevel *>(bias + kBias) + 8 * h, gp, ok, posenc,
                                 tile >= (int64_t)gridDim.x * kWaves ? 4 : 0);
#pragma unroll
                for (int j = 18; j < 34; j++) v[j] = 0.f;
            } else if (feat_level_major == 2) {     // feat: uint32 [16, B], the two channels of a level as bf16 (hs_hash_fwd, hsHashLayout::out_bf16):
                                             // the words ARE this lane's feature inputs (levels 8 h .. 8 h + 7), no conversion
                const uint32_t *fl = reinterpret_cast<const uint32_t *>(feat) + (size_t)(8 * h) * B + (ok ? gp : 0);
#pragma unroll
// k-steps a weight fragment is requested ahead of its MFMAs: layer 0 (W0 from L2) / layer 1 (W1 from LDS)
#ifndef HS_SDF2_AH0
#define HS_SDF2_AH0 1
#endif
#ifndef HS_SDF2_AH1
#define HS_SDF2_AH1 2
#endif
#ifndef HS_NT_SDF_IN
#define HS_NT_SDF_IN 0
#endif
                for (int i = 0; i < 8; i++) fw[i] = ok ? (HS_NT_SDF_IN ? __builtin_nontemporal_load(fl + (size_t)i * B) : fl[(size_t)i * B]) : 0u;      // (NT: the words' only read)
#pragma unroll
                for (int j = 18; j < 34; j++) v[j] = 0.f;
            } else if (feat_level_major) {      // feat [16, B, 2]
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
            if (GATHER || feat_level_major == 2) {
#pragma unroll
                for (int i = 0; i < 8; i++) hin[9 + i] = fw[i];
            }
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
                if (q == 0) phase2<K0S, HS_SDF2_AH0, K0S, 16, false>(acc[0], hin, f0, [](auto) {});
                else phase2<K0S, HS_SDF2_AH0, K0S, 16, true>(acc[q & 1], hin, f0, [&](auto slc) { constexpr int sl = decltype(slc)::value;
                    h0p[16 * (q - 1) + 8 * (sl >> 3) + (sl & 7)] = epilogue_pair(acc[(q & 1) ^ 1][sl >> 3], 2 * (sl & 7)); });
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
            if (q == 0) phase2<HS, HS_SDF2_AH1, 8, 16, true>(acc[0], h0p, f1, [&](auto slc) { constexpr int sl = decltype(slc)::value; h0p[48 + 8 * (sl >> 3) + (sl & 7)] = epilogue_pair(acc[1][sl >> 3], 2 * (sl & 7)); });
            else phase2<HS, HS_SDF2_AH1, HS, 16, true>(acc[q & 1], h0p, f1, [&](auto slc) { constexpr int sl = decltype(slc)::value;
                h1p[16 * (q - 1) + 8 * (sl >> 3) + (sl & 7)] = epilogue_pair(acc[(q & 1) ^ 1][sl >> 3], 2 * (sl & 7)); });
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
            if (kW2LowPlane && lo_plane) {    // + W2lo h1 (wave_tile.h): the low plane's fragments from memory (16 KB, cache-resident; LDS is full)
                uint32_t zlo = 0;
                asm volatile("" : "+v"(zlo));
                const bf16x8 *W2q = reinterpret_cast<const bf16x8 *>(W2f + kW2F) + lane + zlo;
#pragma unroll
                for (int s = 0; s < HS / 2; s++) {
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s) * 64], frag_of(h1p + 4 * (2 * s)), y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s + 1) * 64], frag_of(h1p + 4 * (2 * s + 1)), y1, 0, 0, 0);
                }
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
        if constexpr (!WIDE) {
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
        } else {
            auto store_raw = [&](int base) {      // this lane's 16 outputs of the tile that starts at neuron `base`
                float *dst = out_raw + gp * d_out + base;
                if ((d_out & 3) == 0) {      // 16-byte runs: neurons 8 q + 4 h .. + 3
    #pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (base + 8 * q + 4 * h < d_out) *reinterpret_cast<float4 *>(dst + 8 * q + 4 * h) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
                } else {
    #pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                        if (base + n < d_out) dst[n] = y[i];
                    }
                }
            };
            if (out_raw && ok) store_raw(0);
            const uint16_t *__restrict__ W2b = wide.W2b;
            const float *__restrict__ bias2 = wide.bias2;
            f32x16 &y0 = acc[0][0], &y1 = acc[0][1];
#pragma unroll
            for (int i = 0; i < 16; i++) { y0[i] = 0.f; y1[i] = 0.f; }
            uint32_t zb = 0;
            asm volatile("" : "+v"(zb));
            const bf16x8 *W2q = reinterpret_cast<const bf16x8 *>(W2b) + lane + zb;
#pragma unroll
            for (int s = 0; s < HS / 2; s++) {
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s) * 64], frag_of(h1p + 4 * (2 * s)), y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s + 1) * 64], frag_of(h1p + 4 * (2 * s + 1)), y1, 0, 0, 0);
            }
            if (kW2LowPlane && lo_plane) {
                const bf16x8 *W2r = W2q + (size_t)HS * 64;
#pragma unroll
                for (int s = 0; s < HS / 2; s++) {
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2r[(size_t)(2 * s) * 64], frag_of(h1p + 4 * (2 * s)), y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2r[(size_t)(2 * s + 1) * 64], frag_of(h1p + 4 * (2 * s + 1)), y1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                const float val = y0[i] + y1[i] + bias2[512 + n];
                y[i] = val;
                if (32 + n < d_out) {
                    const bool take = select_mask ? ((select_mask >> (32 + n)) & 1ull) != 0ull : (select < 0 || 32 + n == select);
                    if (take) best = fminf(best, val);
                }
            }
            if (out_raw && ok) store_raw(32);
            best = fminf(best, __shfl_xor(best, 32));
            if (h == 0 && ok) out_min[gp] = best;
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

}  // namespace

extern "C" {

int64_t hs_sdf_mlp2_pack_bytes(int32_t which) {
    switch (which) {
        case 0: return (int64_t)kW0F * 2;
        case 1: return (int64_t)kW1F * 2;
        case 2: return (int64_t)kW2F * 2 * 2;    /* two planes: W2's fragments, then those of W2 - bf16(W2) (wave_tile.h) */
        case 3: return (int64_t)kBias * 4;
        default: return -1;
    }
}

int hs_sdf_mlp2_pack(const float *W0, int32_t ld0, const float *b0, const float *W1, const float *b1, const float *W2, const float *b2, int32_t d_out,
                     void *W0f, void *W1f, void *W2f, float *bias, int32_t log2_domain, void *stream) {
    if (d_out < 1 || d_out > 32 || ld0 < 71) return HS_ERR_ARG;
    if (!W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !W0f || !W1f || !W2f || !bias) return HS_ERR_NULL;
    const int slots = kSdfPackSlots;
    k_sdf_pack2<<<(slots + 255) / 256, 256, 0, (hipStream_t)stream>>>(W0, ld0, b0, W1, b1, W2, b2, d_out, (uint16_t *)W0f, (uint16_t *)W1f,
                                                                      (uint16_t *)W2f, bias, log2_domain ? kAct : 1.f);
    return wt_check_launch();
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
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_sdf_mlp2<false>, (int)lds);
    const int64_t ntiles = (B + kRows - 1) / kRows;
    const int64_t want = (ntiles + kWaves - 1) / kWaves;
    const int grid = (int)(want < 256 ? want : 256);      // one workgroup per CU (146 KB of LDS), wave tiles strided across the grid
    // The no-grad sweeps leave W2's low plane out by default: they only place samples, a 300-iteration training with fp32 sweeps ends where
    // one with these bf16 sweeps does (profiles/r05/bf16_stage_hunt.txt, stage "sampler"), and the plane's fragments -- from memory, this
    // kernel's LDS is full -- cost 3 us per sweep (same-box A/B: 1.606 -> 1.621 ms per iteration).  HOLOSCENE_SDF_W2_PLANES=2 adds it.
    static const int planes = [] { const char *e = getenv("HOLOSCENE_SDF_W2_PLANES"); return (e && e[0] == '2') ? 2 : 1; }();
    k_sdf_mlp2<false><<<grid, kThreadsW, lds, (hipStream_t)stream>>>(x, feat, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias, d_out,
                                                                      select, select_mask, out_min, out_raw, B, gate ? *gate : hsGate{nullptr, nullptr},
                                                                      feat_level_major, planes == 2, WideArgs<false>{});
    return wt_check_launch();
}

// The sampler's sweep as ONE launch: hash gather inside the wave tile (k_sdf_mlp2<WIDE, true>).  One table of 16 levels x 2 channels (a grid of
// fewer levels: HashEncoder.fused_offsets' empty levels), 3-D points; W2f_b / bias_b: the second output tile's pack for 33..64 outputs, else NULL.
int hs_sdf_sweep_fwd(const float *x, const float *x01, const float *embeddings, const int32_t *offsets, float S, uint32_t H, const void *W0f,
                     const void *W1f, const void *W2f, const float *bias, const void *W2f_b, const float *bias_b, int32_t d_out, int32_t select,
                     uint64_t select_mask, float *out_min, float *out_raw, int64_t B, const hsGate *gate, void *stream) {
    const bool wide = d_out > 32;
    if (d_out < 1 || d_out > 64 || select >= d_out) return HS_ERR_ARG;
    if (select_mask && d_out < 64 && (select_mask >> d_out)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!x || !x01 || !embeddings || !offsets || !W0f || !W1f || !W2f || !bias || !out_min || (wide && (!W2f_b || !bias_b))) return HS_ERR_NULL;
    if ((const char *)W2f != (const char *)W1f + (size_t)kW1F * 2) return HS_ERR_ARG;
    const size_t lds = (size_t)(kW1F + kW2F) * sizeof(uint16_t) + kBias * sizeof(float) + 16 * sizeof(GatherLevel);
    const int64_t ntiles = (B + kRows - 1) / kRows;
    const int64_t want = (ntiles + kWaves - 1) / kWaves;
    const int grid = (int)(want < 256 ? want : 256);
    static const int planes = [] { const char *e = getenv("HOLOSCENE_SDF_W2_PLANES"); return (e && e[0] == '2') ? 2 : 1; }();
    GatherArgs<true> ga;
    ga.x01 = x01;
    ga.offsets = offsets;
    static const int stagger = [] { const char *e = getenv("HOLOSCENE_SWEEP_STAGGER"); return e ? atoi(e) : 4; }();      // units of 512 clocks per wave index
    ga.stagger = stagger;
    for (int l = 0; l < 16; l++) ga.scale[l] = exp2f((float)l * S) * (float)H - 1.0f;      // hash_encode.hip: make_scales (hashencoder.cu:152)
    const hsGate g = gate ? *gate : hsGate{nullptr, nullptr};
    if (wide) {
        static hsLdsAttrOnce attr;
        attr.set((const void *)k_sdf_mlp2<true, true>, (int)lds);
        k_sdf_mlp2<true, true><<<grid, kThreadsW, lds, (hipStream_t)stream>>>(x, embeddings, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias,
                                                                              d_out, select, select_mask, out_min, out_raw, B, g, 2, planes == 2,
                                                                              WideArgs<true>{(const uint16_t *)W2f_b, bias_b}, ga);
    } else {
        static hsLdsAttrOnce attr;
        attr.set((const void *)k_sdf_mlp2<false, true>, (int)lds);
        k_sdf_mlp2<false, true><<<grid, kThreadsW, lds, (hipStream_t)stream>>>(x, embeddings, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias,
                                                                               d_out, select, select_mask, out_min, out_raw, B, g, 2, planes == 2,
                                                                               WideArgs<false>{}, ga);
    }
    return wt_check_launch();
}

int hs_sdf_mlp2_fwd_wide(const float *x, const float *feat, const void *W0f, const void *W1f, const void *W2f, const float *bias, const void *W2f_b,
                         const float *bias_b, int32_t d_out, int32_t select, uint64_t select_mask, float *out_min, float *out_raw, int64_t B,
                         const hsGate *gate, int32_t feat_level_major, void *stream) {
    if (d_out < 33 || d_out > 64 || select >= d_out) return HS_ERR_ARG;
    if (select_mask && d_out < 64 && (select_mask >> d_out)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!x || !feat || !W0f || !W1f || !W2f || !bias || !W2f_b || !bias_b || !out_min) return HS_ERR_NULL;
    if ((const char *)W2f != (const char *)W1f + (size_t)kW1F * 2) return HS_ERR_ARG;
    const size_t lds = (size_t)(kW1F + kW2F) * sizeof(uint16_t) + kBias * sizeof(float);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_sdf_mlp2<true>, (int)lds);
    const int64_t ntiles = (B + kRows - 1) / kRows;
    const int64_t want = (ntiles + kWaves - 1) / kWaves;
    const int grid = (int)(want < 256 ? want : 256);
    static const int planes = [] { const char *e = getenv("HOLOSCENE_SDF_W2_PLANES"); return (e && e[0] == '2') ? 2 : 1; }();
    k_sdf_mlp2<true><<<grid, kThreadsW, lds, (hipStream_t)stream>>>(x, feat, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias, d_out,
                                                                     select, select_mask, out_min, out_raw, B, gate ? *gate : hsGate{nullptr, nullptr},
                                                                     feat_level_major, planes == 2, WideArgs<true>{(const uint16_t *)W2f_b, bias_b});
    return wt_check_launch();
}

}  // extern "C"
