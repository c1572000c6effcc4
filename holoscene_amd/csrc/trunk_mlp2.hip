// holoscene_amd/csrc/trunk_mlp2.hip -- the differentiable value+Jacobian trunk pass in "wave tile" form (gfx950).
//
// Same function and same saved tensors as k_trunk_fwd (sdf_mlp.hip): the SDF trunk of ObjectImplicitNetworkGrid.forward
// (model/network.py:203-206) over 4 rows per point -- the value and its three input tangents (DESIGN V1: this replaces the 1 + (K + 1)
// autograd re-traversals of network.py:213-236, 293-299) -- 71 -> 256 -> 256 -> d_out, Softplus(beta = 100) on the value row and
// t <- sigmoid(100 v) * t on the tangent rows.  Structure of sdf_mlp2.hip (wave_tile.h): a wave owns 32 rows (8 points) end to end,
// activations stay in registers as the next layer's B fragments, W1 / W2 are LDS-resident in fragment order, W0 streams from L2, every
// epilogue rides in the MFMA shadow of the next neuron quarter.  What the training pass adds:
//
//   * the input rows are built by the lanes that consume them, straight from x, the hash features [B,32] and dy_dx [L,B,3C] as
//     hs_hash_fwd wrote them (k_trunk_input_fwd's launch and its 80 MB round trip disappear): lane (row, h) of a quad evaluates the
//     octaves 3h..3h+2 -- the value lane keeps sin / cos, tangent lane d keeps +-2^k cos / sin on component d -- and converts the
//     eight levels 8h..8h+7 of the features (value row) or of dy_dx[.., d, :] * d(x01)/dx (tangent row d);
//   * the tangent rule is a DPP quad broadcast: the four rows of a point are the four lanes of a quad of the accumulator layout, and
//     lane j of the quad evaluates the ONE softplus / sigmoid pair of neuron j of the value lane's four (sdf_mlp.hip: tangent_quad);
//   * H0 / H1 leave for the backward kernel and the weight-gradient GEMMs as row-major [M,256] bf16: a lane holds 8-byte runs
//     (4 consecutive neurons); v_permlane32_swap_b32 between the two halves of the wave turns two of them into one 16-byte run per
//     lane, halving the store instructions (the epilogue store tail is issue-bound);
//   * the assembled input rows leave as Xp [M,80] bf16 in the kernel's own column order (lane (row, h) writes its 40 values as one
//     80-byte run): the first layer's weight gradient is taken against Xp and un-permuted on the host side (71 columns).
#include "launch_util.h"
#include "wave_tile.h"

#ifdef HS_TRUNK2_PROFILE     // tools/exp/trunk2_prof.hip: per-phase s_memtime stamps of a wave tile
__device__ unsigned long long g_trunk2_prof[256 * 8 * 8];
#ifndef HS_TSTAMP_TILE
#define HS_TSTAMP_TILE 2      // which of the wave's tiles is stamped (the last ones run on a half-empty chip)
#endif
#define HS_TSTAMP(i) do { if (lane == 0 && tile == (int64_t)wave * gridDim.x + blockIdx.x + (int64_t)HS_TSTAMP_TILE * gridDim.x * kWaves) \
        g_trunk2_prof[(blockIdx.x * 8 + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HS_TSTAMP(i) do { } while (0)
#endif

namespace {

// Softplus(beta = 100) and its derivative for neuron j of the quad's value lane, then broadcast back (sdf_mlp.hip: tangent_quad):
// out = softplus(v) on the value lane, sigmoid(100 v) * a on the tangent lanes.  The accumulators start from zero (the first MFMA of a
// tile takes the constant 0 as C), and bias_j, the bias of neuron j, is added once the lane has its neuron.  The epilogues are the kernel's VALU load (30 k of 41 k cycles per tile pair before
// this form), so the count per quad matters: DPP operands are written so that GCNDPPCombine folds them into the consuming select /
// multiply, and the threshold branch of torch's Softplus is a max (softplus(v) > v below the threshold; above it the capped log
// term is 0.2 < v) and disappears from the derivative (e / (1 + e) rounds to 1 once e = 2^28.85).
struct QuadMasks {      // lane-pattern constants as wave masks (SGPR pairs): lanes whose index in the quad is NOT 1 / 2 / 3 / 0
    uint64_t n1, n2, n3, tangent;
};
__device__ __forceinline__ QuadMasks quad_masks() {
    QuadMasks m = {0xddddddddddddddddull, 0xbbbbbbbbbbbbbbbbull, 0x7777777777777777ull, 0xeeeeeeeeeeeeeeeeull};
    asm volatile("" : "+s"(m.n1), "+s"(m.n2), "+s"(m.n3), "+s"(m.tangent));
    return m;
}

#define HS_DPPQ(k) " quad_perm:[" #k "," #k "," #k "," #k "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void tangent_quad2(const float a[4], float bias_j, const QuadMasks &m, float out[4]) {
    // lane j of the quad takes the value lane's register a[j]: one DPP move (compiler-visible, so that the MFMA -> VALU read hazard of
    // the quad's accumulator registers is covered by the hazard recogniser) and three selects with the DPP read folded in.
    // The cross-lane groups are assembly because the compiler does neither fold (it emits v_mov_dpp + v_pk_mul + v_cndmask, 14
    // instructions for the 8 below) and turns a select chain on the lane index into exec-masked branches.
    float v = dpp_quad<0x00>(a[0]);
    asm("s_mov_b64 vcc, %4\n\t"
        "v_cndmask_b32_dpp %0, %1, %0, vcc" HS_DPPQ(0)
        "s_mov_b64 vcc, %5\n\t"
        "v_cndmask_b32_dpp %0, %2, %0, vcc" HS_DPPQ(0)
        "s_mov_b64 vcc, %6\n\t"
        "v_cndmask_b32_dpp %0, %3, %0, vcc" HS_DPPQ(0)
        : "+v"(v)
        : "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(m.n1), "s"(m.n2), "s"(m.n3)
        : "vcc");
    v += bias_j;
    const float e = __builtin_amdgcn_exp2f(fminf(v * 144.269504f, 28.8539008f));     // e^(100 v), capped at e^20
    const float one_e = 1.f + e;
    const float lg = __builtin_amdgcn_logf(one_e) * (0.69314718f * 0.01f);
    const float ds = e * __builtin_amdgcn_rcpf(one_e);
    float sp;
    asm("v_max_f32 %0, %1, %2" : "=v"(sp) : "v"(v), "v"(lg));          // (fmaxf would first canonicalise v with a v_max v, v)
    // out[k] = sigmoid_k * a[k] on the tangent lanes (vcc), softplus_k on the value lane.  s_nop 1: a DPP read needs two wait states
    // after the VALU write of its source, and the assembler does not insert them inside an asm block
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %4, %6" HS_DPPQ(0)
        "v_mul_f32_dpp %1, %4, %7" HS_DPPQ(1)
        "v_mul_f32_dpp %2, %4, %8" HS_DPPQ(2)
        "v_mul_f32_dpp %3, %4, %9" HS_DPPQ(3)
        "s_mov_b64 vcc, %10\n\t"
        "v_cndmask_b32_dpp %0, %5, %0, vcc" HS_DPPQ(0)
        "v_cndmask_b32_dpp %1, %5, %1, vcc" HS_DPPQ(1)
        "v_cndmask_b32_dpp %2, %5, %2, vcc" HS_DPPQ(2)
        "v_cndmask_b32_dpp %3, %5, %3, vcc" HS_DPPQ(3)
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
        : "v"(ds), "v"(sp), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(m.tangent)
        : "vcc");
}
#undef HS_DPPQ

// Quad q of a tile: accumulator registers 4q..4q+3 = neurons 32 nt + 8 q + 4 h + (0..3) -> the two packed words hw8[2q], hw8[2q+1] of the
// tile's 8-word block (the next layer's B fragments of k-steps 2 nt, 2 nt + 1).
// bt: this lane's four biases of the tile (tile_bias)
__device__ __forceinline__ void quad_words(const f32x16 &acc, int q, const f32x4 &bt, uint32_t *hw8, const QuadMasks &qm) {
    const float a[4] = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    float o[4];
    tangent_quad2(a, bt[q], qm, o);
    hw8[2 * q] = anchor(pack2(o[0], o[1]));
    hw8[2 * q + 1] = anchor(pack2(o[2], o[3]));
}

// The hidden layers' biases sit in LDS in the order the epilogue wants them: lane (h, j) of a quad takes neuron 8 q + 4 h + j of tile nt in
// quad q, so its four biases of a tile are one 16-byte read at float 32 nt + 16 h + 4 j (+ q).  bias: opaque LDS address of the lane's
// (h, j) slot (wave_tile.h: lds_base), boff the layer's float offset.
__device__ __forceinline__ f32x4 tile_bias(uint32_t bias, int boff, int nt) { return lds_at<f32x4>(bias, (uint32_t)(boff + 32 * nt) * 4u); }
__device__ __forceinline__ int bias_slot_source(int i) {      // float i of the LDS image <- float of the packed bias block
    const int r = i & 31, h = r >> 4, j = (r >> 2) & 3, q = r & 3;
    return i < 512 ? (i & ~31) + 8 * q + 4 * h + j : i;
}

// Row-major store of quads 2p, 2p+1 of tile nt (16 neurons of this row): after the half-wave swap lane (row, 0) holds neurons
// 16 p .. 16 p + 7 of the tile, lane (row, 1) neurons 16 p + 8 .. 16 p + 15 -- one 16-byte store each instead of two 8-byte ones.
// (gfx9 loads and stores share vmcnt and return in order: a store in front of a weight-fragment load makes the MFMA that needs the
// fragment wait for the store's acknowledgement too, hence layer 0's fragments travel two tiles ahead of their stores' phase.)
__device__ __forceinline__ void store_pair(const uint32_t *hw8, int p, int nt, uint16_t *__restrict__ Hrow, bool ok, int h) {
    const auto ra = __builtin_amdgcn_permlane32_swap(hw8[4 * p], hw8[4 * p + 2], false, false);       // {own | partner} words of neurons +0,+1
    const auto rb = __builtin_amdgcn_permlane32_swap(hw8[4 * p + 1], hw8[4 * p + 3], false, false);   //                          neurons +2,+3
    if (ok) *reinterpret_cast<uint4 *>(Hrow + 32 * nt + 16 * p + 8 * h) = make_uint4(ra[0], rb[0], ra[1], rb[1]);
}

// Two tiles (nt even, nt + 1) = one full 128-byte line of the row, its four 32-byte pieces stored back to back.  HBM takes the activations
// at 6.5 TB/s when a line's pieces arrive together, 5.5 TB/s as whole lines spaced apart, 4.2 TB/s as 64-byte halves a phase apart and
// 1.6 TB/s as lone 32-byte pieces (tools/exp/store_pattern.hip) -- and this kernel is bound by exactly that drain.
template <int TILES = 2>
__device__ __forceinline__ void store_line(const uint32_t *hw16, int nt, uint16_t *__restrict__ Hrow, bool ok, int h) {
#pragma unroll
    for (int i = 0; i < 2 * TILES; i++) store_pair(hw16 + 8 * (i >> 1), i & 1, nt + (i >> 1), Hrow, ok, h);
}

// WIDE (with SPLIT): 33..64 outputs -- the last layer's second 32-row tile after the first, as in sdf_mlp2.hip's k_sdf_mlp2<true>: fragments (both
// planes) and biases of rows 32.. from a second pack in memory; the split outputs are formed tile by tile and the value row picks between the tiles.
template <bool WIDE> struct TrunkWideArgs {};
template <> struct TrunkWideArgs<true> { const uint16_t *W2b; const float *bias2; };
template <bool SPLIT, bool WIDE = false>
__global__ __launch_bounds__(kThreadsW, 2) void k_trunk_fwd2(const float *__restrict__ x, const float *__restrict__ feat, const float *__restrict__ dydx,
                                                              const uint16_t *__restrict__ W0f, const uint16_t *__restrict__ W1f,
                                                              const uint16_t *__restrict__ W2f, const float *__restrict__ biasg, int d_out,
                                                              uint16_t *__restrict__ H0, uint16_t *__restrict__ H1, float *__restrict__ Y,
                                                              uint16_t *__restrict__ Xp, int64_t M, float jac_scale, hsTrunkSplit sp, int64_t ld, int lo_plane,
                                                              TrunkWideArgs<WIDE> wide) {
    static_assert(SPLIT || !WIDE, "the wide form exists for the split outputs only");
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *W1l = lds;
    uint16_t *W2l = lds + kW1F;
    float *bias = reinterpret_cast<float *>(W2l + kW2F);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, h = lane >> 5;
    const bool is_value = (lane & 3) == 0;
    const QuadMasks qm = quad_masks();
    // resident weights by LDS-DMA, in flight under the first tile's input build and layer 0 (sdf_mlp2.hip)
    {
        constexpr int kChunks = (kW1F + kW2F) * 2 / 1024;
        static_assert(kChunks % kWaves == 0, "resident image must split evenly over the waves");
        const char *src = reinterpret_cast<const char *>(W1f);
        char *dst = reinterpret_cast<char *>(W1l);
#pragma unroll
        for (int i = 0; i < kChunks / kWaves; i++) {
            const int c = wave + i * kWaves;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(dst + (size_t)c * 1024), 16, 0, 0);
        }
        for (int i = threadIdx.x; i < kBias; i += kThreadsW) bias[i] = biasg[bias_slot_source(i)];
    }
    __syncthreads();
    bool resident = false;
    const int64_t ntiles = (M + kRows - 1) / kRows, Bp = M >> 2;
    // opaque LDS bases (wave_tile.h: lds_base): two 64 KB windows of W1, W2, and this half-wave's bias quads
    const uint32_t w1a0_ = lds_base(W1l, lane * 16), w1a1_ = lds_base(W1l, lane * 16 + 65536u), w2a_ = lds_base(W2l, lane * 16), ba_ = lds_base(bias, 64 * h + 16 * (lane & 3)),
                   bo_ = lds_base(bias + 512, 16 * h);
    const int t = lane & 3;                          // 0 value, 1..3 tangent d = t - 1   (tiles start at multiples of 32)
    // The raw inputs of a tile (position, eight levels of features or feature derivatives: 19 registers) are requested a tile EARLY, at
    // the start of the previous tile's layer 2 where layer 0's activations have just died: loads queue behind the stores of all eight
    // waves in the CU's memory pipeline, and at the HBM write rate this kernel runs at that queue is thousands of cycles long
    // (measured: 11-13 k cycles of a 55 k cycle tile waiting for these loads when they were issued at the tile's start).
    struct Raw { float xs[3]; float2 f[8]; };
    auto load_raw = [&](int64_t tl) {
        const int64_t g = tl * kRows + row;
        const int64_t b = g < M ? (g >> 2) : 0;
        Raw r;
        r.xs[0] = x[b * 3]; r.xs[1] = x[b * 3 + 1]; r.xs[2] = x[b * 3 + 2];
        // eight levels 8h..8h+7: features (value row, [B,32] point-major) or dy_dx[level][b][d][c] (tangent row d, [L,B,3C])
        const float *fp = t == 0 ? feat + b * 32 + 16 * h : dydx + ((int64_t)(8 * h) * ld + b) * 6 + 2 * (t - 1);      // ld: points per level of dy_dx
        const int64_t fstride = t == 0 ? 2 : ld * 6;
#pragma unroll
        for (int i = 0; i < 8; i++) r.f[i] = *reinterpret_cast<const float2 *>(fp + i * fstride);
        return r;
    };
    // tile -> (workgroup, wave): wave-major, i.e. tile = wave * gridDim.x + blockIdx.x (+ rounds of gridDim.x * kWaves).  A launch with fewer tiles
    // than 256 x 8 then spreads them over ALL compute units, one or two waves each, instead of filling eight waves of a few (the Eikonal set's
    // 512 tiles: 256 workgroups with two live waves on two SIMDs instead of 64 full ones sharing theirs)
    const int64_t tile0 = (int64_t)wave * gridDim.x + blockIdx.x, tstride = (int64_t)gridDim.x * kWaves;
    Raw raw = load_raw(tile0 < ntiles ? tile0 : 0);
    for (int64_t tile = tile0; tile < ntiles; tile += tstride) {
        HS_TSTAMP(0);
        const uint32_t w1a0 = w1a0_, w1a1 = w1a1_, w2a = w2a_, ba = ba_, bo = bo_;
        const int64_t gr = tile * kRows + row;      // this lane's row; its point and row type
        const bool ok = gr < M;
        // ---- this lane's 40 inputs (wave_tile.h: input_column order), as five B fragments; also stored as Xp[row][40 h ..]
        uint32_t hin[4 * K0S];
        {
            float v[40];
            const float xs[3] = {raw.xs[0], raw.xs[1], raw.xs[2]};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float f = h ? (float)(8 << k) : (float)(1 << k);    // octave 3h + k
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    float sn, cs;
                    __sincosf(xs[d] * f, &sn, &cs);
                    const bool mine = t == d + 1;                           // tangent row d: d/dx_d of sin / cos, zero on the other components
                    v[6 * k + d] = t == 0 ? sn : (mine ? f * cs : 0.f);
                    v[6 * k + 3 + d] = t == 0 ? cs : (mine ? -f * sn : 0.f);
                }
            }
            const float fs = t == 0 ? 1.f : jac_scale;       // tangent rows: dy_dx * d(x01)/dx
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float2 u = raw.f[i];
                v[18 + 2 * i] = u.x * fs;
                v[19 + 2 * i] = u.y * fs;
            }
#pragma unroll
            for (int d = 0; d < 3; d++) v[34 + d] = h ? 0.f : (t == 0 ? xs[d] : (t == d + 1 ? 1.f : 0.f));
            v[37] = v[38] = v[39] = 0.f;
#pragma unroll
            for (int j = 0; j < 40; j += 2) hin[j >> 1] = ok ? pack2(v[j], v[j + 1]) : 0u;
            if (ok) {
                uint4 *xp = reinterpret_cast<uint4 *>(Xp + gr * 80 + 40 * h);
#pragma unroll
                for (int i = 0; i < 5; i++) xp[i] = make_uint4(hin[4 * i], hin[4 * i + 1], hin[4 * i + 2], hin[4 * i + 3]);
            }
        }
        HS_TSTAMP(1);
        uint16_t *h0row = H0 + gr * 256, *h1row = H1 + gr * 256;
        uint32_t h0p[64], h1p[64];
        f32x16 acc[2][2];
        // ---- layer 0: eight single-tile phases, weights from L2 two tiles ahead (3 x 5 fragments).  The H0 stores are spread over the
        //      phases as well (tile nt-2 in phase nt): the chip drains about 10 bytes per clock per CU to HBM and the eight waves of a CU
        //      produce 33 KB of activations per tile each, so the stores have to be paced over the WHOLE tile -- bunched into layers 1 and 2
        //      they back up and stall the issuing waves (measured: 211 us per launch, 135 us with the layer-1 stores removed, while
        //      the same bytes stored back to back by an otherwise idle kernel take 84 us, tools/exp/store_pattern.hip)
        {
            bf16x8 w0[3][K0S];
            // scalar base + 32-bit lane offset + immediate: fragment (s, nt) sits at byte (s NT + nt) 1024 + 16 lane of the image; one opaque
            // offset register per k-step (made here, live through layer 0 only) centred so that every nt lands in the signed 13-bit offset
            // field -- no address arithmetic per load (a 64-bit per-lane pointer costs a v_add_co / v_addc pair each; leaving the constants
            // to the compiler hoists forty scalar bases out of the tile loop and spills them).  Opaque per tile: the tile-invariant
            // loads stay inside the tile loop
            uint32_t w0s[K0S];
            static_for<K0S>([&](auto sc) { constexpr int s = decltype(sc)::value; w0s[s] = lane * 16 + s * NT * 1024 + 4096; asm volatile("" : "+v"(w0s[s])); });
            auto W0at = [&](auto sc, auto nc) {
                constexpr int s = decltype(sc)::value, nt = decltype(nc)::value;
                return *reinterpret_cast<const bf16x8 *>(reinterpret_cast<const char *>(W0f) + (size_t)w0s[s] + (ptrdiff_t)(nt * 1024 - 4096));
            };
            static_for<K0S>([&](auto sc) { constexpr int s = decltype(sc)::value; w0[0][s] = W0at(sc, std::integral_constant<int, 0>{}); });
            static_for<K0S>([&](auto sc) { constexpr int s = decltype(sc)::value; w0[1][s] = W0at(sc, std::integral_constant<int, 1>{}); });
            static_for<NT>([&](auto nc) {
                constexpr int nt = decltype(nc)::value;
                f32x16 &cur = acc[0][nt & 1];
                const f32x4 bt = tile_bias(ba, 0, nt > 0 ? nt - 1 : 0);
                if constexpr (nt + 2 < NT)
                    static_for<K0S>([&](auto sc) { constexpr int s = decltype(sc)::value; w0[(nt + 2) % 3][s] = W0at(sc, std::integral_constant<int, nt + 2>{}); });
                auto f0 = [&](int s) { return w0[nt % 3][s]; };
                if constexpr (nt == 0) phase1<K0S, K0S, 4, false, true>(cur, hin, f0, [](auto) {});
                else if constexpr (nt < 3 || (nt & 1) == 0) phase1<K0S, K0S, 4, true, true>(cur, hin, f0, [&](auto qc) {
                    quad_words(acc[0][(nt - 1) & 1], decltype(qc)::value, bt, h0p + 8 * (nt - 1), qm); });
                else phase1<K0S, K0S, 5, true, true>(cur, hin, f0, [&](auto slc) {
                    constexpr int sl = decltype(slc)::value;
                    if constexpr (sl < 4) quad_words(acc[0][(nt - 1) & 1], sl, bt, h0p + 8 * (nt - 1), qm);
                    else store_line(h0p + 8 * (nt - 3), nt - 3, h0row, ok, h);                      // H0 tiles nt - 3, nt - 2
                });
            });
        }
        HS_TSTAMP(2);
        if (!resident) {        // first tile of this wave: its own DMA requests have landed, then everybody's
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            resident = true;
        }
        // ---- layer 1: quarters 0-3, weights from LDS.  Slices in the MFMA shadow of quarter q: the previous quarter's epilogue quads
        //      (q = 0: the four quads of layer 0's last tile, which sat in accumulator acc[0][1]) and row-major stores (q = 0: H0 tiles
        //      6, 7; q >= 1: H1 tiles 2(q-1), 2(q-1)+1)
        static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            // accumulator sets: quarter q accumulates into set A(q), its epilogue reads set A(q-1).  Layer 0's tile 7 lives in
            // acc[0][1], so quarter 0 uses {acc[1][0], acc[1][1]} and the sets alternate from there: A(q) = acc[(q + 1) & 1]
            f32x16 (&cur)[2] = acc[(q + 1) & 1];
            // biases of the tiles whose epilogue runs in this phase: layer 0's tile 7 (q = 0), else layer 1's tiles 2(q-1), 2(q-1)+1
            const f32x4 bt0 = q == 0 ? tile_bias(ba, 0, 7) : tile_bias(ba, 256, 2 * (q > 0 ? q - 1 : 0));
            const f32x4 bt1 = tile_bias(ba, 256, 2 * (q > 0 ? q - 1 : 0) + 1);
            auto f1 = [&](int s, int j) {
                const uint32_t off = (uint32_t)(s * NT + 2 * q + j) * 1024u;
                return lds_at<bf16x8>(off >> 16 ? w1a1 : w1a0, off & 65535u);
            };
            if constexpr (q == 0) {
                phase2<HS, 2, 10, 5, true, true>(cur, h0p, f1, [&](auto slc) {
                    constexpr int sl = decltype(slc)::value;
                    if constexpr (sl < 4) quad_words(acc[0][1], sl, bt0, h0p + 56, qm);
                    else store_line(h0p + 48, 6, h0row, ok, h);                                     // H0 tiles 6, 7
                });
            } else {
                phase2<HS, 2, HS, 9, true, true>(cur, h0p, f1, [&](auto slc) {
                    constexpr int sl = decltype(slc)::value;
                    if constexpr (sl < 8) quad_words(acc[q & 1][sl >> 2], sl & 3, sl < 4 ? bt0 : bt1, h1p + 16 * (q - 1) + 8 * (sl >> 2), qm);
                    else store_line(h1p + 16 * (q - 1), 2 * (q - 1), h1row, ok, h);                 // H1 tiles 2(q-1), 2(q-1)+1
                });
            }
        });
        HS_TSTAMP(3);
        raw = load_raw(tile + tstride < ntiles ? tile + tstride : tile);      // next tile's inputs (see load_raw)
        __builtin_amdgcn_sched_barrier(0);
        // ---- layer 2 on two partial accumulators; in its shadow layer 1's last epilogue (quarter 3 sits in acc[0]) and the H1 stores
        //      of tiles 6, 7
        f32x16 &y0 = acc[1][0], &y1 = acc[1][1];
        {
            auto f2 = [&](int s, int j) { return lds_at<bf16x8>(w2a, (uint32_t)(2 * s + j) * 1024u); };
            bf16x8 ring[3][2];
            const f32x4 bt0 = tile_bias(ba, 256, 6), bt1 = tile_bias(ba, 256, 7);
            static_for<2>([&](auto sc) { constexpr int s = decltype(sc)::value; ring[s][0] = f2(s, 0); ring[s][1] = f2(s, 1); });
            static_for<HS / 2>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 2 < HS / 2) { ring[(s + 2) % 3][0] = f2(s + 2, 0); ring[(s + 2) % 3][1] = f2(s + 2, 1); }
                if constexpr (s < 4) {       // k-steps 0..11 only need layer 1's quarters 0-2
                    quad_words(acc[0][(2 * s) >> 2], (2 * s) & 3, s < 2 ? bt0 : bt1, h1p + 48 + 8 * ((2 * s) >> 2), qm);
                    quad_words(acc[0][(2 * s + 1) >> 2], (2 * s + 1) & 3, s < 2 ? bt0 : bt1, h1p + 48 + 8 * ((2 * s + 1) >> 2), qm);
                } else if constexpr (s == 4) {
                    store_line(h1p + 48, 6, h1row, ok, h);                                          // H1 tiles 6, 7
                }
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
            if (kW2LowPlane && lo_plane) {    // + W2lo h1 (wave_tile.h): value and tangent rows alike; fragments from memory (16 KB, cache-resident; LDS is full)
                uint32_t zlo = 0;
                asm volatile("" : "+v"(zlo));
                const bf16x8 *W2q = reinterpret_cast<const bf16x8 *>(W2f + kW2F) + lane + zlo;
                static_for<HS / 2>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s) * 64], frag_of(h1p + 4 * (2 * s)), y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s + 1) * 64], frag_of(h1p + 4 * (2 * s + 1)), y1, 0, 0, 0);
                });
            }
        }
        // ---- outputs: register i <-> output 8 (i >> 2) + 4 h + (i & 3); the bias belongs to the value row only
        if constexpr (WIDE) {
            const int K = d_out;
            const int64_t b = gr >> 2, Bp4 = M >> 2, Be = Bp4 - sp.n_main;
            const bool main_pt = b < sp.n_main;
            const int64_t e = b - sp.n_main;
            const int d = t - 1;
            // one 32-column tile of the outputs: its raw columns / its objects' gradient rows leave at once; what the choice between the tiles needs
            // (row minimum and index, this lane's output at the value row's index) is returned
            auto tile_out = [&](int base, const float (&o)[16], float &best, int &bi, float &at_hit, bool &mine) {
                best = INFINITY;
                bi = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                    if (base + n < K && o[i] < best) { best = o[i]; bi = n; }
                }
                {
                    const float ob = __shfl_xor(best, 32);
                    const int oi = __shfl_xor(bi, 32);
                    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                const int hit = __builtin_amdgcn_update_dpp(0, bi, 0x00, 0xf, 0xf, true);
                at_hit = 0.f;
                mine = ((hit >> 2) & 1) == h;
                const int ih = 4 * (hit >> 3) + (hit & 3);
#pragma unroll
                for (int i = 0; i < 16; i++) at_hit = i == ih ? o[i] : at_hit;
                if (ok) {
                    if (is_value) {
                        float *dst = (main_pt ? sp.sdf_raw + b * K : sp.y_eik + e * K) + base;
                        if ((K & 3) == 0) {
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                if (base + 8 * q + 4 * h < K) *reinterpret_cast<float4 *>(dst + 8 * q + 4 * h) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
                        } else {
#pragma unroll
                            for (int i = 0; i < 16; i++) {
                                const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                                if (base + n < K) dst[n] = o[i];
                            }
                        }
                    } else if (!main_pt) {
                        int64_t step = Be * 3;
                        asm volatile("" : "+s"(step));
                        float *gp = sp.grad_theta + ((int64_t)(base + 4 * h) * Be + e) * 3 + d;
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                            if (base + n < K) gp[(int64_t)(8 * (i >> 2) + (i & 3)) * step] = o[i];
                        }
                    }
                }
            };
            float o[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 bv = lds_at<f32x4>(bo, (uint32_t)(8 * q) * 4u);
#pragma unroll
                for (int k = 0; k < 4; k++) o[4 * q + k] = y0[4 * q + k] + y1[4 * q + k] + (is_value ? bv[k] : 0.f);
            }
            float best0, at0, best1, at1;
            int bi0, bi1;
            bool mine0, mine1;
            tile_out(0, o, best0, bi0, at0, mine0);
            {
                uint32_t zb = 0;
                asm volatile("" : "+v"(zb));
                const bf16x8 *W2q = reinterpret_cast<const bf16x8 *>(wide.W2b) + lane + zb;
                static_for<HS / 2>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (s == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[0], frag_of(h1p), zero, 0, 0, 0);
                        y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[64], frag_of(h1p + 4), zero, 0, 0, 0);
                    } else {
                        y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s) * 64], frag_of(h1p + 4 * (2 * s)), y0, 0, 0, 0);
                        y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2q[(size_t)(2 * s + 1) * 64], frag_of(h1p + 4 * (2 * s + 1)), y1, 0, 0, 0);
                    }
                });
                if (kW2LowPlane && lo_plane) {
                    const bf16x8 *W2r = W2q + (size_t)HS * 64;
                    static_for<HS / 2>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2r[(size_t)(2 * s) * 64], frag_of(h1p + 4 * (2 * s)), y0, 0, 0, 0);
                        y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2r[(size_t)(2 * s + 1) * 64], frag_of(h1p + 4 * (2 * s + 1)), y1, 0, 0, 0);
                    });
                }
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                o[i] = y0[i] + y1[i] + (is_value ? wide.bias2[512 + n] : 0.f);
            }
            tile_out(32, o, best1, bi1, at1, mine1);
            // the value row chooses (strict <: equal minima keep the lower index); the quad's tangent rows follow it
            const int w1 = best1 < best0 ? 1 : 0;
            const int wq = __builtin_amdgcn_update_dpp(0, w1, 0x00, 0xf, 0xf, true);
            if (ok) {
                if (is_value) {
                    if (h == 0) {
                        sp.idx[b] = w1 ? 32 + bi1 : bi0;
                        const float best = w1 ? best1 : best0;
                        if (main_pt) sp.sdf[b] = best; else sp.min_eik[e] = best;
                    }
                } else {
                    const float at = wq ? at1 : at0;
                    const bool mine = wq ? mine1 : mine0;
                    if (mine) {
                        if (main_pt) sp.grad[b * 3 + d] = at;
                        else sp.grad_theta[((int64_t)K * Be + e) * 3 + d] = at;
                    }
                }
            }
        } else if constexpr (!SPLIT) {
          if (ok) {
            float *dst = Y + gr * d_out;
            if ((d_out & 3) == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = 8 * q + 4 * h;
                    if (n < d_out) {
                        const f32x4 bv = lds_at<f32x4>(bo, (uint32_t)(8 * q) * 4u);
                        const float4 bb = is_value ? make_float4(bv[0], bv[1], bv[2], bv[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                        *reinterpret_cast<float4 *>(dst + n) = make_float4(y0[4 * q] + y1[4 * q] + bb.x, y0[4 * q + 1] + y1[4 * q + 1] + bb.y,
                                                                           y0[4 * q + 2] + y1[4 * q + 2] + bb.z, y0[4 * q + 3] + y1[4 * q + 3] + bb.w);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                    if (n < d_out) dst[n] = y0[i] + y1[i] + (is_value ? bias[512 + n] : 0.f);
                }
            }
          }
        } else {
            // ---- split outputs (hs_trunk_split_fwd's, encode_ops.hip): the K = d_out per-object SDFs of the value row, their minimum and its
            //      index (lowest among equals), the gradient of the minimum from the three tangent rows -- and for the Eikonal points every
            //      object's gradient.  A row's 32 outputs sit in two lanes (h = 0 / 1, 16 registers each); the quad = the point's four rows.
            const int K = d_out;
            const int64_t b = gr >> 2, Bp4 = M >> 2, Be = Bp4 - sp.n_main;
            float o[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 bv = lds_at<f32x4>(bo, (uint32_t)(8 * q) * 4u);
#pragma unroll
                for (int k = 0; k < 4; k++) o[4 * q + k] = y0[4 * q + k] + y1[4 * q + k] + (is_value ? bv[k] : 0.f);
            }
            float best = INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 16; i++) {          // this half's columns in increasing order: strict < keeps the lowest index
                const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                if (n < K && o[i] < best) { best = o[i]; bi = n; }
            }
            {
                const float ob = __shfl_xor(best, 32);
                const int oi = __shfl_xor(bi, 32);
                if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            const int hit = __builtin_amdgcn_update_dpp(0, bi, 0x00, 0xf, 0xf, true);      // the value row's index, on the quad's four lanes
            // this lane's output at column `hit` (if the column lives in this half)
            float at_hit = 0.f;
            const bool mine = ((hit >> 2) & 1) == h;
            const int ih = 4 * (hit >> 3) + (hit & 3);
#pragma unroll
            for (int i = 0; i < 16; i++) at_hit = i == ih ? o[i] : at_hit;
            if (ok) {
                const bool main_pt = b < sp.n_main;
                const int64_t e = b - sp.n_main;
                if (is_value) {
                    float *dst = main_pt ? sp.sdf_raw + b * K : sp.y_eik + e * K;
                    if ((K & 3) == 0) {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (8 * q + 4 * h < K) *reinterpret_cast<float4 *>(dst + 8 * q + 4 * h) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                            if (n < K) dst[n] = o[i];
                        }
                    }
                    if (h == 0) {
                        sp.idx[b] = bi;
                        if (main_pt) sp.sdf[b] = best; else sp.min_eik[e] = best;
                    }
                } else {
                    const int d = t - 1;
                    if (main_pt) {
                        if (mine) sp.grad[b * 3 + d] = at_hit;
                    } else {
                        // (column stride kept opaque: as a visible loop invariant the sixteen column offsets are hoisted out of the tile
                        //  loop into 32 registers, and this kernel has none to spare)
                        int64_t step = Be * 3;
                        asm volatile("" : "+s"(step));
                        float *gp = sp.grad_theta + ((int64_t)(4 * h) * Be + e) * 3 + d;
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int n = 8 * (i >> 2) + 4 * h + (i & 3);
                            if (n < K) gp[(int64_t)(8 * (i >> 2) + (i & 3)) * step] = o[i];
                        }
                        if (mine) sp.grad_theta[((int64_t)K * Be + e) * 3 + d] = at_hit;
                    }
                }
            }
        }
        HS_TSTAMP(4);
    }
    if (!resident) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int32_t hs_trunk_mlp2_input_column(int32_t c) {
    for (int h = 0; h < 2; h++)
        for (int j = 0; j < 40; j++)
            if (input_column(h, j) == c) return 40 * h + j;
    return -1;
}

int hs_trunk_mlp2_fwd(const float *x, const float *feat, const float *dydx, const void *W0f, const void *W1f, const void *W2f, const float *bias,
                      int32_t d_out, void *H0, void *H1, float *Y, void *Xp, int64_t M, float jac_scale, const hsTrunkSplit *split, int64_t ld, int32_t w2_planes,
                      void *stream) {
    if (d_out < 1 || d_out > 32 || (M & 3) || (ld != 0 && ld < (M >> 2)) || w2_planes < 1 || w2_planes > 2) return HS_ERR_ARG;
    if (ld == 0) ld = M >> 2;
    if (M == 0) return HS_OK;
    if (!x || !feat || !dydx || !W0f || !W1f || !W2f || !bias || !H0 || !H1 || (!Y && !split) || !Xp) return HS_ERR_NULL;
    hsTrunkSplit sp = {0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (split) {
        sp = *split;
        const int64_t Bp = M >> 2;
        if (sp.n_main < 0 || sp.n_main > Bp) return HS_ERR_ARG;
        if (!sp.idx || (sp.n_main > 0 && (!sp.sdf_raw || !sp.sdf || !sp.grad)) || (sp.n_main < Bp && (!sp.y_eik || !sp.min_eik || !sp.grad_theta))) return HS_ERR_NULL;
        Y = nullptr;       // the split outputs replace it
    }
    if ((const char *)W2f != (const char *)W1f + (size_t)kW1F * 2) return HS_ERR_ARG;
    const size_t lds = (size_t)(kW1F + kW2F) * sizeof(uint16_t) + kBias * sizeof(float);
    static hsLdsAttrOnce attr_a, attr_b;
    attr_a.set((const void *)k_trunk_fwd2<false>, (int)lds);
    attr_b.set((const void *)k_trunk_fwd2<true>, (int)lds);
    const int64_t ntiles = (M + kRows - 1) / kRows;
    constexpr bool spread = true;      // (false: packed workgroups, 24.8 against 20.6 us -- DESIGN 14.10)
    const int64_t want = spread ? ntiles : (ntiles + kWaves - 1) / kWaves;
    const int grid = (int)(want < 256 ? want : 256);
    // w2_planes = 1: the caller evaluates nothing but the Eikonal regulariser's points -- only their gradients are used, a training with those
    // points in fp32 ends where one with single-plane bf16 does (profiles/r05/bf16_stage_hunt.txt, stage "eikonal"), and the low plane's
    // fragments (from memory, one tile per wave, nothing to hide them under) cost that 4 096-point launch 6 of its 30 us
    const int lo_plane = w2_planes == 2;
    if (split)
        k_trunk_fwd2<true><<<grid, kThreadsW, lds, (hipStream_t)stream>>>(x, feat, dydx, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias, d_out,
                                                                 (uint16_t *)H0, (uint16_t *)H1, Y, (uint16_t *)Xp, M, jac_scale, sp, ld, lo_plane, TrunkWideArgs<false>{});
    else
        k_trunk_fwd2<false><<<grid, kThreadsW, lds, (hipStream_t)stream>>>(x, feat, dydx, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias, d_out,
                                                                 (uint16_t *)H0, (uint16_t *)H1, Y, (uint16_t *)Xp, M, jac_scale, sp, ld, lo_plane, TrunkWideArgs<false>{});
    return wt_check_launch();
}

int hs_trunk_mlp2_fwd_wide(const float *x, const float *feat, const float *dydx, const void *W0f, const void *W1f, const void *W2f, const float *bias,
                           const void *W2f_b, const float *bias_b, int32_t d_out, void *H0, void *H1, void *Xp, int64_t M, float jac_scale,
                           const hsTrunkSplit *split, int64_t ld, int32_t w2_planes, void *stream) {
    if (d_out < 33 || d_out > 64 || (M & 3) || (ld != 0 && ld < (M >> 2)) || w2_planes < 1 || w2_planes > 2) return HS_ERR_ARG;
    if (ld == 0) ld = M >> 2;
    if (M == 0) return HS_OK;
    if (!x || !feat || !dydx || !W0f || !W1f || !W2f || !bias || !W2f_b || !bias_b || !H0 || !H1 || !split || !Xp) return HS_ERR_NULL;
    const hsTrunkSplit sp = *split;
    const int64_t Bp = M >> 2;
    if (sp.n_main < 0 || sp.n_main > Bp) return HS_ERR_ARG;
    if (!sp.idx || (sp.n_main > 0 && (!sp.sdf_raw || !sp.sdf || !sp.grad)) || (sp.n_main < Bp && (!sp.y_eik || !sp.min_eik || !sp.grad_theta))) return HS_ERR_NULL;
    if ((const char *)W2f != (const char *)W1f + (size_t)kW1F * 2) return HS_ERR_ARG;
    const size_t lds = (size_t)(kW1F + kW2F) * sizeof(uint16_t) + kBias * sizeof(float);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_trunk_fwd2<true, true>, (int)lds);
    const int64_t ntiles = (M + kRows - 1) / kRows;
    constexpr bool spread = true;
    const int64_t want = spread ? ntiles : (ntiles + kWaves - 1) / kWaves;
    const int grid = (int)(want < 256 ? want : 256);
    k_trunk_fwd2<true, true><<<grid, kThreadsW, lds, (hipStream_t)stream>>>(x, feat, dydx, (const uint16_t *)W0f, (const uint16_t *)W1f, (const uint16_t *)W2f, bias,
                                                                            d_out, (uint16_t *)H0, (uint16_t *)H1, nullptr, (uint16_t *)Xp, M, jac_scale, sp, ld,
                                                                            w2_planes == 2, TrunkWideArgs<true>{(const uint16_t *)W2f_b, bias_b});
    return wt_check_launch();
}

}  // extern "C"
