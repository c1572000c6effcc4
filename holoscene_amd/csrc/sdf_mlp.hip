// holoscene_amd/csrc/sdf_mlp.hip -- fused SDF-trunk inference on the CDNA4 matrix cores (gfx950).
//
// The error-bounded sampler evaluates the scene SDF at up to 5 x 128 points per ray per iteration -- 85 % of all
// network evaluations of a Stage-1 step.  The reference runs ObjectImplicitNetworkGrid.forward for
// that (model/network.py:169-210, 305-311): ~20 ATen kernels per sweep, colour branch included.  This kernel does
// the whole SDF branch for a tile of 128 points without leaving the CU:
//
//   in  = [x, sin/cos(2^k x) k<6, hash features(32)]            71 values, zero-padded to 96   (model/embedder.py:11-36)
//   h0  = softplus100(W0 in + b0)   256                          (network.py:203-206, Softplus(beta=100) :163)
//   h1  = softplus100(W1 h0 + b1)   256
//   y   = W2 h1 + b2                d_out (<= 64)
//   out = min_k y_k  (or y_idx)                                   (network.py:305-311 / 316-318)
//
// bf16 operands, fp32 accumulation: v_mfma_f32_32x32x16_bf16.  The product is formed transposed, D[neuron][point] =
// W . H^T, so both operands are 16-byte row reads from row-major LDS images (weights [neuron][k], activations
// [point][k]) and each lane ends up with 4 consecutive neurons of ONE point per accumulator quad -> the epilogue
// (bias, softplus, bf16 pack) writes 8-byte runs straight back into the activation tile.
//   LDS: activation tile 128 x (256+8) bf16 = 66 KB (updated in place layer by layer) + double-buffered weight
//   chunks 2 x 256 x (32+8) bf16 = 40 KB + biases; the +8 row pitches make every ds_read_b128 conflict-free.
//   8 waves (2 per SIMD, so one wave's LDS latency hides under the other's MFMAs) = 4 neuron quarters x 2 point
//   halves: 64 neurons x 64 points per wave, 4 accumulator tiles; operand fragments are double-buffered in registers
//   one k-step ahead; weight chunks stream L2 -> registers -> LDS one chunk ahead.
//   Measured history (131 072 points, MI355X): 276 us (libm sinf/cosf staging) -> 132 us (hardware sincos) ->
//   see DESIGN.md for the current figure.
#include "launch_util.h"
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"
#define HS_MLP_KC 64   // 256-deep layers stream their weights in 64-deep chunks (half the barrier rounds); LDS: 66 + 72 KB
#include "mfma_mlp.h"

namespace {

constexpr int NPE = 39;           // 3 + 6*6 positional-encoding values
constexpr int NFEAT = 32;

// Softplus(beta=100) in the SCALED domain the inference kernel works in.  With t = 100*log2(e) * v:
//   softplus100(v) = ln2/100 * log2(1 + 2^t)
// The factor 100*log2(e) is folded into W0 and the biases (W1 then needs none: its input carries 100/ln2, its output wants
// 100*log2(e), and the two cancel), ln2/100 into W2 -- all at weight-packing time -- so an element costs
// bias-add, v_exp_f32, add, v_log_f32 and one select instead of also three multiplies and a min.  Above t = 30 log2(1 + 2^t) == t in fp32
// (PyTorch's linear branch starts at 100 v = 20, i.e. t = 28.9, where the two differ by 3e-9 relative).
constexpr float kActScale = 100.f * 1.44269504f;    // applied to W0, b0, b1 (hs_sdf_mlp_fwd scales the biases itself)
__device__ __forceinline__ float softplus_scaled(float t) {
#ifdef HS_EXP_NO_EPILOGUE
    return t;
#endif
    const float l = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(fminf(t, 64.f)));
    return t > 30.f ? t : l;
}

// bias + softplus + bf16 pack, written back into the activation tile (all waves have passed the barrier that ends layer_mma)
__device__ __forceinline__ void epilogue_softplus(const float *bias_lds, uint16_t *H, f32x16 acc[2][2], int nq, int ph, int lane) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n0 = nq * 64 + nt * 32 + q * 8 + 4 * (lane >> 5);  // 4 consecutive neurons
            const float4 bi = *reinterpret_cast<const float4 *>(bias_lds + n0);
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const int p = ph * 64 + pt * 32 + (lane & 31);
                const float v0 = softplus_scaled(acc[nt][pt][q * 4 + 0] + bi.x), v1 = softplus_scaled(acc[nt][pt][q * 4 + 1] + bi.y);
                const float v2 = softplus_scaled(acc[nt][pt][q * 4 + 2] + bi.z), v3 = softplus_scaled(acc[nt][pt][q * 4 + 3] + bi.w);
                uint2 pk;
                pk.x = pack_bf16(v0, v1);
                pk.y = pack_bf16(v2, v3);
                *reinterpret_cast<uint2 *>(H + (size_t)p * HP + n0) = pk;
            }
        }
    }
}

template <int NOUT_TILES>  // d_out padded to 32 * NOUT_TILES
__global__ __launch_bounds__(kThreads) void k_sdf_mlp(const float *__restrict__ x, const float *__restrict__ feat, const uint16_t *__restrict__ W0,
                                                       const float *__restrict__ b0, const uint16_t *__restrict__ W1,
                                                       const float *__restrict__ b1, const uint16_t *__restrict__ W2,
                                                       const float *__restrict__ b2, int d_out, int select, uint64_t select_mask, float *__restrict__ out_min,
                                                       float *__restrict__ out_raw, int64_t B, hsGate gate, int feat_level_major) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    if (gate.a != nullptr && !(*gate.a > *gate.b)) return;
    uint16_t *H = lds;                                  // [BM][HP]
    uint16_t *Wc = lds + (size_t)BM * HP;               // 2 x [HID][WP]  (re-used for W2 [32*NOUT_TILES][HP] in the last layer)
    float *bias = reinterpret_cast<float *>(Wc + 2 * (size_t)HID * WP);   // b0[256] b1[256] b2[64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = wave & 3, ph = wave >> 2;
    if (threadIdx.x < HID) { bias[threadIdx.x] = b0[threadIdx.x] * kActScale; bias[HID + threadIdx.x] = b1[threadIdx.x] * kActScale; }   // scaled domain
    if (threadIdx.x < 64) bias[2 * HID + threadIdx.x] = (int)threadIdx.x < d_out ? b2[threadIdx.x] : 0.f;
    const int64_t ntiles = (B + BM - 1) / BM;
    // this thread's share of a tile's inputs (4 threads per point: coordinates + 8 of the 32 features), fetched one tile ahead so
    // that the staging phase below never waits on global memory (it used to expose a full load latency per tile)
    const int sp = threadIdx.x & (BM - 1), spart = threadIdx.x / BM;   // spart 0..3
    float xv[3];
    float4 fv[2];
    auto fetch = [&](int64_t tile_) {
        const int64_t gp = tile_ * BM + sp;
        const bool ok = tile_ < ntiles && gp < B;
        xv[0] = ok ? x[gp * 3] : 0.f; xv[1] = ok ? x[gp * 3 + 1] : 0.f; xv[2] = ok ? x[gp * 3 + 2] : 0.f;
        if (feat_level_major) {   // feat [16, B, 2]: what the hash kernel writes fully coalesced (8 bytes per lane, lanes = points)
            const float2 *fl = reinterpret_cast<const float2 *>(feat) + (size_t)(spart * 4) * B + (ok ? gp : 0);
            const float2 z2 = make_float2(0.f, 0.f);
            const float2 a = ok ? fl[0] : z2, b = ok ? fl[B] : z2, c = ok ? fl[2 * B] : z2, e = ok ? fl[3 * B] : z2;
            fv[0] = make_float4(a.x, a.y, b.x, b.y);
            fv[1] = make_float4(c.x, c.y, e.x, e.y);
        } else {
            const float4 *fp = reinterpret_cast<const float4 *>(feat + (ok ? gp : 0) * NFEAT + spart * 8);
            fv[0] = ok ? fp[0] : make_float4(0.f, 0.f, 0.f, 0.f);
            fv[1] = ok ? fp[1] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" ::: "memory");  // keep the per-tile weight loads inside the loop (hoisted they would pin ~90 VGPRs)
        const int64_t p0 = tile * BM;
        // ---- input features: H[p][0..K0).  sin/cos pairs from the hardware v_sin/v_cos units
        //      (arguments <= 2^5 * 1.75 rad, well inside their range; the bf16 destination keeps 8 bits anyway).
        {
            const int part = spart;
            uint16_t *row = H + (size_t)sp * HP;
            if (part == 0) { row[0] = (uint16_t)f2bf(xv[0]); row[1] = (uint16_t)f2bf(xv[1]); row[2] = (uint16_t)f2bf(xv[2]); }
            if (part < 3) {
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const int k = part * 2 + kk;
                    const float f = (float)(1 << k);
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        float sn, cs;
                        __sincosf(xv[d] * f, &sn, &cs);
                        row[3 + 6 * k + d] = (uint16_t)f2bf(sn);
                        row[3 + 6 * k + 3 + d] = (uint16_t)f2bf(cs);
                    }
                }
            } else {
#pragma unroll
                for (int c = NPE + NFEAT; c < K0; c++) row[c] = 0;
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const float4 v = fv[i];
                uint16_t *dst = row + NPE + part * 8 + 4 * i;   // odd column -> 2-byte stores
                dst[0] = (uint16_t)f2bf(v.x); dst[1] = (uint16_t)f2bf(v.y); dst[2] = (uint16_t)f2bf(v.z); dst[3] = (uint16_t)f2bf(v.w);
            }
        }
        fetch(tile + gridDim.x);   // in flight under the three layers of this tile
        __syncthreads();
        f32x16 acc[2][2];
        zero_acc(acc);
        layer_mma<HP, 32>(W0, K0, K0, H, Wc, acc, nq, ph, lane);
        epilogue_softplus(bias, H, acc, nq, ph, lane);
        __syncthreads();
        zero_acc(acc);
        layer_mma(W1, HID, HID, H, Wc, acc, nq, ph, lane);
        epilogue_softplus(bias + HID, H, acc, nq, ph, lane);
        // ---- layer 2: stage W2 [32*NOUT_TILES][HID] with pitch HP into the chunk area
        // (W2 = two bf16 planes [2][32 NOUT_TILES][HID]: the matrix, then what its rounding dropped -- the last layer's rows are a large common
        //  value plus small learned structure that one plane loses; both planes fit the chunk area)
        static_assert(2 * 32 * NOUT_TILES * HP <= 2 * HID * WP, "both W2 planes must fit the weight-chunk area (holds for HS_MLP_KC = 64; not for the header's default 32)");
        for (int idx = threadIdx.x; idx < 2 * 32 * NOUT_TILES * (HID / 8); idx += kThreads) {
            const int row = idx / (HID / 8), seg = idx - row * (HID / 8);
            *reinterpret_cast<uint4 *>(Wc + (size_t)row * HP + seg * 8) = *reinterpret_cast<const uint4 *>(W2 + (size_t)row * HID + seg * 8);
        }
        __syncthreads();
        if (wave < kRowWaves) {  // 4 point tiles of 32; waves 4..7 have nothing to do in the narrow last layer
            f32x16 y[NOUT_TILES];
#pragma unroll
            for (int t = 0; t < NOUT_TILES; t++)
#pragma unroll
                for (int i = 0; i < 16; i++) y[t][i] = 0.f;
            const int prow = wave * 32 + (lane & 31);
#pragma unroll 4
            for (int ks = 0; ks < HID / 16; ks++) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(H + (size_t)prow * HP + ks * 16 + (lane >> 5) * 8);
#pragma unroll
                for (int t = 0; t < NOUT_TILES; t++) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(Wc + (size_t)(t * 32 + (lane & 31)) * HP + ks * 16 + (lane >> 5) * 8);
                    const bf16x8 al = *reinterpret_cast<const bf16x8 *>(Wc + (size_t)((NOUT_TILES + t) * 32 + (lane & 31)) * HP + ks * 16 + (lane >> 5) * 8);
                    y[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, y[t], 0, 0, 0);
                    y[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, y[t], 0, 0, 0);
                }
            }
            // lane holds, for point prow, neurons t*32 + (i&3) + 8*(i>>2) + 4*(lane>>5)
            const int64_t gp = p0 + prow;
            float best = INFINITY;
            // raw outputs: parked in this point's own (now dead) activation row -- each wave reads only its 32 rows of H, and its
            // LDS operations retire in order -- and stored coalesced by the whole workgroup below (lane-wise 4-byte stores at a
            // d_out*4-byte stride made the raw sweep 4.5x slower than the min sweep)
            float *park = reinterpret_cast<float *>(H + (size_t)prow * HP);
#pragma unroll
            for (int t = 0; t < NOUT_TILES; t++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int n = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    if (n < d_out) {
                        const float v = y[t][i] + bias[2 * HID + n];
                        if (out_raw) park[n] = v;
                        // scene minimum (select < 0, no mask), one object (select >= 0), or the minimum over an object subset (mask)
                        const bool take = select_mask ? ((select_mask >> n) & 1ull) != 0ull : (select < 0 || n == select);
                        if (take) best = fminf(best, v);
                    }
                }
            const float other = __shfl_xor(best, 32);
            best = fminf(best, other);  // for `select`, exactly one half holds the value, the other +inf
            if (lane < 32 && gp < B) out_min[gp] = best;
        }
        if (out_raw) {
            __syncthreads();
            for (int idx = threadIdx.x; idx < BM * d_out; idx += kThreads) {
                const int row = idx / d_out, n = idx - row * d_out;
                if (p0 + row < B) out_raw[(p0 + row) * d_out + n] = reinterpret_cast<const float *>(H + (size_t)row * HP)[n];
            }
        }
        __syncthreads();  // H and the chunk area are rewritten by the next tile
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Training trunk: the same three layers over value+Jacobian rows (4 rows per point: value, d/dx, d/dy, d/dz -- DESIGN
// V1), keeping what the backward pass needs.  Row r of a tile is row type r & 3; the accumulator layout puts the four
// rows of one point in the four lanes of one DPP quad, so the tangent rule  out_d = sigmoid(100 v) * acc_d  is a
// quad broadcast of the value lane's sigmoid -- no LDS round trip.
//   X  [M][K0] bf16 (k_trunk_input, zero-padded columns)      H0, H1 [M][256] bf16 = layer OUTPUTS (value row: softplus,
//   tangent rows: s * pre-activation; hs_softplus_tangent_bwd_h consumes exactly that)     Y [M][d_out] fp32
// Softplus and its derivative are needed for the VALUE row only, but the four rows of a point sit in the four lanes of a quad,
// so a lane-wise evaluation would run the exp/log/rcp sequence on all four lanes and use one.  Instead lane j of the quad takes
// neuron j of the value lane's four consecutive neurons (4 DPP broadcasts + select), evaluates ONE softplus pair, and the
// results travel back by quad_perm [j,j,j,j]: a quarter of the transcendental work for 12 DPP moves per 4 elements
// (ablation: the lane-wise form cost 106 us of k_trunk_fwd's 328).
__device__ __forceinline__ void tangent_quad(const float a[4], const float4 bi, int lane, bool is_value, float out[4]) {
#ifdef HS_EXP_NO_EPILOGUE
    out[0] = a[0] + bi.x; out[1] = a[1] + bi.y; out[2] = a[2] + bi.z; out[3] = a[3] + bi.w;
    return;
#endif
    const float t0 = quad_bcast0(a[0] + bi.x), t1 = quad_bcast0(a[1] + bi.y), t2 = quad_bcast0(a[2] + bi.z), t3 = quad_bcast0(a[3] + bi.w);
    const int j = lane & 3;
    const float v = j == 0 ? t0 : (j == 1 ? t1 : (j == 2 ? t2 : t3));   // pre-activation of neuron j of the point's value row
    const float t = v * 100.f;
    const float e = __builtin_amdgcn_exp2f(fminf(t, 20.f) * 1.44269504f);
    const float one_e = 1.f + e;
    const bool lin = t > 20.f;
    const float sp = lin ? v : __builtin_amdgcn_logf(one_e) * (0.69314718f * 0.01f);
    const float ds = lin ? 1.f : e * __builtin_amdgcn_rcpf(one_e);
    const float s0 = dpp_quad<0x00>(ds), s1 = dpp_quad<0x55>(ds), s2 = dpp_quad<0xAA>(ds), s3 = dpp_quad<0xFF>(ds);
    const float p0 = dpp_quad<0x00>(sp), p1 = dpp_quad<0x55>(sp), p2 = dpp_quad<0xAA>(sp), p3 = dpp_quad<0xFF>(sp);
    out[0] = is_value ? p0 : s0 * a[0];
    out[1] = is_value ? p1 : s1 * a[1];
    out[2] = is_value ? p2 : s2 * a[2];
    out[3] = is_value ? p3 : s3 * a[3];
}

__device__ __forceinline__ void epilogue_tangent(const float *bias_lds, uint16_t *H, f32x16 acc[2][2], int nq, int ph, int lane) {
    const bool is_value = (lane & 3) == 0;
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n0 = nq * 64 + nt * 32 + q * 8 + 4 * (lane >> 5);
            const float4 bi = *reinterpret_cast<const float4 *>(bias_lds + n0);
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const int p = ph * 64 + pt * 32 + (lane & 31);
                const float a[4] = {acc[nt][pt][q * 4 + 0], acc[nt][pt][q * 4 + 1], acc[nt][pt][q * 4 + 2], acc[nt][pt][q * 4 + 3]};
                float o[4];
                tangent_quad(a, bi, lane, is_value, o);
                uint2 pk;
                pk.x = pack_bf16(o[0], o[1]);
                pk.y = pack_bf16(o[2], o[3]);
                *reinterpret_cast<uint2 *>(H + (size_t)p * HP + n0) = pk;
            }
        }
    }
}

template <int NOUT_TILES>
__global__ __launch_bounds__(kThreads) void k_trunk_fwd(const uint16_t *__restrict__ X, const uint16_t *__restrict__ W0, const float *__restrict__ b0,
                                                         const uint16_t *__restrict__ W1, const float *__restrict__ b1,
                                                         const uint16_t *__restrict__ W2, const float *__restrict__ b2, int d_out,
                                                         uint16_t *__restrict__ H0, uint16_t *__restrict__ H1, float *__restrict__ Y, int64_t M,
                                                         const float *__restrict__ xs, const float *__restrict__ feat, const float *__restrict__ dydx,
                                                         uint16_t *__restrict__ Xout, int L, int C, float jac_scale) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *H = lds;
    uint16_t *Wc = lds + (size_t)BM * HP;
    float *bias = reinterpret_cast<float *>(Wc + 2 * (size_t)HID * WP);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = wave & 3, ph = wave >> 2;
    if (threadIdx.x < HID) { bias[threadIdx.x] = b0[threadIdx.x]; bias[HID + threadIdx.x] = b1[threadIdx.x]; }
    if (threadIdx.x < 64) bias[2 * HID + threadIdx.x] = (int)threadIdx.x < d_out ? b2[threadIdx.x] : 0.f;
    const int64_t ntiles = (M + BM - 1) / BM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" ::: "memory");
        const int64_t r0 = tile * BM;
        if (X) {
            for (int idx = threadIdx.x; idx < BM * (K0 / 8); idx += kThreads) {
                const int row = idx / (K0 / 8), seg = idx - row * (K0 / 8);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (r0 + row < M) v = *reinterpret_cast<const uint4 *>(X + (size_t)(r0 + row) * K0 + seg * 8);
                *reinterpret_cast<uint4 *>(H + (size_t)row * HP + seg * 8) = v;
            }
            __syncthreads();
        } else {
            // ---- build the 4-row input in place (what hs_trunk_input_fwd writes): value row [x, sin/cos(2^k x), features],
            //      tangent row d its derivative w.r.t. x_d (unit vector | +-2^k cos/sin on component d | dy_dx * d(x01)/dx)
            const int64_t Bp = M >> 2, pb = r0 >> 2;
            for (int idx = threadIdx.x; idx < (BM / 4) * K0; idx += kThreads) {
                const int pt = idx / K0, c = idx - pt * K0;
                const int64_t b = pb + pt;
                float v0 = 0.f, t[3] = {0.f, 0.f, 0.f};
                if (b < Bp && c < NPE + NFEAT) {
                    if (c < 3) {
                        v0 = xs[b * 3 + c];
                        t[c] = 1.f;
                    } else if (c < NPE) {
                        const int k = (c - 3) / 6, r = (c - 3) - 6 * k, d = r % 3;
                        const float f = (float)(1 << k);
                        float sn, cs;
                        __sincosf(xs[b * 3 + d] * f, &sn, &cs);
                        if (r < 3) { v0 = sn; t[d] = f * cs; }
                        else { v0 = cs; t[d] = -f * sn; }
                    } else {
                        const int lc = c - NPE, l = lc / C, ch = lc - l * C;
                        v0 = feat[b * NFEAT + lc];
                        const float *j = dydx + ((int64_t)l * Bp + b) * 3 * C + ch;
                        t[0] = j[0] * jac_scale; t[1] = j[C] * jac_scale; t[2] = j[2 * C] * jac_scale;
                    }
                }
                uint16_t *col = H + (size_t)(4 * pt) * HP + c;
                col[0] = (uint16_t)f2bf(v0); col[HP] = (uint16_t)f2bf(t[0]); col[2 * HP] = (uint16_t)f2bf(t[1]); col[3 * HP] = (uint16_t)f2bf(t[2]);
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < BM * (K0 / 8); idx += kThreads) {   // kept for the weight gradient of layer 0
                const int row = idx / (K0 / 8), seg = idx - row * (K0 / 8);
                if (r0 + row < M) *reinterpret_cast<uint4 *>(Xout + (size_t)(r0 + row) * K0 + seg * 8) = *reinterpret_cast<const uint4 *>(H + (size_t)row * HP + seg * 8);
            }
        }
        f32x16 acc[2][2];
        zero_acc(acc);
        layer_mma<HP, 32>(W0, K0, K0, H, Wc, acc, nq, ph, lane);
        epilogue_tangent(bias, H, acc, nq, ph, lane);
        __syncthreads();
        store_tile(H, H0, r0, M);   // reads of H; the next epilogue's writes sit behind layer_mma's barriers
        zero_acc(acc);
        layer_mma(W1, HID, HID, H, Wc, acc, nq, ph, lane);
        epilogue_tangent(bias + HID, H, acc, nq, ph, lane);
        // (W2 = two bf16 planes [2][32 NOUT_TILES][HID]: the matrix, then what its rounding dropped -- the last layer's rows are a large common
        //  value plus small learned structure that one plane loses; both planes fit the chunk area)
        static_assert(2 * 32 * NOUT_TILES * HP <= 2 * HID * WP, "both W2 planes must fit the weight-chunk area (holds for HS_MLP_KC = 64; not for the header's default 32)");
        for (int idx = threadIdx.x; idx < 2 * 32 * NOUT_TILES * (HID / 8); idx += kThreads) {
            const int row = idx / (HID / 8), seg = idx - row * (HID / 8);
            *reinterpret_cast<uint4 *>(Wc + (size_t)row * HP + seg * 8) = *reinterpret_cast<const uint4 *>(W2 + (size_t)row * HID + seg * 8);
        }
        __syncthreads();
        store_tile(H, H1, r0, M);
        if (wave < kRowWaves) {
            f32x16 y[NOUT_TILES];
#pragma unroll
            for (int t = 0; t < NOUT_TILES; t++)
#pragma unroll
                for (int i = 0; i < 16; i++) y[t][i] = 0.f;
            const int prow = wave * 32 + (lane & 31);
#pragma unroll 4
            for (int ks = 0; ks < HID / 16; ks++) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(H + (size_t)prow * HP + ks * 16 + (lane >> 5) * 8);
#pragma unroll
                for (int t = 0; t < NOUT_TILES; t++) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(Wc + (size_t)(t * 32 + (lane & 31)) * HP + ks * 16 + (lane >> 5) * 8);
                    const bf16x8 al = *reinterpret_cast<const bf16x8 *>(Wc + (size_t)((NOUT_TILES + t) * 32 + (lane & 31)) * HP + ks * 16 + (lane >> 5) * 8);
                    y[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, y[t], 0, 0, 0);
                    y[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, y[t], 0, 0, 0);
                }
            }
            const int64_t gr = r0 + prow;
            const bool is_value = (lane & 3) == 0;
            if (gr < M) {
#pragma unroll
                for (int t = 0; t < NOUT_TILES; t++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int n = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                        if (n < d_out) Y[gr * d_out + n] = y[t][i] + (is_value ? bias[2 * HID + n] : 0.f);
                    }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward data path of the training trunk: cotangent of the outputs -> cotangents of both hidden pre-activations,
// one kernel.  Per 128-row tile:   G1 = g . W2        (matrix cores, K = padded d_out)
//                                  gA1 = softplus-tangent backward of G1 at H1     (epilogue, in LDS)    -> global
//                                  G0 = gA1 . W1 ;  gA0 = ... at H0                                       -> global
// The library path needs 2 GEMM launches + 2 elementwise launches and moves 2.2 GB for this (DESIGN); here every
// operand is read once and every result written once (0.9 GB).  Weight operands arrive TRANSPOSED ([in][out] row-major),
// so the same D = W . H^T machinery applies.  The value-row rule  gA_v = s*g_v + 100(1-s) * sum_d H_d*g_d  needs the three
// tangent lanes of the quad.
//
// Four consecutive neurons of one row (the lane's 8-byte cell of the tile): G = cotangents of the layer OUTPUTS, hv = the four outputs
// (bf16), result = cotangents of the pre-activations.  Lane j of the quad evaluates the value row's neuron j ONCE (1 - sigmoid and its
// slope from h = softplus100(v): one exponential per four elements instead of one per element on every lane of the quad), and the
// cross-lane reads are folded into the consuming multiply / add as DPP operands (hand-written: the compiler emits v_mov_dpp + the
// arithmetic separately; see trunk_mlp2.hip's forward epilogue).  9.7 k -> 6.5 k cycles per layer and 128-row tile (tools/exp/tbwd_prof.hip).
// LDS byte address of a pointer into the dynamic shared segment, and an 8-row MFMA fragment (rows r0..r0+7 of the lane's column) by two
// transposing reads `step` bytes (4 rows) apart
__device__ __forceinline__ uint32_t lds_addr_of(const uint16_t *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint16_t *)p; }
__device__ __forceinline__ bf16x8 tr_frag(uint32_t addr, uint32_t step) {
    uint2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\t"
                 "ds_read_b64_tr_b16 %1, %3\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "v"(addr + step)
                 : "memory");
    const uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
    return *reinterpret_cast<const bf16x8 *>(w);
}

struct BwdMasks { uint64_t n1, n2, n3; };      // lanes whose index in the quad is NOT 1 / 2 / 3 (wave masks in SGPR pairs)
__device__ __forceinline__ BwdMasks bwd_masks() {
    BwdMasks m = {0xddddddddddddddddull, 0xbbbbbbbbbbbbbbbbull, 0x7777777777777777ull};
    asm volatile("" : "+s"(m.n1), "+s"(m.n2), "+s"(m.n3));
    return m;
}
#define HS_DPPQ(k) " quad_perm:[" #k "," #k "," #k "," #k "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void bwd_act4(const float G[4], uint2 hv, const BwdMasks &m, bool is_value, float out[4]) {
    const float h[4] = {__uint_as_float(hv.x << 16), __uint_as_float(hv.x & 0xffff0000u), __uint_as_float(hv.y << 16), __uint_as_float(hv.y & 0xffff0000u)};
#ifdef HS_EXP_NO_EPILOGUE
    for (int k = 0; k < 4; k++) out[k] = G[k] + h[k];
    return;
#endif
    // lane j takes the value lane's h[j] (first read compiler-visible: it covers the hazards of a freshly written source)
    float hj = dpp_quad<0x00>(h[0]);
    asm("s_mov_b64 vcc, %4\n\t"
        "v_cndmask_b32_dpp %0, %1, %0, vcc" HS_DPPQ(0)
        "s_mov_b64 vcc, %5\n\t"
        "v_cndmask_b32_dpp %0, %2, %0, vcc" HS_DPPQ(0)
        "s_mov_b64 vcc, %6\n\t"
        "v_cndmask_b32_dpp %0, %3, %0, vcc" HS_DPPQ(0)
        : "+v"(hj)
        : "v"(h[1]), "v"(h[2]), "v"(h[3]), "s"(m.n1), "s"(m.n2), "s"(m.n3)
        : "vcc");
    const float e = __builtin_amdgcn_exp2f(hj * (-100.f * 1.44269504f));      // 1 - sigmoid(100 v) from h = softplus100(v)
    const float sj = 1.f - e, cj = 100.f * e;
    float p[4], dot[4];
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = h[k] * G[k];
    // dot_k = sum over the three tangent lanes of h_k * G_k (every lane of the quad gets it; only the value lane uses it)
    asm("s_nop 1\n\t"
        "v_mov_b32_dpp %0, %4" HS_DPPQ(1)
        "v_mov_b32_dpp %1, %5" HS_DPPQ(1)
        "v_mov_b32_dpp %2, %6" HS_DPPQ(1)
        "v_mov_b32_dpp %3, %7" HS_DPPQ(1)
        "v_add_f32_dpp %0, %4, %0" HS_DPPQ(2)
        "v_add_f32_dpp %1, %5, %1" HS_DPPQ(2)
        "v_add_f32_dpp %2, %6, %2" HS_DPPQ(2)
        "v_add_f32_dpp %3, %7, %3" HS_DPPQ(2)
        "v_add_f32_dpp %0, %4, %0" HS_DPPQ(3)
        "v_add_f32_dpp %1, %5, %1" HS_DPPQ(3)
        "v_add_f32_dpp %2, %6, %2" HS_DPPQ(3)
        "v_add_f32_dpp %3, %7, %3" HS_DPPQ(3)
        : "=&v"(dot[0]), "=&v"(dot[1]), "=&v"(dot[2]), "=&v"(dot[3])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));
#pragma unroll
    for (int k = 0; k < 4; k++) dot[k] = is_value ? dot[k] : 0.f;
    // out_k = s_k G_k (+ c_k dot_k on the value lane), s_k / c_k from lane k
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %4, %6" HS_DPPQ(0)
        "v_mul_f32_dpp %1, %4, %7" HS_DPPQ(1)
        "v_mul_f32_dpp %2, %4, %8" HS_DPPQ(2)
        "v_mul_f32_dpp %3, %4, %9" HS_DPPQ(3)
        "v_fmac_f32_dpp %0, %5, %10" HS_DPPQ(0)
        "v_fmac_f32_dpp %1, %5, %11" HS_DPPQ(1)
        "v_fmac_f32_dpp %2, %5, %12" HS_DPPQ(2)
        "v_fmac_f32_dpp %3, %5, %13" HS_DPPQ(3)
        : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3])
        : "v"(sj), "v"(cj), "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(G[3]), "v"(dot[0]), "v"(dot[1]), "v"(dot[2]), "v"(dot[3]));
}
#undef HS_DPPQ

// H holds the layer-output tile on entry and the pre-activation cotangent tile on exit (same element, same lane: in place)
__device__ __forceinline__ void epilogue_bwd(uint16_t *H, f32x16 acc[2][2], int nq, int ph, int lane) {
    const bool is_value = (lane & 3) == 0;
    const BwdMasks bm = bwd_masks();
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n0 = nq * 64 + nt * 32 + q * 8 + 4 * (lane >> 5);
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const int p = ph * 64 + pt * 32 + (lane & 31);
                uint2 *cell = reinterpret_cast<uint2 *>(H + (size_t)p * HP + n0);
                const float G[4] = {acc[nt][pt][q * 4 + 0], acc[nt][pt][q * 4 + 1], acc[nt][pt][q * 4 + 2], acc[nt][pt][q * 4 + 3]};
                float v[4];
                bwd_act4(G, *cell, bm, is_value, v);
                uint2 pk;
                pk.x = pack_bf16(v[0], v[1]);
                pk.y = pack_bf16(v[2], v[3]);
                *cell = pk;
            }
        }
    }
}

#ifndef HS_TBWD_DEEP
#define HS_TBWD_DEEP true       // weight chunks two rounds ahead (mfma_mlp.h: layer_mma): 357-376 -> 344-352 us per launch, same box
#endif
#ifdef HS_TBWD_PROFILE     // tools/exp/tbwd_prof.hip: s_memtime stamps of the phases of one tile (the third of every workgroup)
__device__ unsigned long long g_tbwd_prof[256 * 16];
#define HS_BSTAMP(i) do { if (threadIdx.x == 0 && tile == (int64_t)blockIdx.x + 2 * (int64_t)gridDim.x) g_tbwd_prof[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HS_BSTAMP(i) do { } while (0)
#endif

template <int KP>  // padded d_out: 32 or 64
__global__ __launch_bounds__(kThreads) void k_trunk_bwd(const uint16_t *__restrict__ g, const uint16_t *__restrict__ H1, const uint16_t *__restrict__ H0,
                                                         const uint16_t *__restrict__ W2t, const uint16_t *__restrict__ W1t,
                                                         uint16_t *__restrict__ gA1, uint16_t *__restrict__ gA0, float *__restrict__ gb1,
                                                         float *__restrict__ gb0, const uint16_t *__restrict__ W0t, float *__restrict__ g_feat,
                                                         float *__restrict__ g_dydx, int L, int C, float jac_scale, int64_t M,
                                                         float *__restrict__ gb2, float *__restrict__ dW2_part, int64_t ld) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *H = lds;
    uint16_t *Wc = lds + (size_t)BM * HP;
    constexpr int GP = KP + 8;                                   // row pitch of the side copy of the output-cotangent tile
    uint16_t *Gs = Wc + 2 * (size_t)HID * WP;                     // [BM][GP], only when dW2_part != NULL (the launcher sizes the LDS)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = wave & 3, ph = wave >> 2;
    float sum1 = 0.f, sum0 = 0.f, sum2 = 0.f;
    // last layer's weight gradient dW2 = g^T . H1 (reduction over the ROWS): both operand tiles sit in LDS once H1 has been staged, so the
    // wave owning 32 of the 256 columns accumulates its [KP x 32] block over all tiles of the workgroup in registers (16 VGPRs per 32
    // outputs) instead of a library GEMM re-reading H1 (214 MB) and g.  The fragments run along the rows of row-major tiles, hence 2-byte
    // LDS reads (8 per operand and k-step): ~3 % more work in this kernel for one GEMM launch (66 us) less.
    constexpr int kWaves = kThreads / 64, kColPasses = HID / (32 * kWaves);     // 32-column blocks of dW2 per wave: 1 (8 waves) or 2 (4 waves, BM = 64)
    f32x16 accW[kColPasses][KP / 32];
#pragma unroll
    for (int cp = 0; cp < kColPasses; cp++)
#pragma unroll
        for (int mt = 0; mt < KP / 32; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) accW[cp][mt][i] = 0.f;
    const int64_t ntiles = (M + BM - 1) / BM;
    // The H0 tile (64 KB) is requested early, right after the K = 32 product -- requested where it is consumed, each layer-output tile
    // costs ~8 k cycles of exposed wait (tools/exp/tbwd_prof.hip).  Requesting the NEXT tile's H1 under the input-cotangent phase as well
    // measured the same in isolation (353-357 vs 354 us) but keeps 32 more registers live across the loop: 22 spilled registers and
    // 140 MB of scratch traffic per launch in the production build; not done.
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" ::: "memory");
        HS_BSTAMP(0);
        const int64_t r0 = tile * BM;
        for (int idx = threadIdx.x; idx < BM * (KP / 8); idx += kThreads) {
            const int row = idx / (KP / 8), seg = idx - row * (KP / 8);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (r0 + row < M) v = *reinterpret_cast<const uint4 *>(g + (size_t)(r0 + row) * KP + seg * 8);
            *reinterpret_cast<uint4 *>(H + (size_t)row * HP + seg * 8) = v;
            if (dW2_part) *reinterpret_cast<uint4 *>(Gs + (size_t)row * GP + seg * 8) = v;
        }
        __syncthreads();
        if (gb2) {   // last layer's bias gradient: column sums of the VALUE rows of the output cotangent (thread = column x row group)
            const int col = threadIdx.x % KP, rg = threadIdx.x / KP;
            for (int r = 4 * rg; r < BM; r += 4 * (kThreads / KP)) sum2 += __uint_as_float((uint32_t)H[(size_t)r * HP + col] << 16);
        }
        f32x16 acc[2][2];
        HS_BSTAMP(1);
        TileRegs hr1 = load_tile_regs(H1, r0, M);      // in flight under the K = 32 product
        zero_acc(acc);
        layer_mma<HP, 32, HID, HS_TBWD_DEEP>(W2t, KP, KP, H, Wc, acc, nq, ph, lane);
        store_tile_regs(H, hr1);
#ifndef HS_TBWD_H0_LATE
        TileRegs hr = load_tile_regs(H0, r0, M);    // in flight under the epilogue, the gA1 store and the 256-deep product
#endif
        __syncthreads();
        HS_BSTAMP(2);
        if (dW2_part) {   // H = H1 tile, Gs = g tile: accW[kout][col] += sum_rows g[row][kout] * H1[row][col], this wave's 32 columns
            // Both operands run along the ROWS of row-major tiles.  gfx950's transposing LDS read does that gather: in every group of 16
            // lanes, lane L passes the address of 4 consecutive bf16 and lane i receives element i & 3 of the slots of lanes (i >> 2) + 4 j,
            // j = 0..3 (tools/exp/tr_b16_sem.hip) -- with lane L pointing at tile[r0 + (L >> 2)][c0 + 4 (L & 3)], lane i gets rows
            // r0..r0+3 of column c0 + i.  Two reads per 8-row fragment instead of eight 2-byte reads and seven shifts / ors
            // (this phase: 4.9 k -> see tools/exp/tbwd_prof.hip).
            const int L16 = lane & 15, cg = (lane >> 4) & 1;
            const uint32_t hb = lds_addr_of(H + (size_t)((lane >> 5) * 8 + (L16 >> 2)) * HP + wave * 32 + 16 * cg + 4 * (L16 & 3));
            const uint32_t gb_ = lds_addr_of(Gs + (size_t)((lane >> 5) * 8 + (L16 >> 2)) * GP + 16 * cg + 4 * (L16 & 3));
#pragma unroll 2
            for (int ks = 0; ks < BM / 16; ks++) {
#pragma unroll
                for (int cp = 0; cp < kColPasses; cp++) {
                    const bf16x8 bfrag = tr_frag(hb + (uint32_t)((ks * 16 * HP + cp * 32 * kWaves) * 2), 4 * HP * 2);
#pragma unroll
                    for (int mt = 0; mt < KP / 32; mt++) {
                        const bf16x8 afrag = tr_frag(gb_ + (uint32_t)((ks * 16 * GP + mt * 32) * 2), 4 * GP * 2);
                        accW[cp][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, accW[cp][mt], 0, 0, 0);
                    }
                }
            }
            __syncthreads();   // the epilogue below rewrites H in place
        }
        HS_BSTAMP(3);
        epilogue_bwd(H, acc, nq, ph, lane);
        __syncthreads();
        HS_BSTAMP(4);
        store_tile(H, gA1, r0, M);
        sum1 += tile_colsum<4>(H);
        HS_BSTAMP(5);
#ifdef HS_TBWD_H0_LATE
        TileRegs hr = load_tile_regs(H0, r0, M);
#endif
        zero_acc(acc);
        layer_mma<HP, KC, HID, HS_TBWD_DEEP>(W1t, HID, HID, H, Wc, acc, nq, ph, lane);
        store_tile_regs(H, hr);
        __syncthreads();
        HS_BSTAMP(6);
        epilogue_bwd(H, acc, nq, ph, lane);
        __syncthreads();
        HS_BSTAMP(7);
        store_tile(H, gA0, r0, M);
        sum0 += tile_colsum<4>(H);
        HS_BSTAMP(8);
        if (W0t) {
            // ---- cotangent of the trunk input: gX = gA0 . W0 (K0 = 96 columns; W0t = W0^T zero-padded to 256 rows).  Only the
            //      waves owning neurons < 96 multiply; the others keep streaming weight chunks.  Saves the library GEMM's
            //      second read of gA0 (214 MB at M = 417 792).
            zero_acc(acc);
            // rows >= 128 of W0^T are padding and are not streamed.  (Staging the live 128 x 256 part in ONE round -- it fits the two chunk
            // buffers -- instead of four 64-deep ones was slower: 9.5 k vs 6.6 k cycles, the whole load latency exposed at once.)
            layer_mma<HP, KC, 128, HS_TBWD_DEEP>(W0t, HID, HID, H, Wc, acc, nq, ph, lane, nq < 2);
            HS_BSTAMP(10);
            if (nq < 2) {
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    if (nq * 64 + nt * 32 >= K0) continue;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int n0 = nq * 64 + nt * 32 + q * 8 + 4 * (lane >> 5);
#pragma unroll
                        for (int pt = 0; pt < 2; pt++) {
                            const int p = ph * 64 + pt * 32 + (lane & 31);
                            uint2 pk;
                            pk.x = pack_bf16(acc[nt][pt][q * 4 + 0], acc[nt][pt][q * 4 + 1]);
                            pk.y = pack_bf16(acc[nt][pt][q * 4 + 2], acc[nt][pt][q * 4 + 3]);
                            *reinterpret_cast<uint2 *>(H + (size_t)p * HP + n0) = pk;
                        }
                    }
                }
            }
            __syncthreads();
            HS_BSTAMP(11);
            // the hash-feature columns of that tile ARE the cotangents the table scatter consumes (hs_trunk_input_bwd's slicing):
            // value rows -> g_feat [L, B, C] (level-major); tangent row d -> g_dydx [L, B, 3*C] scaled by d(x01)/dx.  Coalesced fp32 runs.
            const int64_t Bp = M >> 2, pb = r0 >> 2;     // points in total / first point of this tile (BM/4 points per tile)
            const int LC = L * C;
            if (C == 2) {       // the stock grid: every divisor below a compile-time constant (the generic loops spend 7 k cycles per tile
                                // in integer divisions by (BM / 4) * C and 3 * C -- as much as the product itself, tools/exp/tbwd_prof.hip)
                constexpr int P = BM / 4;
                for (int idx = threadIdx.x; idx < P * NFEAT; idx += kThreads) {          // [L, B, 2]: 2 P floats per level
                    const int l = idx / (2 * P), rem = idx % (2 * P), pt = rem >> 1, c = rem & 1;
                    if (pb + pt < Bp) g_feat[((size_t)l * ld + pb + pt) * 2 + c] = __uint_as_float((uint32_t)H[(size_t)(4 * pt) * HP + NPE + l * 2 + c] << 16);
                }
                for (int idx = threadIdx.x; idx < (NFEAT / 2) * P * 6; idx += kThreads) {  // [L, B, 3, 2]: 6 P floats per level
                    const int l = idx / (6 * P), rem = idx % (6 * P), pt = rem / 6, dc = rem % 6, d = dc >> 1, c = dc & 1;
                    if (pb + pt < Bp)
                        g_dydx[((size_t)l * ld + pb + pt) * 6 + dc] = jac_scale * __uint_as_float((uint32_t)H[(size_t)(4 * pt + 1 + d) * HP + NPE + l * 2 + c] << 16);
                }
            } else {
            for (int idx = threadIdx.x; idx < (BM / 4) * LC; idx += kThreads) {   // level-major [L, B, C]
                const int l = idx / ((BM / 4) * C), rem = idx - l * ((BM / 4) * C), pt = rem / C, c = rem - pt * C;
                if (pb + pt < Bp)
                    g_feat[((size_t)l * ld + pb + pt) * C + c] = __uint_as_float((uint32_t)H[(size_t)(4 * pt) * HP + NPE + l * C + c] << 16);
            }
            const int run = (BM / 4) * 3 * C;
            for (int idx = threadIdx.x; idx < L * run; idx += kThreads) {
                const int l = idx / run, rem = idx - l * run, pt = rem / (3 * C), dc = rem - pt * (3 * C), d = dc / C, c = dc - d * C;
                if (pb + pt < Bp)
                    g_dydx[((size_t)l * ld + pb + pt) * (3 * C) + dc] =
                        jac_scale * __uint_as_float((uint32_t)H[(size_t)(4 * pt + 1 + d) * HP + NPE + l * C + c] << 16);
            }
            }
        }
        __syncthreads();
        HS_BSTAMP(9);
    }
    if (dW2_part) {   // this workgroup's slice [KP][256]; lane: column wave*32 + (lane & 31), outputs (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
        float *dst = dW2_part + (size_t)blockIdx.x * KP * HID + wave * 32 + (lane & 31);
#pragma unroll
        for (int cp = 0; cp < kColPasses; cp++)
#pragma unroll
            for (int mt = 0; mt < KP / 32; mt++)
#pragma unroll
                for (int i = 0; i < 16; i++) dst[(size_t)(mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * HID + cp * 32 * kWaves] = accW[cp][mt][i];
    }
    if (threadIdx.x < HID) {
        if (gb1) unsafeAtomicAdd(gb1 + threadIdx.x, sum1);
        if (gb0) unsafeAtomicAdd(gb0 + threadIdx.x, sum0);
    }
    if (gb2) {   // workgroup reduction first: 256 workgroups x 512 same-address atomics would serialise for ~0.2 ms
        float *red = reinterpret_cast<float *>(lds);
        __syncthreads();
        red[threadIdx.x] = sum2;
        __syncthreads();
        if (threadIdx.x < KP) {
            float t = 0.f;
            for (int i = threadIdx.x; i < kThreads; i += KP) t += red[i];
            unsafeAtomicAdd(gb2 + threadIdx.x, t);
        }
    }
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

}  // namespace

extern "C" {

int hs_sdf_mlp_fwd(const float *x, const float *feat, const void *W0, const float *b0, const void *W1, const float *b1, const void *W2,
                   const float *b2, int32_t d_out, int32_t select, uint64_t select_mask, float *out_min, float *out_raw, int64_t B, const hsGate *gate,
                   int32_t feat_level_major,
                   void *stream) {
    if (d_out < 1 || d_out > 64 || select >= d_out) return HS_ERR_ARG;
    if (select_mask && d_out < 64 && (select_mask >> d_out)) return HS_ERR_ARG;   // a bit beyond the last object
    if (B == 0) return HS_OK;
    if (!x || !feat || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !out_min) return HS_ERR_NULL;
    const size_t lds = ((size_t)BM * HP + 2 * (size_t)HID * WP) * sizeof(uint16_t) + (2 * HID + 64) * sizeof(float);
    const int64_t ntiles = (B + BM - 1) / BM;
    const int grid = (int)(ntiles < kGridCap ? ntiles : kGridCap);  // one workgroup per CU (111 KB LDS), tiles strided across the grid
    hipStream_t st = (hipStream_t)stream;
    if (d_out <= 32) {
        static hsLdsAttrOnce attr1;
        attr1.set((const void *)k_sdf_mlp<1>, (int)lds);
        k_sdf_mlp<1><<<grid, kThreads, lds, st>>>(x, feat, (const uint16_t *)W0, b0, (const uint16_t *)W1, b1, (const uint16_t *)W2, b2, d_out, select, select_mask,
                                                   out_min, out_raw, B, gate ? *gate : hsGate{nullptr, nullptr}, feat_level_major);
    } else {
        static hsLdsAttrOnce attr2;
        attr2.set((const void *)k_sdf_mlp<2>, (int)lds);
        k_sdf_mlp<2><<<grid, kThreads, lds, st>>>(x, feat, (const uint16_t *)W0, b0, (const uint16_t *)W1, b1, (const uint16_t *)W2, b2, d_out, select, select_mask,
                                                   out_min, out_raw, B, gate ? *gate : hsGate{nullptr, nullptr}, feat_level_major);
    }
    return check_launch();
}

int hs_trunk_mlp_fwd(const void *X, const void *W0, const float *b0, const void *W1, const float *b1, const void *W2, const float *b2,
                     int32_t d_out, void *H0, void *H1, float *Y, int64_t M, const float *x, const float *feat, const float *dydx, void *Xout, int32_t L,
                     int32_t C, float jac_scale, void *stream) {
    if (d_out < 1 || d_out > 64 || (M & 3)) return HS_ERR_ARG;
    if (M == 0) return HS_OK;
    if (!X && (L < 1 || C < 1 || L * C != NFEAT)) return HS_ERR_ARG;
    if ((!X && (!x || !feat || !dydx || !Xout)) || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !H0 || !H1 || !Y) return HS_ERR_NULL;
    const size_t lds = ((size_t)BM * HP + 2 * (size_t)HID * WP) * sizeof(uint16_t) + (2 * HID + 64) * sizeof(float);
    const int64_t ntiles = (M + BM - 1) / BM;
    const int grid = (int)(ntiles < kGridCap ? ntiles : kGridCap);
    hipStream_t st = (hipStream_t)stream;
    if (d_out <= 32) {
        static hsLdsAttrOnce attr1;
        attr1.set((const void *)k_trunk_fwd<1>, (int)lds);
        k_trunk_fwd<1><<<grid, kThreads, lds, st>>>((const uint16_t *)X, (const uint16_t *)W0, b0, (const uint16_t *)W1, b1, (const uint16_t *)W2, b2,
                                                     d_out, (uint16_t *)H0, (uint16_t *)H1, Y, M, x, feat, dydx, (uint16_t *)Xout, L, C, jac_scale);
    } else {
        static hsLdsAttrOnce attr2;
        attr2.set((const void *)k_trunk_fwd<2>, (int)lds);
        k_trunk_fwd<2><<<grid, kThreads, lds, st>>>((const uint16_t *)X, (const uint16_t *)W0, b0, (const uint16_t *)W1, b1, (const uint16_t *)W2, b2,
                                                     d_out, (uint16_t *)H0, (uint16_t *)H1, Y, M, x, feat, dydx, (uint16_t *)Xout, L, C, jac_scale);
    }
    return check_launch();
}

int32_t hs_trunk_bwd_parts(int64_t M) {
    const int64_t ntiles = (M + BM - 1) / BM;
    return (int32_t)(ntiles < kGridCap ? ntiles : kGridCap);
}

int hs_trunk_mlp_bwd(const void *g, int32_t g_pitch, const void *H1, const void *H0, const void *W2t, const void *W1t, void *gA1, void *gA0,
                     float *gb1, float *gb0, const void *W0t, float *g_feat, float *g_dydx, int32_t L, int32_t C, float jac_scale, int64_t M,
                     float *gb2, float *dW2_part, int64_t ld, void *stream) {
    if ((g_pitch != 32 && g_pitch != 64) || (M & 3) || (ld != 0 && ld < (M >> 2))) return HS_ERR_ARG;
    if (ld == 0) ld = M >> 2;       /* points per level of the g_feat / g_dydx buffers (>= M / 4 when they hold more points than this call's) */
    if (M == 0) return HS_OK;
    if (W0t && (L < 1 || C < 1 || L * C != NFEAT)) return HS_ERR_ARG;
    if (!g || !H1 || !H0 || !W2t || !W1t || !gA1 || !gA0 || (W0t && (!g_feat || !g_dydx))) return HS_ERR_NULL;
    const size_t lds = ((size_t)BM * HP + 2 * (size_t)HID * WP + (dW2_part ? (size_t)BM * (g_pitch + 8) : 0)) * sizeof(uint16_t);
    const int64_t ntiles = (M + BM - 1) / BM;
    const int grid = (int)(ntiles < kGridCap ? ntiles : kGridCap);   // == hs_trunk_bwd_parts(M)
    hipStream_t st = (hipStream_t)stream;
    if (g_pitch == 32) {
        static hsLdsAttrOnce attr1;
        attr1.set((const void *)k_trunk_bwd<32>, (int)lds);
        k_trunk_bwd<32><<<grid, kThreads, lds, st>>>((const uint16_t *)g, (const uint16_t *)H1, (const uint16_t *)H0, (const uint16_t *)W2t,
                                                      (const uint16_t *)W1t, (uint16_t *)gA1, (uint16_t *)gA0, gb1, gb0, (const uint16_t *)W0t, g_feat, g_dydx, L, C, jac_scale, M, gb2, dW2_part, ld);
    } else {
        static hsLdsAttrOnce attr2;
        attr2.set((const void *)k_trunk_bwd<64>, (int)lds);
        k_trunk_bwd<64><<<grid, kThreads, lds, st>>>((const uint16_t *)g, (const uint16_t *)H1, (const uint16_t *)H0, (const uint16_t *)W2t,
                                                      (const uint16_t *)W1t, (uint16_t *)gA1, (uint16_t *)gA0, gb1, gb0, (const uint16_t *)W0t, g_feat, g_dydx, L, C, jac_scale, M, gb2, dW2_part, ld);
    }
    return check_launch();
}

}  // extern "C"
