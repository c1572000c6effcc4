// holoscene_amd/csrc/appearance_mlp.hip -- the colour branch of a rendered sample, fused on the CDNA4 matrix cores (gfx950): the 128-point
// WORKGROUP-TILE form of rounds 2-3.  The benchmarked path runs the wave-tile form (appearance2.hip); this one stays as the tests' independent second
// implementation of the same function (tests/test_model_gpu.py: appear2 vs appear, ReLU ballots) and for direct calls to hs_appearance_fwd / _bwd.
//
// Reference (per rendered point, model/network.py):
//   fv  = Linear(256,256)(ReLU(Linear(32,256)(colour hash features)))                   :186-188  (colour_grid_feature_map_mlp :99-101)
//   x   = [posenc4(point), posenc4(view dir), posenc4(normal), fv]            337       :586-596
//   rgb = sigmoid(W2 ReLU(W1 ReLU(W0 x + b0) + b1) + b2)                      337->256->256->3      :598-612
// = 5 GEMMs + 2 concatenations + 5 elementwise passes forward and ~20 launches backward in the reference's formulation, each
// bouncing a [100 352, 256] activation through HBM.  Here one kernel per direction walks a 128-point tile through all five
// layers (csrc/mfma_mlp.h machinery): the feature vector never leaves LDS between the colour MLP and the rendering MLP; the
// 81 positional-encoding inputs sit in a second small LDS tile and enter layer R0 as a second matrix product into the same
// accumulators.  What the weight-gradient GEMMs (library, split-M) need is written once: layer outputs forward, pre-activation
// cotangents backward.
//
//   k_appear_fwd : featc, points, dirs, normals -> rgb;  keeps xin = [featc | posenc] , hc, fv, r0, r1 (bf16)
//   k_appear_bwd : d rgb -> g_y, gA_r1, gA_r0, g_fv, gA_hc (bf16), d normals, d featc (fp32), bias gradients
#include "launch_util.h"
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"
#include "mfma_mlp.h"

namespace {

constexpr int NFC = 32;        // colour hash features per point (L*C)
constexpr int NF = 4;          // posenc frequencies of the rendering network (multires_view = 4)
constexpr int NPE1 = 3 + 6 * NF;   // 27 values per encoded vector
constexpr int PW = 96;         // 3 x 27 = 81 posenc inputs, zero-padded
constexpr int PP = PW + 8;     // LDS pitch of the posenc tile
constexpr int XW = NFC + PW;   // row of the saved input image: [featc(32) | posenc(96)]
constexpr int SP = 33;         // fp32 scratch pitch

// ReLU masks: the backward only needs the SIGN of the three ReLU layers' outputs, so the forward leaves them as wave ballots -- ballot
// k = ((nt 4 + q) 2 + pt) 4 + j of a wave covers its 64 lanes' element j of cell (nt, q, pt), the same decomposition both kernels'
// epilogues walk -- 4 KB per 128-point tile and layer instead of the 64 KB bf16 tile the backward used to read back (3 x 51 MB per
// iteration).  Lane k of the wave keeps ballot k and stores it as one 8-byte word: masks[((tile 3 + layer) 8 + wave) 64 + lane].
template <int ACT>  // 0: linear, 1: ReLU
__device__ __forceinline__ void epilogue_act(const float *bias_lds, uint16_t *H, f32x16 acc[2][2], int nq, int ph, int lane, uint64_t *mask_out = nullptr) {
    uint32_t mlo = 0u, mhi = 0u;
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n0 = nq * 64 + nt * 32 + q * 8 + 4 * (lane >> 5);
            const float4 bi = *reinterpret_cast<const float4 *>(bias_lds + n0);
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const int p = ph * 64 + pt * 32 + (lane & 31);
                float v0 = acc[nt][pt][q * 4 + 0] + bi.x, v1 = acc[nt][pt][q * 4 + 1] + bi.y;
                float v2 = acc[nt][pt][q * 4 + 2] + bi.z, v3 = acc[nt][pt][q * 4 + 3] + bi.w;
                if (ACT == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                uint2 pk;
                pk.x = pack_bf16(v0, v1);
                pk.y = pack_bf16(v2, v3);
                *reinterpret_cast<uint2 *>(H + (size_t)p * HP + n0) = pk;
                if (ACT == 1 && mask_out != nullptr) {     // positive AFTER the bf16 rounding: exactly what the backward's old test on the stored tile saw
                    const int k = ((nt * 4 + q) * 2 + pt) * 4;
                    const bool on[4] = {(int16_t)(pk.x & 0xffffu) > 0, (int32_t)pk.x > 0xffff, (int16_t)(pk.y & 0xffffu) > 0, (int32_t)pk.y > 0xffff};
                    const uint64_t b0 = __builtin_amdgcn_ballot_w64(on[0]), b1 = __builtin_amdgcn_ballot_w64(on[1]);
                    const uint64_t b2 = __builtin_amdgcn_ballot_w64(on[2]), b3 = __builtin_amdgcn_ballot_w64(on[3]);
                    // (s_nop 3: a ballot is a VALU write of an SGPR pair, and v_writelane reading it needs four wait states that the
                    //  assembler does not insert inside an asm block -- without them 22 % of the mask bits were stale)
                    asm("s_nop 3\n\t"
                        "v_writelane_b32 %0, %2, %10\n\t"
                        "v_writelane_b32 %1, %3, %10\n\t"
                        "v_writelane_b32 %0, %4, %11\n\t"
                        "v_writelane_b32 %1, %5, %11\n\t"
                        "v_writelane_b32 %0, %6, %12\n\t"
                        "v_writelane_b32 %1, %7, %12\n\t"
                        "v_writelane_b32 %0, %8, %13\n\t"
                        "v_writelane_b32 %1, %9, %13"
                        : "+v"(mlo), "+v"(mhi)
                        : "s"((uint32_t)b0), "s"((uint32_t)(b0 >> 32)), "s"((uint32_t)b1), "s"((uint32_t)(b1 >> 32)), "s"((uint32_t)b2),
                          "s"((uint32_t)(b2 >> 32)), "s"((uint32_t)b3), "s"((uint32_t)(b3 >> 32)), "n"(k), "n"(k + 1), "n"(k + 2), "n"(k + 3));
                }
            }
        }
    }
    if (ACT == 1 && mask_out != nullptr) mask_out[lane] = ((uint64_t)mhi << 32) | mlo;
}

// MASK = 1: H holds the layer's ReLU output r on entry; exit: (r > 0 ? acc : 0) in place.  MASK = 0: H = acc.
// MASK = 2: the ReLU mask comes from the forward's ballots (epilogue_act): `mine` = this lane's word of the wave's 64, ballot k read back
// with v_readlane and applied as the select's lane mask
template <int MASK>
__device__ __forceinline__ void epilogue_grad(uint16_t *H, f32x16 acc[2][2], int nq, int ph, int lane, uint64_t mine = 0) {
    const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n0 = nq * 64 + nt * 32 + q * 8 + 4 * (lane >> 5);
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const int p = ph * 64 + pt * 32 + (lane & 31);
                uint2 *cell = reinterpret_cast<uint2 *>(H + (size_t)p * HP + n0);
                float v0 = acc[nt][pt][q * 4 + 0], v1 = acc[nt][pt][q * 4 + 1], v2 = acc[nt][pt][q * 4 + 2], v3 = acc[nt][pt][q * 4 + 3];
                if (MASK == 1) {   // bf16 bit patterns: positive <=> sign clear and not zero
                    const uint2 r = *cell;
                    v0 = (int16_t)(r.x & 0xffffu) > 0 ? v0 : 0.f;
                    v1 = (int32_t)r.x > 0xffff ? v1 : 0.f;
                    v2 = (int16_t)(r.y & 0xffffu) > 0 ? v2 : 0.f;
                    v3 = (int32_t)r.y > 0xffff ? v3 : 0.f;
                }
                if (MASK == 2) {
                    const int k = ((nt * 4 + q) * 2 + pt) * 4;
                    float *vv[4] = {&v0, &v1, &v2, &v3};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(mhi, k + j) << 32) | (uint32_t)__builtin_amdgcn_readlane(mlo, k + j);
                        float o;
                        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o) : "v"(*vv[j]), "s"(m));
                        *vv[j] = o;
                    }
                }
                uint2 pk;
                pk.x = pack_bf16(v0, v1);
                pk.y = pack_bf16(v2, v3);
                *cell = pk;
            }
        }
    }
}

// narrow layers (<= 32 outputs): W [32][HID] bf16 staged with pitch HP into the chunk area; waves 0..3 take 32 rows each
__device__ __forceinline__ void stage_small(uint16_t *Wc, const uint16_t *__restrict__ W) {
    for (int idx = threadIdx.x; idx < 32 * (HID / 8); idx += kThreads) {
        const int row = idx / (HID / 8), seg = idx - row * (HID / 8);
        *reinterpret_cast<uint4 *>(Wc + (size_t)row * HP + seg * 8) = *reinterpret_cast<const uint4 *>(W + (size_t)row * HID + seg * 8);
    }
}

// lane result: row wave*32 + (lane&31), outputs n = (i&3) + 8*(i>>2) + 4*(lane>>5), i < 16
__device__ __forceinline__ f32x16 small_mma(const uint16_t *Wc, const uint16_t *H, int wave, int lane) {
    f32x16 y;
#pragma unroll
    for (int i = 0; i < 16; i++) y[i] = 0.f;
    const int prow = wave * 32 + (lane & 31);
#pragma unroll 4
    for (int ks = 0; ks < HID / 16; ks++) {
        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(H + (size_t)prow * HP + ks * 16 + (lane >> 5) * 8);
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(Wc + (size_t)(lane & 31) * HP + ks * 16 + (lane >> 5) * 8);
        y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, y, 0, 0, 0);
    }
    return y;
}

__global__ __launch_bounds__(kThreads) void k_appear_fwd(const float *__restrict__ featc, const float *__restrict__ pts, const float *__restrict__ dirs,
                                                          const float *__restrict__ nrm, const uint16_t *__restrict__ Wc0,
                                                          const uint16_t *__restrict__ Wc1, const uint16_t *__restrict__ Wr0f,
                                                          const uint16_t *__restrict__ Wr0p, const uint16_t *__restrict__ Wr1,
                                                          const uint16_t *__restrict__ Wr2, const float *__restrict__ bc0, const float *__restrict__ bc1,
                                                          const float *__restrict__ br0, const float *__restrict__ br1, const float *__restrict__ br2,
                                                          uint16_t *__restrict__ xin, uint16_t *__restrict__ hc, uint16_t *__restrict__ fv,
                                                          uint16_t *__restrict__ r0o, uint16_t *__restrict__ r1o, float *__restrict__ rgb, int64_t B, uint64_t *__restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *H = lds;                                   // [BM][HP]
    uint16_t *Wc = H + (size_t)BM * HP;                  // 2 x [HID][WP]
    uint16_t *P = Wc + 2 * (size_t)HID * WP;             // [BM][PP]
    float *bias = reinterpret_cast<float *>(P + (size_t)BM * PP);   // bc0 | bc1 | br0 | br1 | br2(4)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = wave & 3, ph = wave >> 2;
    if (threadIdx.x < HID) {
        bias[threadIdx.x] = bc0[threadIdx.x]; bias[HID + threadIdx.x] = bc1[threadIdx.x];
        bias[2 * HID + threadIdx.x] = br0[threadIdx.x]; bias[3 * HID + threadIdx.x] = br1[threadIdx.x];
    }
    if (threadIdx.x < 4) bias[4 * HID + threadIdx.x] = threadIdx.x < 3 ? br2[threadIdx.x] : 0.f;
    const int64_t ntiles = (B + BM - 1) / BM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" ::: "memory");
        const int64_t p0 = tile * BM;
        {   // ---- inputs: parts 0..2 encode one vector each into P, part 3 converts the colour features into H[:, 0:32)
            const int p = threadIdx.x & (BM - 1), part = threadIdx.x / BM;
            const int64_t gp = p0 + p;
            const bool ok = gp < B;
            if (part < 3) {
                const float *src = part == 0 ? pts : (part == 1 ? dirs : nrm);
                float v[3] = {0.f, 0.f, 0.f};
                if (ok) { v[0] = src[gp * 3]; v[1] = src[gp * 3 + 1]; v[2] = src[gp * 3 + 2]; }
                uint16_t *row = P + (size_t)p * PP + part * NPE1;
                row[0] = (uint16_t)f2bf(v[0]); row[1] = (uint16_t)f2bf(v[1]); row[2] = (uint16_t)f2bf(v[2]);
#pragma unroll
                for (int k = 0; k < NF; k++) {
                    const float f = (float)(1 << k);
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        float sn, cs;
                        __sincosf(v[d] * f, &sn, &cs);
                        row[3 + 6 * k + d] = (uint16_t)f2bf(sn);
                        row[3 + 6 * k + 3 + d] = (uint16_t)f2bf(cs);
                    }
                }
            } else {
                uint16_t *prow = P + (size_t)p * PP;
#pragma unroll
                for (int c = 3 * NPE1; c < PW; c++) prow[c] = 0;
                // featc is level-major [16, B, 2] (the hash kernel's coalesced store order): 8 bytes per level, consecutive points adjacent
                const float2 *fl = reinterpret_cast<const float2 *>(featc) + (ok ? gp : 0);
                uint16_t *hrow = H + (size_t)p * HP;
#pragma unroll
                for (int i = 0; i < NFC / 4; i++) {
                    const float2 a = ok ? fl[(size_t)(2 * i) * B] : make_float2(0.f, 0.f), b = ok ? fl[(size_t)(2 * i + 1) * B] : make_float2(0.f, 0.f);
                    uint2 pk;
                    pk.x = pack_bf16(a.x, a.y);
                    pk.y = pack_bf16(b.x, b.y);
                    *reinterpret_cast<uint2 *>(hrow + 4 * i) = pk;
                }
            }
        }
        __syncthreads();
        // saved input image for the weight gradients: xin[p] = [featc(32) | posenc(96)], 16 x 16 B per row
        for (int idx = threadIdx.x; idx < BM * (XW / 8); idx += kThreads) {
            const int row = idx / (XW / 8), seg = idx - row * (XW / 8);
            if (p0 + row < B) {
                const uint4 v = seg < NFC / 8 ? *reinterpret_cast<const uint4 *>(H + (size_t)row * HP + seg * 8)
                                              : *reinterpret_cast<const uint4 *>(P + (size_t)row * PP + (seg - NFC / 8) * 8);
                *reinterpret_cast<uint4 *>(xin + (size_t)(p0 + row) * XW + seg * 8) = v;
            }
        }
        f32x16 acc[2][2];
        zero_acc(acc);
        layer_mma(Wc0, NFC, NFC, H, Wc, acc, nq, ph, lane);
        ChunkRegs<KC> wn = first_chunk(Wc1, HID);     // the next layer's first weight chunk rides under this layer's epilogue and store
        uint64_t *mk = masks ? masks + ((size_t)tile * 3 * 8 + wave) * 64 : nullptr;      // this wave's words of layer 0 (layers 1, 2: + 512, + 1024)
        epilogue_act<1>(bias, H, acc, nq, ph, lane, mk);
        __syncthreads();
        store_tile(H, hc, p0, B);
        zero_acc(acc);
        layer_mma(Wc1, HID, HID, H, Wc, acc, nq, ph, lane, true, &wn);
        wn = first_chunk(Wr0f, HID);
        epilogue_act<0>(bias + HID, H, acc, nq, ph, lane);
        __syncthreads();
        store_tile(H, fv, p0, B);
        zero_acc(acc);
        layer_mma(Wr0f, HID, HID, H, Wc, acc, nq, ph, lane, true, &wn);   // feature-vector columns 81..336 of W_R0
        layer_mma<PP>(Wr0p, PW, PW, P, Wc, acc, nq, ph, lane);        // + positional-encoding columns 0..80
        wn = first_chunk(Wr1, HID);
        epilogue_act<1>(bias + 2 * HID, H, acc, nq, ph, lane, mk ? mk + 512 : nullptr);
        __syncthreads();
        store_tile(H, r0o, p0, B);
        zero_acc(acc);
        layer_mma(Wr1, HID, HID, H, Wc, acc, nq, ph, lane, true, &wn);
        epilogue_act<1>(bias + 3 * HID, H, acc, nq, ph, lane, mk ? mk + 1024 : nullptr);
        stage_small(Wc, Wr2);
        __syncthreads();
        store_tile(H, r1o, p0, B);
        if (wave < kRowWaves) {
            const f32x16 y = small_mma(Wc, H, wave, lane);
            const int64_t gp = p0 + wave * 32 + (lane & 31);
            if (lane < 32 && gp < B) {   // outputs 0..2 live in accumulator entries 0..2 of the lower half-wave
#pragma unroll
                for (int n = 0; n < 3; n++) rgb[gp * 3 + n] = 1.f / (1.f + __expf(-(y[n] + bias[4 * HID + n])));
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kThreads) void k_appear_bwd(const float *__restrict__ g_rgb, const float *__restrict__ rgb, const float *__restrict__ nrm,
                                                          const uint16_t *__restrict__ r1, const uint16_t *__restrict__ r0, const uint16_t *__restrict__ hc,
                                                          const uint16_t *__restrict__ Wr2t, const uint16_t *__restrict__ Wr1t,
                                                          const uint16_t *__restrict__ Wr0ft, const uint16_t *__restrict__ Wr0nt,
                                                          const uint16_t *__restrict__ Wc1t, const uint16_t *__restrict__ Wc0t,
                                                          uint16_t *__restrict__ gy, uint16_t *__restrict__ gA_r1, uint16_t *__restrict__ gA_r0,
                                                          uint16_t *__restrict__ g_fv, uint16_t *__restrict__ gA_hc, float *__restrict__ d_nrm,
                                                          float *__restrict__ g_featc, float *__restrict__ gb, int64_t B, const uint64_t *__restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *H = lds;
    uint16_t *Wc = H + (size_t)BM * HP;
    float *S = reinterpret_cast<float *>(Wc + 2 * (size_t)HID * WP);   // [BM][SP] fp32 scratch for the narrow products
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = wave & 3, ph = wave >> 2;
    float s_r1 = 0.f, s_r0 = 0.f, s_c1 = 0.f, s_c0 = 0.f, s_y = 0.f;
    const int64_t ntiles = (B + BM - 1) / BM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("" ::: "memory");
        const int64_t p0 = tile * BM;
        {   // ---- cotangent of the pre-sigmoid outputs, padded to 32 columns: one 16-byte segment per thread
            const int row = threadIdx.x >> 2, seg = threadIdx.x & 3;
            const int64_t gp = p0 + row;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (seg == 0 && gp < B) {
                float g[3];
#pragma unroll
                for (int n = 0; n < 3; n++) {
                    const float c = rgb[gp * 3 + n];
                    g[n] = g_rgb[gp * 3 + n] * c * (1.f - c);
                }
                v.x = pack_bf16(g[0], g[1]);
                v.y = pack_bf16(g[2], 0.f);
            }
            *reinterpret_cast<uint4 *>(H + (size_t)row * HP + seg * 8) = v;
            if (gp < B) *reinterpret_cast<uint4 *>(gy + (size_t)gp * 32 + seg * 8) = v;
        }
        __syncthreads();
        {   // output-layer bias gradient: column sums of the 3 live columns (thread = column x row group)
            const int col = threadIdx.x & 3, rg = threadIdx.x >> 2;
            if (col < 3)
                for (int r = rg; r < BM; r += kThreads / 4) s_y += __uint_as_float((uint32_t)H[(size_t)r * HP + col] << 16);
        }
        f32x16 acc[2][2];
        // ReLU masks of the three masked layers: with the forward's ballots (masks != NULL) three 8-byte words per lane replace three
        // 64 KB tiles of saved layer outputs per workgroup tile (and their trip through registers and LDS)
        const uint64_t *mk = masks ? masks + ((size_t)tile * 3 * 8 + wave) * 64 + lane : nullptr;
        const uint64_t m_hc = mk ? mk[0] : 0, m_r0 = mk ? mk[512] : 0, m_r1 = mk ? mk[1024] : 0;
        TileRegs hr;
        if (!mk) hr = load_tile_regs(r1, p0, B);
        zero_acc(acc);
        layer_mma(Wr2t, 32, 32, H, Wc, acc, nq, ph, lane);
        if (!mk) { store_tile_regs(H, hr); __syncthreads(); }
        ChunkRegs<KC> wn = first_chunk(Wr1t, HID);    // the next product's first weight chunk rides under the epilogue and the store
        if (mk) epilogue_grad<2>(H, acc, nq, ph, lane, m_r1); else epilogue_grad<1>(H, acc, nq, ph, lane);
        __syncthreads();
        store_tile(H, gA_r1, p0, B);
        s_r1 += tile_colsum<1>(H);
        if (!mk) hr = load_tile_regs(r0, p0, B);
        zero_acc(acc);
        layer_mma(Wr1t, HID, HID, H, Wc, acc, nq, ph, lane, true, &wn);
        if (!mk) { store_tile_regs(H, hr); __syncthreads(); }
        if (mk) epilogue_grad<2>(H, acc, nq, ph, lane, m_r0); else epilogue_grad<1>(H, acc, nq, ph, lane);
        stage_small(Wc, Wr0nt);
        __syncthreads();
        store_tile(H, gA_r0, p0, B);
        s_r0 += tile_colsum<1>(H);
        // ---- normals: cotangent of the 27 encoded-normal inputs (columns 54..80 of W_R0), then the posenc chain rule
        if (wave < kRowWaves) {
            const f32x16 y = small_mma(Wc, H, wave, lane);
            float *srow = S + (size_t)(wave * 32 + (lane & 31)) * SP;
#pragma unroll
            for (int i = 0; i < 16; i++) srow[(i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)] = y[i];
        }
        __syncthreads();
        if (threadIdx.x < BM * 3) {
            const int row = threadIdx.x / 3, d = threadIdx.x - row * 3;
            const int64_t gp = p0 + row;
            if (gp < B) {
                const float *y = S + (size_t)row * SP;
                const float x = nrm[gp * 3 + d];
                float r = y[d];
#pragma unroll
                for (int k = 0; k < NF; k++) {
                    const float f = (float)(1 << k);
                    float sn, cs;
                    __sincosf(x * f, &sn, &cs);
                    r += f * (cs * y[3 + 6 * k + d] - sn * y[3 + 6 * k + 3 + d]);
                }
                d_nrm[gp * 3 + d] = r;
            }
        }
        // ---- feature vector: g_fv = gA_r0 . W_R0[:, 81:]  (no activation between the colour MLP and the rendering MLP)
        zero_acc(acc);
        layer_mma(Wr0ft, HID, HID, H, Wc, acc, nq, ph, lane);   // its first barrier also fences the scratch reads above
        wn = first_chunk(Wc1t, HID);
        epilogue_grad<0>(H, acc, nq, ph, lane);
        __syncthreads();
        store_tile(H, g_fv, p0, B);
        s_c1 += tile_colsum<1>(H);
        if (!mk) hr = load_tile_regs(hc, p0, B);
        zero_acc(acc);
        layer_mma(Wc1t, HID, HID, H, Wc, acc, nq, ph, lane, true, &wn);
        if (!mk) { store_tile_regs(H, hr); __syncthreads(); }
        if (mk) epilogue_grad<2>(H, acc, nq, ph, lane, m_hc); else epilogue_grad<1>(H, acc, nq, ph, lane);
        stage_small(Wc, Wc0t);
        __syncthreads();
        store_tile(H, gA_hc, p0, B);
        s_c0 += tile_colsum<1>(H);
        if (wave < kRowWaves) {
            const f32x16 y = small_mma(Wc, H, wave, lane);
            float *srow = S + (size_t)(wave * 32 + (lane & 31)) * SP;
#pragma unroll
            for (int i = 0; i < 16; i++) srow[(i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)] = y[i];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < BM * NFC; idx += kThreads) {   // level-major [16, B, 2] fp32: coalesced runs for the hash scatter
            const int l = idx / (BM * 2), rem = idx - l * (BM * 2), row = rem >> 1, ch = rem & 1;
            if (p0 + row < B) g_featc[((size_t)l * B + p0 + row) * 2 + ch] = S[(size_t)row * SP + l * 2 + ch];
        }
        __syncthreads();
    }
    if (gb && threadIdx.x < HID) {
        unsafeAtomicAdd(gb + threadIdx.x, s_r1);
        unsafeAtomicAdd(gb + HID + threadIdx.x, s_r0);
        unsafeAtomicAdd(gb + 2 * HID + threadIdx.x, s_c1);
        unsafeAtomicAdd(gb + 3 * HID + threadIdx.x, s_c0);
    }
    if (gb) {   // workgroup reduction before the (same-address) atomics
        float *red = reinterpret_cast<float *>(lds);
        __syncthreads();
        red[threadIdx.x] = s_y;
        __syncthreads();
        if (threadIdx.x < 3) {
            float t = 0.f;
            for (int i = threadIdx.x; i < kThreads; i += 4) t += red[i];
            unsafeAtomicAdd(gb + 4 * HID + threadIdx.x, t);
        }
    }
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

constexpr size_t kLdsFwd = ((size_t)BM * HP + 2 * (size_t)HID * WP + (size_t)BM * PP) * sizeof(uint16_t) + (4 * HID + 4) * sizeof(float);
constexpr size_t kLdsBwd = ((size_t)BM * HP + 2 * (size_t)HID * WP) * sizeof(uint16_t) + (size_t)BM * SP * sizeof(float);

}  // namespace

extern "C" {

int64_t hs_appearance_mask_words(int64_t B) { return ((B + BM - 1) / BM) * 3 * 8 * 64; }

int hs_appearance_fwd(const float *featc, const float *points, const float *dirs, const float *normals, const void *Wc0, const void *Wc1,
                      const void *Wr0f, const void *Wr0p, const void *Wr1, const void *Wr2, const float *bc0, const float *bc1, const float *br0,
                      const float *br1, const float *br2, void *xin, void *hc, void *fv, void *r0, void *r1, float *rgb, int64_t B, uint64_t *relu_masks,
                      void *stream) {
    if (B == 0) return HS_OK;
    if (!featc || !points || !dirs || !normals || !Wc0 || !Wc1 || !Wr0f || !Wr0p || !Wr1 || !Wr2 || !bc0 || !bc1 || !br0 || !br1 || !br2 || !xin ||
        !hc || !fv || !r0 || !r1 || !rgb)
        return HS_ERR_NULL;
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_appear_fwd, (int)kLdsFwd);
    const int64_t ntiles = (B + BM - 1) / BM;
    const int grid = (int)(ntiles < kGridCap ? ntiles : kGridCap);
    k_appear_fwd<<<grid, kThreads, kLdsFwd, (hipStream_t)stream>>>(
        featc, points, dirs, normals, (const uint16_t *)Wc0, (const uint16_t *)Wc1, (const uint16_t *)Wr0f, (const uint16_t *)Wr0p, (const uint16_t *)Wr1,
        (const uint16_t *)Wr2, bc0, bc1, br0, br1, br2, (uint16_t *)xin, (uint16_t *)hc, (uint16_t *)fv, (uint16_t *)r0, (uint16_t *)r1, rgb, B, relu_masks);
    return check_launch();
}

int hs_appearance_bwd(const float *g_rgb, const float *rgb, const float *normals, const void *r1, const void *r0, const void *hc, const void *Wr2t,
                      const void *Wr1t, const void *Wr0ft, const void *Wr0nt, const void *Wc1t, const void *Wc0t, void *gy, void *gA_r1, void *gA_r0,
                      void *g_fv, void *gA_hc, float *d_normals, float *g_featc, float *gbias, int64_t B, const uint64_t *relu_masks, void *stream) {
    if (B == 0) return HS_OK;
    if (!g_rgb || !rgb || !normals || (!relu_masks && (!r1 || !r0 || !hc)) || !Wr2t || !Wr1t || !Wr0ft || !Wr0nt || !Wc1t || !Wc0t || !gy || !gA_r1 || !gA_r0 || !g_fv ||
        !gA_hc || !d_normals || !g_featc)
        return HS_ERR_NULL;
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_appear_bwd, (int)kLdsBwd);
    const int64_t ntiles = (B + BM - 1) / BM;
    const int grid = (int)(ntiles < kGridCap ? ntiles : kGridCap);
    k_appear_bwd<<<grid, kThreads, kLdsBwd, (hipStream_t)stream>>>(
        g_rgb, rgb, normals, (const uint16_t *)r1, (const uint16_t *)r0, (const uint16_t *)hc, (const uint16_t *)Wr2t, (const uint16_t *)Wr1t,
        (const uint16_t *)Wr0ft, (const uint16_t *)Wr0nt, (const uint16_t *)Wc1t, (const uint16_t *)Wc0t, (uint16_t *)gy, (uint16_t *)gA_r1,
        (uint16_t *)gA_r0, (uint16_t *)g_fv, (uint16_t *)gA_hc, d_normals, g_featc, gbias, B, relu_masks);
    return check_launch();
}

}  // extern "C"
