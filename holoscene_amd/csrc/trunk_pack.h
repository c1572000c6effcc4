// trunk_pack.h -- fragment-order weight images of the trunk's training kernels (trunk_rr.hip), as per-slot device functions so that more
// than one launcher can use them (trunk_rr.hip: k_rr_pack, k_trunk_pack_all; appearance2.hip: k_pack_iteration -- every image of an
// iteration, both networks, in one launch).
#pragma once
#include "wave_tile.h"

namespace {

constexpr int XS = 3;                       // 32-slot tiles of the transposed input product (96 slots: 48 per lane half)
constexpr int kW0TF = HS * XS * 64 * 8;     // bf16 elements of the W0^T image   [16 k-steps][3 tiles][64 lanes] x 8
constexpr int kW2TF = 2 * NT * 64 * 8;      //                    W2^T image   [2 k-steps][8 tiles][64 lanes] x 8
constexpr float kC = 144.269504f;           // 100 log2(e)

// slot sigma (0..47) of lane half hh of the transposed input product <-> reference input column (wave_tile.h: input_column), -1 = padding
__host__ __device__ inline int slot_column(int hh, int sigma) { return sigma < 40 ? input_column(hh, sigma) : -1; }

// ---------------------------------------------------------------------------------------------------------------- packing
// W1^T, W0^T (slot order), W2^T as fragment images, W2 as an fp32 gather table [32][256]
constexpr int kRrPackSlots = HS * NT * 64 + HS * XS * 64 + 2 * NT * 64 + 32 * 256 / 4;
__device__ __forceinline__ void rr_pack_slot(int idx, const float *__restrict__ W0, int ld0, const float *__restrict__ W1, const float *__restrict__ W2, int d_out,
                                             uint16_t *__restrict__ W1Tf, uint16_t *__restrict__ W0Tf, uint16_t *__restrict__ W2Tf,
                                             float *__restrict__ W2tab) {
    constexpr int n1 = HS * NT * 64, n0 = HS * XS * 64, n2 = 2 * NT * 64, nt = 32 * 256 / 4;
    float v[8];
    uint16_t *dst;
    if (idx < n1) {
        const int s = idx / (NT * 64), mt = (idx / 64) % NT, lane = idx & 63, m = 32 * mt + (lane & 31), kh = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = W1[(size_t)(16 * s + 8 * (e >> 2) + 4 * kh + (e & 3)) * 256 + m];
        dst = W1Tf + (size_t)idx * 8;
    } else if (idx < n1 + n0) {
        const int i = idx - n1, s = i / (XS * 64), t = (i / 64) % XS, lane = i & 63, m = lane & 31, kh = lane >> 5;
        const int q = m >> 3, hh = (m >> 2) & 1, j = m & 3, col = slot_column(hh, 16 * t + 4 * q + j);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = col >= 0 ? W0[(size_t)(16 * s + 8 * (e >> 2) + 4 * kh + (e & 3)) * ld0 + col] : 0.f;
        dst = W0Tf + (size_t)i * 8;
    } else if (idx < n1 + n0 + n2) {
        const int i = idx - n1 - n0, s = i / (NT * 64), ntile = (i / 64) % NT, lane = i & 63, m = 32 * ntile + (lane & 31), kh = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int obj = 16 * s + 8 * kh + e;
            v[e] = obj < d_out ? W2[(size_t)obj * 256 + m] : 0.f;
        }
        dst = W2Tf + (size_t)i * 8;
    } else if (idx < n1 + n0 + n2 + nt) {
        const int i = idx - n1 - n0 - n2, k = (4 * i) / 256, n = (4 * i) % 256;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < d_out) o = *reinterpret_cast<const float4 *>(W2 + (size_t)k * 256 + n);
        *reinterpret_cast<float4 *>(W2tab + 4 * i) = o;
        return;
    } else {
        return;
    }
    uint4 pk;
    pk.x = pack2(v[0], v[1]); pk.y = pack2(v[2], v[3]); pk.z = pack2(v[4], v[5]); pk.w = pack2(v[6], v[7]);
    *reinterpret_cast<uint4 *>(dst) = pk;
}

// the row-major bf16 transposes W1^T [256,256], W2^T [256,32] (columns >= d_out zero), W0^T [256,256] (rows >= f_in zero) the value+Jacobian
// backward of the Eikonal points reads (sdf_mlp.hip: k_trunk_bwd)
constexpr int kTransSlots = (256 * 256 + 256 * 32 + 256 * 256) / 8;
__device__ __forceinline__ void trunk_trans_slot(int idx, const float *__restrict__ W0, int ld0, int f_in, const float *__restrict__ W1,
                                                 const float *__restrict__ W2, int d_out, uint16_t *__restrict__ w1t, uint16_t *__restrict__ w2t,
                                                 uint16_t *__restrict__ w0t) {
    if (idx >= kTransSlots) return;
    float v[8];
    uint16_t *dst;
    if (idx < 256 * 32) {                          // W1^T
        const int r = idx >> 5, c = (idx & 31) * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = W1[(size_t)(c + e) * 256 + r];
        dst = w1t + (size_t)r * 256 + c;
    } else if (idx < 256 * 32 + 256 * 4) {         // W2^T, padded to 32 columns
        const int i = idx - 256 * 32, r = i >> 2, c = (i & 3) * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = c + e < d_out ? W2[(size_t)(c + e) * 256 + r] : 0.f;
        dst = w2t + (size_t)r * 32 + c;
    } else {                                       // W0^T, padded to 256 rows
        const int i = idx - 256 * 32 - 256 * 4, r = i >> 5, c = (i & 31) * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = r < f_in ? W0[(size_t)(c + e) * ld0 + r] : 0.f;
        dst = w0t + (size_t)r * 256 + c;
    }
    uint4 pk;
    pk.x = pack2(v[0], v[1]); pk.y = pack2(v[2], v[3]); pk.z = pack2(v[4], v[5]); pk.w = pack2(v[6], v[7]);
    *reinterpret_cast<uint4 *>(dst) = pk;
}

// slot idx of "every weight image a training pass of the trunk needs": the plain-domain fragment images + bias block (sdf_mlp2.hip's packing with
// act = 1), the transposed fragment images, and (w1t != NULL) the row-major transposes
constexpr int kTrunkPackSlots = kSdfPackSlots + kRrPackSlots + kTransSlots;
__device__ __forceinline__ void trunk_pack_all_slot(int idx, const float *__restrict__ W0, int ld0, int f_in, const float *__restrict__ b0,
                                                    const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
                                                    const float *__restrict__ b2, int d_out, uint16_t *__restrict__ W0f, uint16_t *__restrict__ W1f,
                                                    uint16_t *__restrict__ W2f, float *__restrict__ bias, uint16_t *__restrict__ W1Tf,
                                                    uint16_t *__restrict__ W0Tf, uint16_t *__restrict__ W2Tf, float *__restrict__ W2tab,
                                                    uint16_t *__restrict__ w1t, uint16_t *__restrict__ w2t, uint16_t *__restrict__ w0t) {
    if (idx < kSdfPackSlots) { sdf_pack2_slot(idx, W0, ld0, b0, W1, b1, W2, b2, d_out, W0f, W1f, W2f, bias, 1.f); return; }
    idx -= kSdfPackSlots;
    if (idx < kRrPackSlots) { rr_pack_slot(idx, W0, ld0, W1, W2, d_out, W1Tf, W0Tf, W2Tf, W2tab); return; }
    idx -= kRrPackSlots;
    if (w1t != nullptr) trunk_trans_slot(idx, W0, ld0, f_in, W1, W2, d_out, w1t, w2t, w0t);
}

}  // namespace
