// split_bwd.h -- the output-cotangent image of the value+Jacobian trunk backward as a device function of (workgroup, workgroups), so that it can run
// as a launch of its own (encode_ops.hip: k_trunk_split_bwd) or as a range of workgroups of another launch (trunk_rr.hip: k_rr_gy_split).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

namespace {

// cotangent of Y as the bf16 [4*B, KP] image k_trunk_bwd reads (columns >= K zero); any input may be NULL (= zero)
__device__ __forceinline__ void trunk_split_bwd_body(int block, int nblocks, const float *__restrict__ g_raw, const float *__restrict__ g_sdf, const int64_t *__restrict__ idx,
                                                          const float *__restrict__ g_grad, const float *__restrict__ g_yeik,
                                                          const float *__restrict__ g_mineik, const float *__restrict__ g_theta, int64_t B,
                                                          int64_t n_main, int K, int KP, __hip_bfloat16 *__restrict__ g) {
    // one thread = 8 consecutive columns of one row (one 16-byte store); KP is a multiple of 32
    const int groups = KP >> 3;
    const int64_t total = B * 4 * groups, Be = B - n_main;
    for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < total; i += (int64_t)nblocks * 256) {
        const int k0 = (int)(i % groups) * 8;
        const int64_t row = i / groups, b = row >> 2;
        const int r = (int)(row & 3);
        const int hit_k = (int)idx[b];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = 0.f;
        const int jh = hit_k - k0;                       // position of the minimum's column inside this 8-column segment, if any
        const bool main_pt = b < n_main;
        const int64_t e = b - n_main;
        if (r == 0) {                                    // value row: the per-object cotangents (+ the minimum's at its column)
            const float *src = main_pt ? (g_raw ? g_raw + b * K + k0 : nullptr) : (g_yeik ? g_yeik + e * K + k0 : nullptr);
            if (src != nullptr) {
                if ((K & 3) == 0 && k0 + 8 <= K) {       // two 16-byte reads (rows of K floats, K % 4 == 0: aligned)
                    const float4 a = reinterpret_cast<const float4 *>(src)[0], c4 = reinterpret_cast<const float4 *>(src)[1];
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c4.x; v[5] = c4.y; v[6] = c4.z; v[7] = c4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (k0 + j < K) v[j] = src[j];
                }
            }
            const float *gm = main_pt ? g_sdf : g_mineik;
            if (gm != nullptr && jh >= 0 && jh < 8 && hit_k < K) {
                const float add = gm[main_pt ? b : e];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = j == jh ? v[j] + add : v[j];
            }
        } else if (main_pt) {                            // tangent row of a rendered point: only the minimum's column is live
            if (g_grad != nullptr && jh >= 0 && jh < 8 && hit_k < K) {
                const float gv = g_grad[b * 3 + (r - 1)];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = j == jh ? gv : v[j];
            }
        } else if (g_theta != nullptr) {                 // tangent row of an Eikonal point: every object's gradient row (+ the minimum's)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = k0 + j;
                if (k < K) v[j] = g_theta[((int64_t)k * Be + e) * 3 + (r - 1)] + (k == hit_k ? g_theta[((int64_t)K * Be + e) * 3 + (r - 1)] : 0.f);
            }
        }
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        uint4 out;
        uint32_t *o = reinterpret_cast<uint32_t *>(&out);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const f2 p = {v[2 * j], v[2 * j + 1]};
            const bf2 q = __builtin_convertvector(p, bf2);
            o[j] = *reinterpret_cast<const uint32_t *>(&q);
        }
        *reinterpret_cast<uint4 *>(g + row * KP + k0) = out;
    }
}

}  // namespace
