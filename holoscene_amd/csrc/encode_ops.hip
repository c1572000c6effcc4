// holoscene_amd/csrc/encode_ops.hip -- fused input builders for the SDF trunk and the rendering network (gfx950).
//
// The reference assembles network inputs from dozens of tiny ops: Embedder.embed loops over frequencies with one
// sin, one cos and two muls each (model/embedder.py:22-36), then torch.cat (network.py:181-185, 586-596).  With the
// value+Jacobian trunk the tangent rows need the same again for the derivative.  Per iteration that was ~250 of the
// ~850 kernel launches.  Here each network input is written by ONE kernel, and its backward by one more.
//
//  trunk input   [B,4,F], F = 3+6*nf+LC:   row 0 = [x, sin(2^k x), cos(2^k x) ..., feat]
//                                          row 1+d = d(row 0)/dx_d = [e_d, 2^k cos(2^k x_i) d_id, -2^k sin(2^k x_i) d_id ..., dydx_d * s]
//                backward: g_feat[b,:] = G[b,0,P:], g_dydx[l,b,d*C+c] = s * G[b,1+d,P+l*C+c]   (x is not differentiated)
//  render input  [B, 3*(3+6*nf)+Fv]:       [pe(points), pe(view_dirs), pe(normals), feature_vectors]
//                backward: d_normals = pe'(normals)^T G[:, 2P:3P], d_feat = G[:, 3P:]
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"
#include "split_bwd.h"

namespace {

constexpr int kThreads = 256;

template <class T> struct IO;
template <> struct IO<float> {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct IO<__hip_bfloat16> {
    static __device__ __forceinline__ float ld(const __hip_bfloat16 *p) { return __uint_as_float((uint32_t)(*reinterpret_cast<const uint16_t *>(p)) << 16); }
    static __device__ __forceinline__ void st(__hip_bfloat16 *p, float v) {
        const uint32_t u = __float_as_uint(v);
        *reinterpret_cast<uint16_t *>(p) = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
};

// one thread per (point, column) of the [4, F] block
template <class T>
__global__ __launch_bounds__(kThreads) void k_trunk_input_fwd(const float *__restrict__ x, const float *__restrict__ feat,
                                                               const float *__restrict__ dydx, T *__restrict__ out, int64_t B, int nf, int L, int C,
                                                               float jac_scale, int pitch) {
    const int P = 3 + 6 * nf, LC = L * C, F = pitch;   // columns >= P+LC (row padding for aligned MFMA operands) are zero-filled
    const int64_t total = B * F;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t b = i / F;
        const int c = (int)(i - b * F);
        T *o = out + b * 4 * F + c;
        if (c >= P + LC) {
            IO<T>::st(o, 0.f); IO<T>::st(o + F, 0.f); IO<T>::st(o + 2 * F, 0.f); IO<T>::st(o + 3 * F, 0.f);
            continue;
        }
        float v0, t[3] = {0.f, 0.f, 0.f};
        if (c < 3) {
            v0 = x[b * 3 + c];
            t[c] = 1.f;
        } else if (c < P) {
            const int k = (c - 3) / 6, r = (c - 3) - 6 * k, d = r % 3;
            const float f = (float)(1 << k), a = x[b * 3 + d] * f;
            float sn, cs;
            if (sizeof(T) == 2) __sincosf(a, &sn, &cs);   // bf16 destination: the hardware v_sin/v_cos units are exact to more bits than it keeps
            else sincosf(a, &sn, &cs);
            if (r < 3) { v0 = sn; t[d] = f * cs; }
            else { v0 = cs; t[d] = -f * sn; }
        } else {
            const int lc = c - P, l = lc / C, ch = lc - l * C;
            v0 = feat[b * LC + lc];
            const float *j = dydx + ((int64_t)l * B + b) * 3 * C + ch;
            t[0] = j[0] * jac_scale; t[1] = j[C] * jac_scale; t[2] = j[2 * C] * jac_scale;
        }
        IO<T>::st(o, v0);
        IO<T>::st(o + F, t[0]);
        IO<T>::st(o + 2 * F, t[1]);
        IO<T>::st(o + 3 * F, t[2]);
    }
}

template <class T>
__global__ __launch_bounds__(kThreads) void k_trunk_input_bwd(const T *__restrict__ G, float *__restrict__ g_feat, float *__restrict__ g_dydx,
                                                               int64_t B, int nf, int L, int C, float jac_scale, int pitch) {
    const int P = 3 + 6 * nf, LC = L * C, F = pitch;
    const int64_t total = B * LC;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t b = i / LC;
        const int lc = (int)(i - b * LC), l = lc / C, ch = lc - l * C;
        const T *g = G + b * 4 * F + P + lc;
        g_feat[b * LC + lc] = IO<T>::ld(g);
        float *j = g_dydx + ((int64_t)l * B + b) * 3 * C + ch;
        j[0] = IO<T>::ld(g + F) * jac_scale;
        j[C] = IO<T>::ld(g + 2 * F) * jac_scale;
        j[2 * C] = IO<T>::ld(g + 3 * F) * jac_scale;
    }
}

// posenc value of column c (0 <= c < P) of a 3-vector v, and optionally its derivative w.r.t. v[d]
__device__ __forceinline__ float pe_col(const float v[3], int c, int &d, float &deriv) {
    if (c < 3) { d = c; deriv = 1.f; return v[c]; }
    const int k = (c - 3) / 6, r = (c - 3) - 6 * k;
    d = r % 3;
    const float f = (float)(1 << k);
    float sn, cs;
    sincosf(v[d] * f, &sn, &cs);
    if (r < 3) { deriv = f * cs; return sn; }
    deriv = -f * sn;
    return cs;
}

template <class T>
__global__ __launch_bounds__(kThreads) void k_render_input_fwd(const float *__restrict__ pts, const float *__restrict__ dirs,
                                                                const float *__restrict__ nrm, const T *__restrict__ fv, T *__restrict__ out,
                                                                int64_t B, int nf, int Fv) {
    const int P = 3 + 6 * nf, W = 3 * P + Fv;
    const int64_t total = B * W;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t b = i / W;
        const int c = (int)(i - b * W);
        if (c >= 3 * P) { out[i] = fv[b * Fv + (c - 3 * P)]; continue; }
        const int which = c / P, cc = c - which * P;
        const float *src = (which == 0 ? pts : (which == 1 ? dirs : nrm)) + b * 3;
        const float v[3] = {src[0], src[1], src[2]};
        int d; float deriv;
        IO<T>::st(out + i, pe_col(v, cc, d, deriv));
    }
}

// d_normals[b,d] = sum_c G[b, 2P+c] * d pe_c/d n_d (one thread per point; P columns are few).  d_feat = G[:, 3P:] is copied
// only when the caller wants a packed tensor (d_fv != NULL); the Python layer hands autograd a strided view instead.
template <class T>
__global__ __launch_bounds__(kThreads) void k_render_input_bwd(const T *__restrict__ G, const float *__restrict__ nrm, float *__restrict__ d_nrm,
                                                                T *__restrict__ d_fv, int64_t B, int nf, int Fv) {
    const int P = 3 + 6 * nf, W = 3 * P + Fv;
    const int per = d_fv ? Fv + 1 : 1;
    const int64_t total = B * per;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t b = i / per;
        const int c = (int)(i - b * per);
        if (c + 1 < per) { d_fv[b * Fv + c] = G[b * W + 3 * P + c]; continue; }
        const float v[3] = {nrm[b * 3], nrm[b * 3 + 1], nrm[b * 3 + 2]};
        float acc[3] = {0.f, 0.f, 0.f};
        const T *g = G + b * W + 2 * P;
        for (int cc = 0; cc < P; cc++) {
            int d; float deriv;
            (void)pe_col(v, cc, d, deriv);
            acc[d] += IO<T>::ld(g + cc) * deriv;
        }
        d_nrm[b * 3] = acc[0]; d_nrm[b * 3 + 1] = acc[1]; d_nrm[b * 3 + 2] = acc[2];
    }
}

// x = o + z*d per (ray, sample) and its image in the hash grid's unit cube, x01 = (x/divide_factor + 1)/2 -- the five whole-tensor
// passes the sampler made per round before each SDF sweep (ray_sampler.py:151-153, network.py:176, hashgrid.py:158).
// Same operation order and roundings as those passes (no contraction; a tensor divided by a scalar is computed by ATen
// as a multiplication by the scalar's fp32 reciprocal, reproduced here).
__global__ __launch_bounds__(kThreads) void k_ray_points(const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ z,
                                                          float *__restrict__ x, float *__restrict__ x01, int64_t R, int S, float divide_factor, hsGate gate) {
    if (gate.a != nullptr && !(*gate.a > *gate.b)) return;
    const int64_t total = R * S * 3;
    const float inv_df = __fdiv_rn(1.0f, divide_factor);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t p = i / 3;
        const int c = (int)(i - p * 3);
        const int64_t r = p / S;
        const float v = __fadd_rn(o[r * 3 + c], __fmul_rn(z[p], d[r * 3 + c]));
        x[i] = v;
        x01[i] = __fmul_rn(__fadd_rn(__fmul_rn(v, inv_df), 1.0f), 0.5f);
    }
}

// ---- consumers of the trunk's value+Jacobian output Y [4*B, K] (rows per point: value, d/dx, d/dy, d/dz), one wave per point.
// Rendered points (b < n_main) need the per-object SDFs, their minimum with its index, and the gradient OF THE MINIMUM only
// (network.py:289-299); Eikonal points (b >= n_main) need everything (network.py:856-866).  Replaces the [B,K,3] Jacobian
// materialisation + min + gather (forward) and the zero-fill / scatter / slice-pad kernels autograd runs for them (backward).
__global__ __launch_bounds__(256) void k_trunk_split_fwd(const float *__restrict__ Y, int64_t B, int64_t n_main, int K, float *__restrict__ sdf_raw,
                                                          float *__restrict__ sdf, int64_t *__restrict__ idx, float *__restrict__ grad,
                                                          float *__restrict__ y_eik, float *__restrict__ min_eik, float *__restrict__ gtheta) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int64_t Be = B - n_main;
    for (int64_t b = wave0; b < B; b += nwaves) {
        const float *y = Y + b * 4 * K;
        const float v = lane < K ? y[lane] : INFINITY;
        float best = v;
        int bi = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {   // minimum, lowest index among equals
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) idx[b] = bi;
        if (b < n_main) {
            if (lane < K) sdf_raw[b * K + lane] = v;
            if (lane == 0) sdf[b] = best;
            if (lane < 3) grad[b * 3 + lane] = y[(1 + lane) * K + bi];
        } else {
            // Eikonal point e: per-object SDFs, their minimum, and the stacked gradient rows the reference builds with K+1
            // autograd passes (network.py:212-254): row k*Be + e = d sdf_k/dx, row K*Be + e = d min_k sdf_k/dx
            const int64_t e = b - n_main;
            if (lane == 0) min_eik[e] = best;
            if (lane < K) {
                y_eik[e * K + lane] = v;
#pragma unroll
                for (int d = 0; d < 3; d++) gtheta[((int64_t)lane * Be + e) * 3 + d] = y[(1 + d) * K + lane];
            }
            if (lane < 3) gtheta[((int64_t)K * Be + e) * 3 + lane] = y[(1 + lane) * K + bi];
        }
    }
}

// cotangent of Y as the bf16 [4*B, KP] image k_trunk_bwd reads (split_bwd.h)
__global__ __launch_bounds__(256) void k_trunk_split_bwd(const float *__restrict__ g_raw, const float *__restrict__ g_sdf, const int64_t *__restrict__ idx,
                                                          const float *__restrict__ g_grad, const float *__restrict__ g_yeik,
                                                          const float *__restrict__ g_mineik, const float *__restrict__ g_theta, int64_t B,
                                                          int64_t n_main, int K, int KP, __hip_bfloat16 *__restrict__ g) {
    trunk_split_bwd_body((int)blockIdx.x, (int)gridDim.x, g_raw, g_sdf, idx, g_grad, g_yeik, g_mineik, g_theta, B, n_main, K, KP, g);
}

// Every position the iteration's render pass evaluates, in one launch (network.py:805-811, 843-854): the R*N rendered samples
// o + z d, then the Eikonal set = [uniform points | near-surface points o + z_eik d] followed by the same 2R points jittered by
// (u - 0.5) * 0.01; plus each position's hash-grid coordinate (x/divide_factor + 1)/2 and the per-sample view directions.
// Same roundings as the whole-tensor ops it replaces (~20 launches): no contraction, scalar division as multiplication by the
// fp32 reciprocal.
__global__ __launch_bounds__(kThreads) void k_render_points(const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ z,
                                                             const float *__restrict__ z_eik, const float *__restrict__ eik_uniform,
                                                             const float *__restrict__ eik_jitter, int64_t R, int N, float divide_factor,
                                                             float *__restrict__ x, float *__restrict__ x01, float *__restrict__ dirs, float eik_scale,
                                                             float eik_shift) {
    const int64_t n_main = R * N, n_eik = z_eik ? 4 * R : 0;
    const int64_t total = (n_main + n_eik) * 3;
    const float inv_df = __fdiv_rn(1.0f, divide_factor);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t p = i / 3;
        const int c = (int)(i - p * 3);
        float v;
        if (p < n_main) {
            const int64_t r = p / N;
            v = __fadd_rn(o[r * 3 + c], __fmul_rn(z[p], d[r * 3 + c]));
            dirs[i] = d[r * 3 + c];
        } else {
            int64_t e = p - n_main;                 // 0..4R: [uniform R | near R | jittered copies 2R]
            const bool jit = e >= 2 * R;
            if (jit) e -= 2 * R;
            v = e < R ? __fadd_rn(__fmul_rn(eik_uniform[e * 3 + c], eik_scale), eik_shift) : __fadd_rn(o[(e - R) * 3 + c], __fmul_rn(z_eik[e - R], d[(e - R) * 3 + c]));
            if (jit) v = __fadd_rn(v, __fmul_rn(__fsub_rn(eik_jitter[e * 3 + c], 0.5f), 0.01f));
        }
        x[i] = v;
        x01[i] = __fmul_rn(__fadd_rn(__fmul_rn(v, inv_df), 1.0f), 0.5f);
    }
}

// ---- batch assembly from device-resident images: dst[i, :] = src[idx[i], :] for several arrays in one launch (the pixel gather of
// a training batch -- uv, colour, depth, normal, label rows of the sampled pixels, datasets/scene_dataset.py:143-167 does it with
// fancy indexing on the host -- written straight into the captured graph's static input block).  Rows are copied in 4-byte words.
struct GatherJobs { hsGatherJob j[HS_GATHER_MAX_JOBS]; };

__global__ __launch_bounds__(256) void k_gather_rows(GatherJobs jobs) {
    const hsGatherJob jb = jobs.j[blockIdx.y];
    const int words = jb.row_bytes >> 2;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(jb.src);
    uint32_t *dst = reinterpret_cast<uint32_t *>(jb.dst);
    const int64_t total = jb.n * words;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / words;
        const int w = (int)(i - r * words);
        dst[i] = src[jb.idx[r] * words + w];
    }
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

int grid_for(int64_t total) {
    const int64_t want = (total + kThreads - 1) / kThreads;
    return (int)(want < 1 ? 1 : (want < 256 * 16 ? want : 256 * 16));
}

}  // namespace

extern "C" {

int hs_trunk_input_fwd(const float *x, const float *feat, const float *dydx, void *out, int64_t B, int32_t nfreq, int32_t L, int32_t C,
                       float jac_scale, int32_t pitch, int32_t dtype, void *stream) {
    if (nfreq < 0 || nfreq > 16 || L < 1 || C < 1 || pitch < 3 + 6 * nfreq + L * C || (dtype != HS_F32 && dtype != HS_BF16)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!x || !feat || !dydx || !out) return HS_ERR_NULL;
    const int64_t total = B * pitch;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HS_F32) k_trunk_input_fwd<float><<<grid_for(total), kThreads, 0, st>>>(x, feat, dydx, (float *)out, B, nfreq, L, C, jac_scale, pitch);
    else k_trunk_input_fwd<__hip_bfloat16><<<grid_for(total), kThreads, 0, st>>>(x, feat, dydx, (__hip_bfloat16 *)out, B, nfreq, L, C, jac_scale, pitch);
    return check_launch();
}

int hs_trunk_input_bwd(const void *G, float *g_feat, float *g_dydx, int64_t B, int32_t nfreq, int32_t L, int32_t C, float jac_scale, int32_t pitch,
                       int32_t dtype, void *stream) {
    if (nfreq < 0 || nfreq > 16 || L < 1 || C < 1 || pitch < 3 + 6 * nfreq + L * C || (dtype != HS_F32 && dtype != HS_BF16)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!G || !g_feat || !g_dydx) return HS_ERR_NULL;
    const int64_t total = B * L * C;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HS_F32) k_trunk_input_bwd<float><<<grid_for(total), kThreads, 0, st>>>((const float *)G, g_feat, g_dydx, B, nfreq, L, C, jac_scale, pitch);
    else k_trunk_input_bwd<__hip_bfloat16><<<grid_for(total), kThreads, 0, st>>>((const __hip_bfloat16 *)G, g_feat, g_dydx, B, nfreq, L, C, jac_scale, pitch);
    return check_launch();
}

int hs_render_input_fwd(const float *points, const float *view_dirs, const float *normals, const void *feature_vectors, void *out, int64_t B,
                        int32_t nfreq, int32_t Fv, int32_t dtype, void *stream) {
    if (nfreq < 0 || nfreq > 16 || Fv < 0 || (dtype != HS_F32 && dtype != HS_BF16)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!points || !view_dirs || !normals || !feature_vectors || !out) return HS_ERR_NULL;
    const int64_t total = B * (3 * (3 + 6 * nfreq) + Fv);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HS_F32) k_render_input_fwd<float><<<grid_for(total), kThreads, 0, st>>>(points, view_dirs, normals, (const float *)feature_vectors, (float *)out, B, nfreq, Fv);
    else k_render_input_fwd<__hip_bfloat16><<<grid_for(total), kThreads, 0, st>>>(points, view_dirs, normals, (const __hip_bfloat16 *)feature_vectors, (__hip_bfloat16 *)out, B, nfreq, Fv);
    return check_launch();
}

int hs_render_input_bwd(const void *G, const float *normals, float *d_normals, void *d_feature_vectors, int64_t B, int32_t nfreq, int32_t Fv,
                        int32_t dtype, void *stream) {
    if (nfreq < 0 || nfreq > 16 || Fv < 0 || (dtype != HS_F32 && dtype != HS_BF16)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!G || !normals || !d_normals) return HS_ERR_NULL;
    const int64_t total = B * (d_feature_vectors ? Fv + 1 : 1);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HS_F32) k_render_input_bwd<float><<<grid_for(total), kThreads, 0, st>>>((const float *)G, normals, d_normals, (float *)d_feature_vectors, B, nfreq, Fv);
    else k_render_input_bwd<__hip_bfloat16><<<grid_for(total), kThreads, 0, st>>>((const __hip_bfloat16 *)G, normals, d_normals, (__hip_bfloat16 *)d_feature_vectors, B, nfreq, Fv);
    return check_launch();
}

int hs_ray_points(const float *cam_loc, const float *ray_dirs, const float *z, float *x, float *x01, int64_t R, int32_t S, float divide_factor,
                  const hsGate *gate, void *stream) {
    if (S < 0 || divide_factor == 0.f) return HS_ERR_ARG;
    if (R == 0 || S == 0) return HS_OK;
    if (!cam_loc || !ray_dirs || !z || !x || !x01) return HS_ERR_NULL;
    k_ray_points<<<grid_for(R * S * 3), kThreads, 0, (hipStream_t)stream>>>(cam_loc, ray_dirs, z, x, x01, R, S, divide_factor, gate ? *gate : hsGate{nullptr, nullptr});
    return check_launch();
}

int hs_trunk_split_fwd(const float *Y, int64_t B, int64_t n_main, int32_t K, float *sdf_raw, float *sdf, int64_t *idx, float *grad, float *y_eik,
                       float *min_eik, float *grad_theta, void *stream) {
    if (K < 1 || K > 64 || n_main < 0 || n_main > B) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!Y || !idx || (n_main > 0 && (!sdf_raw || !sdf || !grad)) || (n_main < B && (!y_eik || !min_eik || !grad_theta))) return HS_ERR_NULL;
    const int64_t want = (B + 3) / 4;
    k_trunk_split_fwd<<<(int)(want < 8192 ? want : 8192), 256, 0, (hipStream_t)stream>>>(Y, B, n_main, K, sdf_raw, sdf, idx, grad, y_eik, min_eik, grad_theta);
    return check_launch();
}

int hs_trunk_split_bwd(const float *g_sdf_raw, const float *g_sdf, const int64_t *idx, const float *g_grad, const float *g_y_eik,
                       const float *g_min_eik, const float *g_grad_theta, int64_t B, int64_t n_main, int32_t K, int32_t KP, void *g, void *stream) {
    if (K < 1 || K > KP || n_main < 0 || n_main > B) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!g || !idx) return HS_ERR_NULL;
    if (KP & 31) return HS_ERR_ARG;
    k_trunk_split_bwd<<<grid_for(B * 4 * (KP / 8)), 256, 0, (hipStream_t)stream>>>(g_sdf_raw, g_sdf, idx, g_grad, g_y_eik, g_min_eik, g_grad_theta, B, n_main, K, KP,
                                                                             (__hip_bfloat16 *)g);
    return check_launch();
}

int hs_render_points(const float *cam_loc, const float *ray_dirs, const float *z_vals, const float *z_eik, const float *eik_uniform,
                     const float *eik_jitter, int64_t R, int32_t N, float divide_factor, float *x, float *x01, float *dirs_flat, float eik_scale,
                     float eik_shift, void *stream) {
    if (N < 1 || divide_factor == 0.f) return HS_ERR_ARG;
    if (R == 0) return HS_OK;
    if (!cam_loc || !ray_dirs || !z_vals || !x || !x01 || !dirs_flat || (z_eik && (!eik_uniform || !eik_jitter))) return HS_ERR_NULL;
    k_render_points<<<grid_for((R * N + (z_eik ? 4 * R : 0)) * 3), kThreads, 0, (hipStream_t)stream>>>(cam_loc, ray_dirs, z_vals, z_eik, eik_uniform,
                                                                                                      eik_jitter, R, N, divide_factor, x, x01, dirs_flat, eik_scale, eik_shift);
    return check_launch();
}

int hs_gather_rows(const hsGatherJob *jobs, int32_t n_jobs, void *stream) {
    if (n_jobs < 0 || n_jobs > HS_GATHER_MAX_JOBS) return HS_ERR_ARG;
    if (n_jobs == 0) return HS_OK;
    if (!jobs) return HS_ERR_NULL;
    GatherJobs gj;
    int64_t most = 0;
    for (int i = 0; i < n_jobs; i++) {
        const hsGatherJob &j = jobs[i];
        if (j.n < 0 || j.row_bytes <= 0 || (j.row_bytes & 3)) return HS_ERR_ARG;
        if (j.n > 0 && (!j.src || !j.dst || !j.idx)) return HS_ERR_NULL;
        gj.j[i] = j;
        const int64_t t = j.n * (j.row_bytes >> 2);
        most = t > most ? t : most;
    }
    if (most == 0) return HS_OK;
    const int64_t want = (most + 255) / 256;
    k_gather_rows<<<dim3((unsigned)(want < 64 ? want : 64), n_jobs), 256, 0, (hipStream_t)stream>>>(gj);
    return check_launch();
}

}  // extern "C"
