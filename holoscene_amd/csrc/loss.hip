// holoscene_amd/csrc/loss.hip -- fused Stage-1 objective, value and gradient in one pass (gfx950).
//
// The reference evaluates MonoSDFLoss + HoloSceneLoss (model/loss.py:211-346, 487-492, 611-666) as ~80 small tensor ops
// and autograd replays ~120 more for the backward -- on tensors of a few thousand elements, i.e. pure launch overhead
// (1 ms of a 7 ms iteration).  All terms below are closed-form functions of the per-ray outputs, so this file computes
// each term together with its analytic gradient:
//
//   k_loss_rays   (one workgroup; rays strided over the lanes, block reductions in LDS)
//       rgb       mean |rgb - gt|                                                     loss.py:211-214
//       depth     scale/shift-invariant: (w,q) = argmin sum (w p + q - t)^2, mean(min((w p + q - t)^2, 1))   :181-193, 246-277
//                 -- the gradient flows through the least-squares solution (w, q) as autograd's does
//       normal    l1 and (1 - cos) between normalised predicted and prior normals, foreground rays only      :279-288, 313-320
//       opacity   binary cross entropy of the per-object opacities against the one-hot instance label        :487-492
//   k_loss_eikonal (many workgroups over the stacked gradient rows)
//       eikonal   mean (|g| - 1)^2 over the first half of the rows                                         :232-234
//       smooth    mean | g1/(|g1|+1e-5) - g2/(|g2|+1e-5) |, row i of the first half against row i of the second   :236-244
//
// Gradients are written already multiplied by the loss weights, so the autograd Function only scales them by the
// incoming cotangent.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"
#include "wave_ops.h"

namespace {

constexpr int kBlock = 1024;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// sum over the whole block; every thread gets the result.  scratch: >= kBlock/64 floats
__device__ float block_sum(float v, float *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += scratch[w];
    return t;
}

// N sums over the whole block at once; every thread gets all of them.  One rendezvous for the lot: k_loss_rays forms eleven block sums, and as
// eleven calls of block_sum they were 22 barriers over 16 waves and 66 LDS-crossbar shuffles -- most of that 13 us single-workgroup launch.
// scratch: >= N * kBlock / 64 floats.  (wave sums on DPP operands, wave partials added in wave order: deterministic)
template <int N>
__device__ __forceinline__ void block_sum_n(float (&v)[N], float *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = hs_wave::sum(v[i]);
    __syncthreads();          // (the previous call's readers are done with scratch)
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; i++) scratch[wave * N + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = 0.f;
    for (int w = 0; w < nw; w++) {
#pragma unroll
        for (int i = 0; i < N; i++) v[i] += scratch[w * N + i];
    }
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }

// Ray-local heavy parts, one wave per ray: the foreground flag (does the SDF change sign along the ray, loss.py:313-315: N
// values) and the opacity BCE with its gradient (K values).  The single-workgroup kernel below used to walk both per thread
// (1 024 threads striding 392-byte rows on ONE CU: 100 us); here they are coalesced wave reads spread over the chip.
__device__ __forceinline__ void ray_local_block(int block, const float *__restrict__ sdf, const float *__restrict__ opac, const int64_t *__restrict__ segs,
                                                const float *__restrict__ gt_mask, int R, int N, int K, float w_opac, float *__restrict__ fg,
                                                float *__restrict__ bce, float *__restrict__ g_opac) {
    const int lane = threadIdx.x & 63;
    const int r = block * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    bool pos = false, neg = false;
    for (int i = lane; i < N; i += 64) {
        const float s = sdf[(size_t)r * N + i];
        pos |= s > 0.f;
        neg |= s < 0.f;
    }
    const bool any_pos = __ballot(pos) != 0ull, any_neg = __ballot(neg) != 0ull;
    const int lab = (int)segs[r];
    const float scale = w_opac / (float)R / (float)K;
    float s_op = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float o = opac[(size_t)r * K + k];
        const float p = fminf(fmaxf(o, 1e-4f), 1.f - 1e-4f);
        const bool inside = o >= 1e-4f && o <= 1.f - 1e-4f;
        if (k == lab) { s_op += -fmaxf(logf(p), -100.f); g_opac[(size_t)r * K + k] = inside ? -scale / p : 0.f; }
        else { s_op += -fmaxf(logf(1.f - p), -100.f); g_opac[(size_t)r * K + k] = inside ? scale / (1.f - p) : 0.f; }
    }
    s_op = wave_sum(s_op);
    if (lane == 0) {
        fg[r] = (any_pos && any_neg && gt_mask[r] > 0.5f) ? 1.f : 0.f;
        bce[r] = s_op;
    }
}

__global__ __launch_bounds__(256) void k_loss_ray_local(const float *__restrict__ sdf, const float *__restrict__ opac, const int64_t *__restrict__ segs,
                                                         const float *__restrict__ gt_mask, int R, int N, int K, float w_opac,
                                                         float *__restrict__ fg, float *__restrict__ bce, float *__restrict__ g_opac) {
    ray_local_block(blockIdx.x, sdf, opac, segs, gt_mask, R, N, K, w_opac, fg, bce, g_opac);
}

// out[0..4] = rgb, depth, normal_l1, normal_cos, opacity losses (unweighted).  acc2 != NULL (hs_loss_stage1: k_loss_eikonal has
// run): also out[5..6] = eikonal, smooth (= acc2 / H) and out[7] = the weighted total of all seven terms.
__global__ __launch_bounds__(kBlock) void k_loss_rays(const float *__restrict__ rgb, const float *__restrict__ rgb_gt, const float *__restrict__ depth,
                                                      const float *__restrict__ depth_gt, const float *__restrict__ nmap,
                                                      const float *__restrict__ n_gt, const float *__restrict__ fg,
                                                      const float *__restrict__ bce, int R, int K, float w_rgb, float w_depth,
                                                      float w_l1, float w_cos, float *__restrict__ out, float *__restrict__ g_rgb,
                                                      float *__restrict__ g_depth, float *__restrict__ g_nmap, const float *__restrict__ acc2, int nparts, float invH,
                                                      float w_opac, float w_eik, float w_smooth) {
    __shared__ float scratch[8 * kBlock / 64];
    const float invR = 1.f / (float)R;
    // ---- pass 1: rgb, normals, opacity (ray-local), and the five sums of the depth least-squares system
    float s_rgb = 0.f, s_l1 = 0.f, s_cos = 0.f, s_op = 0.f, sA = 0.f, sB = 0.f, sD = 0.f, sE = 0.f;
    for (int r = threadIdx.x; r < R; r += kBlock) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d = rgb[3 * r + c] - rgb_gt[3 * r + c];
            s_rgb += fabsf(d);
            g_rgb[3 * r + c] = w_rgb * sgn(d) * invR * (1.f / 3.f);
        }
        const float m = fg[r];   // foreground flag (k_loss_ray_local)
        float v[3], t[3];
        float vn = 0.f, tn = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            v[c] = nmap[3 * r + c] * m;
            t[c] = n_gt[3 * r + c];
            vn += v[c] * v[c];
            tn += t[c] * t[c];
        }
        vn = sqrtf(vn);
        tn = fmaxf(sqrtf(tn), 1e-12f);
        const float den = fmaxf(vn, 1e-12f);  // F.normalize(eps=1e-12)
        float np[3], gn[3], dot = 0.f, ndg = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            np[c] = v[c] / den;
            t[c] /= tn;
            const float d = np[c] - t[c];
            s_l1 += fabsf(d);
            dot += np[c] * t[c];
            gn[c] = w_l1 * sgn(d) * invR - w_cos * t[c] * invR;   // dL/d n_pred
        }
        s_cos += 1.f - dot;
#pragma unroll
        for (int c = 0; c < 3; c++) ndg += np[c] * gn[c];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float gv = (vn >= 1e-12f) ? (gn[c] - np[c] * ndg) / den : gn[c] / den;
            g_nmap[3 * r + c] = gv * m;
        }
        s_op += bce[r];
        const float p = depth[r], tt = depth_gt[r];
        sA += p * p; sB += p; sD += p * tt; sE += tt;
    }
    float s8[8] = {s_rgb, s_l1, s_cos, s_op, sA, sB, sD, sE};
    block_sum_n<8>(s8, scratch);
    const float Lrgb = s8[0] * invR * (1.f / 3.f);
    const float Ll1 = s8[1] * invR;
    const float Lcos = s8[2] * invR;
    const float Lop = s8[3] * invR / (float)K;
    const float A = s8[4], Bs = s8[5], D = s8[6], E = s8[7];
    const float Cn = (float)R;
    const float det = A * Cn - Bs * Bs;
    const float w = (Cn * D - Bs * E) / det, q = (A * E - Bs * D) / det;
    // ---- pass 2: clipped residuals and the two sums the (w, q) derivative needs
    float s_d = 0.f, s0 = 0.f, s1 = 0.f;
    for (int r = threadIdx.x; r < R; r += kBlock) {
        const float p = depth[r], res = w * p + q - depth_gt[r];
        const float sq = res * res;
        s_d += fminf(sq, 1.f);
        if (sq <= 1.f) { s0 += res; s1 += res * p; }
    }
    float s3[3] = {s_d, s0, s1};
    block_sum_n<3>(s3, scratch);
    const float Ld = s3[0] * invR;
    const float S0 = s3[1], S1 = s3[2];
    for (int r = threadIdx.x; r < R; r += kBlock) {
        const float p = depth[r], tt = depth_gt[r], res = w * p + q - tt;
        const float c = (res * res <= 1.f) ? 1.f : 0.f;
        const float ddet = 2.f * p * Cn - 2.f * Bs;
        const float dw = ((Cn * tt - E) - w * ddet) / det;
        const float dq = ((2.f * p * E - D - Bs * tt) - q * ddet) / det;
        g_depth[r] = w_depth * 2.f * invR * (c * res * w + S1 * dw + S0 * dq);
    }
    float se_ = 0.f, ss_ = 0.f;      // the Eikonal blocks' partial sums (acc2 [nparts][2]), added up by wave 0 in a fixed order
    if (acc2 && threadIdx.x < 64) {
        for (int i = threadIdx.x; i < nparts; i += 64) { se_ += acc2[2 * i]; ss_ += acc2[2 * i + 1]; }
        se_ = wave_sum(se_);
        ss_ = wave_sum(ss_);
    }
    if (threadIdx.x == 0) {
        out[0] = Lrgb; out[1] = Ld; out[2] = Ll1; out[3] = Lcos; out[4] = Lop;
        if (acc2) {
            const float Le = se_ * invH, Ls = ss_ * invH;
            out[5] = Le; out[6] = Ls;
            out[7] = w_rgb * Lrgb + w_depth * Ld + w_l1 * Ll1 + w_cos * Lcos + w_opac * Lop + w_eik * Le + w_smooth * Ls;
        }
    }
}

// n = g/(|g|+eps): returns n, and `back` maps a cotangent of n to a cotangent of g
struct Unit { float n[3]; float r, inv; };
__device__ __forceinline__ Unit unit(const float g[3], float eps) {
    Unit u;
    u.r = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    u.inv = 1.f / (u.r + eps);
#pragma unroll
    for (int c = 0; c < 3; c++) u.n[c] = g[c] * u.inv;
    return u;
}

// acc[0] += sum (|g1|-1)^2, acc[1] += sum |n1-n2|  (the host divides by H); gradients carry 1/H and the weights
// part != NULL: this block's two sums go to part[2 block], part[2 block + 1] (summed by k_loss_rays: deterministic) instead of into acc by atomics
__device__ __forceinline__ void eikonal_block(int block, int nblocks, const float *__restrict__ g1, const float *__restrict__ g2, int64_t H, float w_eik,
                                              float w_smooth, float *__restrict__ acc, float *__restrict__ part, float *__restrict__ d_g1,
                                              float *__restrict__ d_g2) {
    __shared__ float scratch[4];
    const float invH = 1.f / (float)H;
    float s_e = 0.f, s_s = 0.f;
    for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < H; i += (int64_t)nblocks * 256) {
        const float a[3] = {g1[3 * i], g1[3 * i + 1], g1[3 * i + 2]};
        const float b[3] = {g2[3 * i], g2[3 * i + 1], g2[3 * i + 2]};
        const Unit ua = unit(a, 1e-5f), ub = unit(b, 1e-5f);
        const float e = ua.r - 1.f;
        s_e += e * e;
        float d[3], s = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) { d[c] = ua.n[c] - ub.n[c]; s += d[c] * d[c]; }
        s = sqrtf(s);
        s_s += s;
        const float is = s > 0.f ? w_smooth * invH / s : 0.f;   // d|d|/dd = d/|d| (0 at d = 0, as torch.norm's backward)
        float da = 0.f, db = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) { da += a[c] * d[c] * is; db -= b[c] * d[c] * is; }
        const float ke = ua.r > 0.f ? w_eik * 2.f * invH * e / ua.r : 0.f;
        const float ca = ua.r > 0.f ? da * ua.inv * ua.inv / ua.r : 0.f, cb = ub.r > 0.f ? db * ub.inv * ub.inv / ub.r : 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            d_g1[3 * i + c] = ke * a[c] + d[c] * is * ua.inv - a[c] * ca;
            d_g2[3 * i + c] = -d[c] * is * ub.inv - b[c] * cb;
        }
    }
    s_e = wave_sum(s_e);
    s_s = wave_sum(s_s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { scratch[wave] = s_e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = scratch[0] + scratch[1] + scratch[2] + scratch[3];
        if (part) part[2 * block] = v;
        else unsafeAtomicAdd(acc, v);
    }
    __syncthreads();
    if (lane == 0) { scratch[wave] = s_s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = scratch[0] + scratch[1] + scratch[2] + scratch[3];
        if (part) part[2 * block + 1] = v;
        else unsafeAtomicAdd(acc + 1, v);
    }
}

__global__ __launch_bounds__(256) void k_loss_eikonal(const float *__restrict__ g1, const float *__restrict__ g2, int64_t H, float w_eik, float w_smooth,
                                                       float *__restrict__ acc, float *__restrict__ d_g1, float *__restrict__ d_g2) {
    eikonal_block(blockIdx.x, gridDim.x, g1, g2, H, w_eik, w_smooth, acc, nullptr, d_g1, d_g2);
}

// hs_loss_stage1's first launch: the ray-local blocks and the Eikonal blocks are independent -- one launch for both (a launch costs ~5 us
// in the replayed iteration whatever its size)
__global__ __launch_bounds__(256) void k_loss_local(const float *__restrict__ sdf, const float *__restrict__ opac, const int64_t *__restrict__ segs,
                                                     const float *__restrict__ gt_mask, int R, int N, int K, float w_opac, float *__restrict__ fg,
                                                     float *__restrict__ bce, float *__restrict__ g_opac, int nb_local, const float *__restrict__ g1,
                                                     const float *__restrict__ g2, int64_t H, float w_eik, float w_smooth, float *__restrict__ part,
                                                     float *__restrict__ d_g1, float *__restrict__ d_g2) {
    if ((int)blockIdx.x < nb_local) ray_local_block(blockIdx.x, sdf, opac, segs, gt_mask, R, N, K, w_opac, fg, bce, g_opac);
    else eikonal_block((int)blockIdx.x - nb_local, (int)gridDim.x - nb_local, g1, g2, H, w_eik, w_smooth, nullptr, part, d_g1, d_g2);
}

// Background-surface smoothness (model/loss.py:519-547 compute_grad_error on the depth and on the three normal channels of the
// side x side background patch, :549-557): 4 scales (stride 1, 2, 4, 8), masked absolute first differences along x and y,
// each scale divided by its number of masked pixels.  The reference evaluates this with ~190 small tensor ops and autograd
// replays ~220 more; the patch has 1 024 pixels, so one workgroup computes the value and the analytic gradient: thread = pixel,
// every pixel sums the pairs it belongs to (left / right / up / down neighbour at each scale), no atomics.
//   labels != 0 marks the pixels where something occludes the background (:652-654).
__global__ __launch_bounds__(kBlock) void k_bg_smooth(const float *__restrict__ depth, const float *__restrict__ normal, const int64_t *__restrict__ labels,
                                                      int side, float *__restrict__ out, float *__restrict__ g_depth, float *__restrict__ g_normal) {
    __shared__ float scratch[kBlock / 64];
    __shared__ float val[4][kBlock];   // masked values v = m * x per channel (depth, nx, ny, nz)
    __shared__ float msk[kBlock];
    const int P = side * side, p = threadIdx.x;
    const bool live = p < P;
    const int y = live ? p / side : 0, x = live ? p - y * side : 0;
    const float m = live && labels[p] != 0 ? 1.f : 0.f;
    msk[p] = m;
    val[0][p] = live ? m * depth[p] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) val[c + 1][p] = live ? m * normal[3 * p + c] : 0.f;
    __syncthreads();
    float total_d = 0.f, total_n = 0.f, g[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < 4; s++) {
        const int step = 1 << s;
        const bool on = live && (y % step == 0) && (x % step == 0);
        const float div = block_sum(on ? m : 0.f, scratch);
        float sum_d = 0.f, sum_n = 0.f;
        if (on) {
            const float inv = div > 0.f ? 1.f / fmaxf(div, 1.f) : 0.f;
            // neighbours at this scale: (dx, dy) in {(+step,0), (0,+step)} own the pair's value; all four feed this pixel's gradient
            const int nx[4] = {x + step, x, x - step, x}, ny[4] = {y, y + step, y, y - step};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (nx[k] < 0 || ny[k] < 0 || nx[k] >= side || ny[k] >= side) continue;
                const int q = ny[k] * side + nx[k];
                const float pw = m * msk[q];
                if (pw == 0.f) continue;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float d = k < 2 ? val[c][q] - val[c][p] : val[c][p] - val[c][q];   // later element minus earlier element
                    if (k < 2) { if (c == 0) sum_d += pw * fabsf(d); else sum_n += pw * fabsf(d); }
                    const float sg = sgn(d) * pw * m * inv;       // d v_p / d x_p = m
                    g[c] += k < 2 ? -sg : sg;
                }
            }
        }
        const float Sd = block_sum(sum_d, scratch), Sn = block_sum(sum_n, scratch);
        if (div > 0.f) { total_d += Sd / fmaxf(div, 1.f); total_n += Sn / fmaxf(div, 1.f); }
    }
    if (live) {
        g_depth[p] = g[0];
#pragma unroll
        for (int c = 0; c < 3; c++) g_normal[3 * p + c] = g[c + 1];
    }
    if (p == 0) out[0] = total_d + total_n;
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

}  // namespace

extern "C" {

int hs_loss_rays(const float *rgb, const float *rgb_gt, const float *depth, const float *depth_gt, const float *normal_map, const float *normal_gt,
                 const float *gt_mask, const float *sdf, const float *opacity, const int64_t *segs, int32_t R, int32_t N, int32_t K, float w_rgb,
                 float w_depth, float w_l1, float w_cos, float w_opac, float *out5, float *g_rgb, float *g_depth, float *g_normal_map,
                 float *g_opacity, float *scratch, void *stream) {
    if (R < 1 || N < 1 || K < 1) return HS_ERR_ARG;
    if (!rgb || !rgb_gt || !depth || !depth_gt || !normal_map || !normal_gt || !gt_mask || !sdf || !opacity || !segs || !out5 || !g_rgb || !g_depth ||
        !g_normal_map || !g_opacity || !scratch)
        return HS_ERR_NULL;
    k_loss_ray_local<<<(R + 3) / 4, 256, 0, (hipStream_t)stream>>>(sdf, opacity, segs, gt_mask, R, N, K, w_opac, scratch, scratch + R, g_opacity);
    k_loss_rays<<<1, kBlock, 0, (hipStream_t)stream>>>(rgb, rgb_gt, depth, depth_gt, normal_map, normal_gt, scratch, scratch + R, R, K, w_rgb, w_depth, w_l1,
                                                        w_cos, out5, g_rgb, g_depth, g_normal_map, nullptr, 0, 0.f, 0.f, 0.f, 0.f);
    return check_launch();
}

int hs_loss_stage1(const float *rgb, const float *rgb_gt, const float *depth, const float *depth_gt, const float *normal_map, const float *normal_gt,
                   const float *gt_mask, const float *sdf, const float *opacity, const int64_t *segs, int32_t R, int32_t N, int32_t K, const float *g1,
                   const float *g2, int64_t H, const float *weights7, float *out8, float *g_rgb, float *g_depth, float *g_normal_map, float *g_opacity,
                   float *d_g1, float *d_g2, float *scratch, void *stream) {
    if (R < 1 || N < 1 || K < 1 || H < 1) return HS_ERR_ARG;
    if (!rgb || !rgb_gt || !depth || !depth_gt || !normal_map || !normal_gt || !gt_mask || !sdf || !opacity || !segs || !g1 || !g2 || !weights7 || !out8 ||
        !g_rgb || !g_depth || !g_normal_map || !g_opacity || !d_g1 || !d_g2 || !scratch)
        return HS_ERR_NULL;
    const float *w = weights7;   // rgb, depth, normal_l1, normal_cos, opacity, eikonal, smooth
    float *part = scratch + 2 * (size_t)R;      // [HS_LOSS_EIK_BLOCKS][2]: per-block sums of the Eikonal / smoothness terms
    const int64_t want = (H + 255) / 256;
    const int nb_local = (R + 3) / 4, nb_eik = (int)(want < HS_LOSS_EIK_BLOCKS ? want : HS_LOSS_EIK_BLOCKS);
    k_loss_local<<<nb_local + nb_eik, 256, 0, (hipStream_t)stream>>>(sdf, opacity, segs, gt_mask, R, N, K, w[4], scratch, scratch + R, g_opacity, nb_local, g1, g2,
                                                                     H, w[5], w[6], part, d_g1, d_g2);
    k_loss_rays<<<1, kBlock, 0, (hipStream_t)stream>>>(rgb, rgb_gt, depth, depth_gt, normal_map, normal_gt, scratch, scratch + R, R, K, w[0], w[1], w[2], w[3],
                                                        out8, g_rgb, g_depth, g_normal_map, part, nb_eik, 1.f / (float)H, w[4], w[5], w[6]);
    return check_launch();
}

int hs_loss_eikonal(const float *g1, const float *g2, int64_t H, float w_eik, float w_smooth, float *acc2, float *d_g1, float *d_g2, void *stream) {
    if (H < 1) return HS_ERR_ARG;
    if (!g1 || !g2 || !acc2 || !d_g1 || !d_g2) return HS_ERR_NULL;
    const int64_t want = (H + 255) / 256;
    const int grid = (int)(want < 2048 ? want : 2048);
    k_loss_eikonal<<<grid, 256, 0, (hipStream_t)stream>>>(g1, g2, H, w_eik, w_smooth, acc2, d_g1, d_g2);
    return check_launch();
}

int hs_bg_smooth_loss(const float *depth, const float *normal, const int64_t *labels, int32_t side, float *out, float *g_depth, float *g_normal,
                      void *stream) {
    if (side < 1 || side * side > kBlock) return HS_ERR_ARG;
    if (!depth || !normal || !labels || !out || !g_depth || !g_normal) return HS_ERR_NULL;
    k_bg_smooth<<<1, kBlock, 0, (hipStream_t)stream>>>(depth, normal, labels, side, out, g_depth, g_normal);
    return check_launch();
}

}  // extern "C"
