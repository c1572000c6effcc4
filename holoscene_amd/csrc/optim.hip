// holoscene_amd/csrc/optim.hip -- fused dense Adam over one flat parameter buffer (gfx950).
//
// The reference steps torch.optim.Adam over 24.7 M parameters in three groups (hash grids at lr*20, MLPs and
// beta at lr; betas (0.9, 0.99), eps 1e-15) followed by an ExponentialLR step (training/holoscene_train.py:156-169,
// 374, 428).  Here all parameters, gradients and both moments live in four flat fp32 buffers; one streaming kernel
// (16 B per lane, 7 x 4 B of HBM traffic per parameter = the algorithmic minimum) applies the update, reading the
// step count, bias corrections and per-group learning rates from a small device-resident state block that a
// one-thread "tick" kernel advances -- so the whole optimiser is two launches with no host involvement and can sit
// inside a captured HIP graph.  A [begin, end) element range lets each rank update only its shard (ZeRO-1).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"
#include "adam_math.h"

namespace {

constexpr int kThreads = 256;

// advance step, bias corrections and learning rates (lr_g = lr0_g * gamma^(step-1): ExponentialLR stepped after each update)
__global__ void k_adam_tick(hsAdamState *st, float beta1, float beta2, double gamma) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t step = st->step + 1;
    st->step = step;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const double decay = pow(gamma, (double)(step - 1));
    for (int g = 0; g < HS_ADAM_MAX_GROUPS; g++) {
        const double lr = (double)st->lr0[g] * decay;
        st->lr[g] = (float)lr;
        st->step_size[g] = (float)(lr / bc1);
    }
    st->bc2_sqrt = (float)sqrt(bc2);
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool STREAM>
__global__ __launch_bounds__(kThreads) void k_adam_flat(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                         float *__restrict__ v, int64_t begin, int64_t end, const hsAdamState *__restrict__ st,
                                                         float beta1, float beta2, float eps, float gscale, int64_t g_base, int64_t mv_base) {
    const float bc2_sqrt = st->bc2_sqrt;
    const int64_t e0 = st->group_end[0], e1 = st->group_end[1];
    const float s0 = st->step_size[0], s1 = st->step_size[1], s2 = st->step_size[2];
    const int64_t q0 = begin >> 2, q1 = end >> 2;  // begin, end are multiples of 4 (checked by the host wrapper)
    // Walk the buffer from its END to its start: the hash tables sit at the front of the flat layout, so they are the last
    // lines Adam touches and have the best chance of still being in the 256 MB Infinity Cache when the next iteration's
    // first hash-encode gathers from them.  Measured: that launch 124 -> 107 us (the same kernel on a warm table later in
    // the iteration: ~50 us), i.e. the cold-table penalty after the 0.7 GB optimiser sweep is reduced, not removed.
    for (int64_t qq = (int64_t)blockIdx.x * kThreads + threadIdx.x; qq < q1 - q0; qq += (int64_t)gridDim.x * kThreads) {
        const int64_t q = q1 - 1 - qq;
        // g / m / v may be shard-local buffers whose element 0 is flat element g_base / mv_base (ZeRO-1: 1/N of the moment storage)
        const int64_t qg = q - (g_base >> 2), qm = q - (mv_base >> 2);
        float4 pp = reinterpret_cast<float4 *>(p)[q], mm, vv, gg;
        if constexpr (STREAM) {     // gradient and moments are not read again before the next update: streaming (nt) accesses
            const f32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(m) + qm), b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(v) + qm),
                          c = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(g) + qg);
            mm = make_float4(a.x, a.y, a.z, a.w); vv = make_float4(b.x, b.y, b.z, b.w); gg = make_float4(c.x, c.y, c.z, c.w);
        } else {
            mm = reinterpret_cast<float4 *>(m)[qm]; vv = reinterpret_cast<float4 *>(v)[qm]; gg = reinterpret_cast<const float4 *>(g)[qg];
        }
        const int64_t i = q << 2;
        float ss[4];
#pragma unroll
        for (int k = 0; k < 4; k++) ss[k] = (i + k < e0) ? s0 : ((i + k < e1) ? s1 : s2);
        adam1(pp.x, gg.x, mm.x, vv.x, ss[0], bc2_sqrt, beta1, beta2, eps, gscale);
        adam1(pp.y, gg.y, mm.y, vv.y, ss[1], bc2_sqrt, beta1, beta2, eps, gscale);
        adam1(pp.z, gg.z, mm.z, vv.z, ss[2], bc2_sqrt, beta1, beta2, eps, gscale);
        adam1(pp.w, gg.w, mm.w, vv.w, ss[3], bc2_sqrt, beta1, beta2, eps, gscale);
        reinterpret_cast<float4 *>(p)[q] = pp;
        if constexpr (STREAM) {
            __builtin_nontemporal_store(f32x4_t{mm.x, mm.y, mm.z, mm.w}, reinterpret_cast<f32x4_t *>(m) + qm);
            __builtin_nontemporal_store(f32x4_t{vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<f32x4_t *>(v) + qm);
        } else {
            reinterpret_cast<float4 *>(m)[qm] = mm;
            reinterpret_cast<float4 *>(v)[qm] = vv;
        }
    }
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

}  // namespace

// ---- many small tensors -> their slots in a flat buffer, one launch.  (torch._foreach_copy_ over the ~30 MLP gradients of an
// iteration, 1.3 MB, is 24 workgroups that each walk a 64 k-element chunk: 17 us.)
struct CopyJobs { hsCopyJob j[HS_COPY_MAX_JOBS]; };

__global__ __launch_bounds__(256) void k_copy_many(CopyJobs jobs) {
    const hsCopyJob jb = jobs.j[blockIdx.y];
    const bool quads = (((uintptr_t)jb.src | (uintptr_t)jb.dst) & 15) == 0;
    const int64_t nq = quads ? jb.n >> 2 : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256)
        reinterpret_cast<float4 *>(jb.dst)[i] = reinterpret_cast<const float4 *>(jb.src)[i];
    for (int64_t i = 4 * nq + (int64_t)blockIdx.x * 256 + threadIdx.x; i < jb.n; i += (int64_t)gridDim.x * 256) jb.dst[i] = jb.src[i];
}

extern "C" {

int hs_copy_many(const hsCopyJob *jobs, int32_t n_jobs, void *stream) {
    if (n_jobs <= 0) return HS_OK;
    if (!jobs) return HS_ERR_NULL;
    if (n_jobs > HS_COPY_MAX_JOBS) return HS_ERR_ARG;
    CopyJobs cj;
    int64_t most = 0;
    for (int i = 0; i < n_jobs; i++) {
        if (!jobs[i].src || !jobs[i].dst) return HS_ERR_NULL;
        if (jobs[i].n < 0) return HS_ERR_ARG;
        cj.j[i] = jobs[i];
        most = jobs[i].n > most ? jobs[i].n : most;
    }
    int64_t gx = (most / 4 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    k_copy_many<<<dim3((unsigned)gx, (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream>>>(cj);
    return check_launch();
}

int hs_adam_tick(hsAdamState *state, float beta1, float beta2, double gamma, void *stream) {
    if (!state) return HS_ERR_NULL;
    k_adam_tick<<<1, 1, 0, (hipStream_t)stream>>>(state, beta1, beta2, gamma);
    return check_launch();
}

int hs_adam_flat(float *p, const float *g, float *m, float *v, int64_t begin, int64_t end, const hsAdamState *state, float beta1, float beta2,
                 float eps, float grad_scale, void *stream) {
    return hs_adam_flat_shard(p, g, m, v, begin, end, 0, 0, state, beta1, beta2, eps, grad_scale, stream);
}

int hs_adam_flat_shard(float *p, const float *g, float *m, float *v, int64_t begin, int64_t end, int64_t g_base, int64_t mv_base,
                       const hsAdamState *state, float beta1, float beta2, float eps, float grad_scale, void *stream) {
    if (end <= begin) return HS_OK;
    if ((begin & 3) || (end & 3) || (g_base & 3) || (mv_base & 3) || g_base > begin || mv_base > begin || g_base < 0 || mv_base < 0)
        return HS_ERR_ARG;  // flat buffers are padded to whole 16-byte quads; a shard-local buffer starts at or before `begin`
    if (!p || !g || !m || !v || !state) return HS_ERR_NULL;
    const int64_t quads = (end - begin) >> 2;
    int64_t want = (quads + kThreads - 1) / kThreads;
    if (want < 1) want = 1;
    const int grid = (int)(want < 256 * 8 ? want : 256 * 8);
    // HOLOSCENE_ADAM_STREAM=0: plain accesses (A/B).  Eight alternating runs in one session: 2.210-2.219 ms per iteration with the
    // streaming accesses, 2.217-2.226 without (the kernel itself is unchanged at ~125 us; the next iteration's first gathers find
    // more of the parameter tables still cached).
    constexpr bool stream_mv = true;
    if (stream_mv) k_adam_flat<true><<<grid, kThreads, 0, (hipStream_t)stream>>>(p, g, m, v, begin, end, state, beta1, beta2, eps, grad_scale, g_base, mv_base);
    else k_adam_flat<false><<<grid, kThreads, 0, (hipStream_t)stream>>>(p, g, m, v, begin, end, state, beta1, beta2, eps, grad_scale, g_base, mv_base);
    return check_launch();
}

}  // extern "C"
