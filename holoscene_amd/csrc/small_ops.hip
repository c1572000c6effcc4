// holoscene_amd/csrc/small_ops.hip -- the last few hundred floats of a backward stage without a launch per arithmetic operator (gfx950).
//
// A training iteration is ~2 ms of large kernels; what autograd's whole-tensor operators add around them -- "bias gradient = accumulator
// + column 80 of that partial sum", "weight gradient = these 71 columns of the padded result", |beta| + beta_min -- costs a ~5 us launch
// each inside the replayed graph whatever its size (25 such launches were 6 % of the step).  hs_assemble evaluates a list of small
// matrices dst[r, c] = sum over terms of (column-gathered, optionally row-reduced) fp32 sources in ONE launch; hs_abs_shift is the
// density's beta (model/density.py:28-30 of the reference: beta.abs() + beta_min) with its backward.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "holoscene_hip.h"

namespace {

struct AsmJobs { hsAsmJob j[HS_ASM_MAX_JOBS]; };

// A job with a long reduction (a term with red >= 16: column sums over per-workgroup partials) gives every element a whole WAVE -- lane l
// adds blocks l, l + 64, ..., the lanes meet by shuffles -- instead of one thread walking hundreds of dependent-latency loads (30 us for
// 32 sums over 512 blocks); the other jobs keep one thread per element.
__global__ __launch_bounds__(256) void k_assemble(AsmJobs jobs) {
    const hsAsmJob &jb = jobs.j[blockIdx.y];
    const int64_t total = (int64_t)jb.rows * jb.cols;
    bool wide = false;
    for (int t = 0; t < jb.n_terms; t++) wide = wide || jb.term[t].red >= 16;
    const int lane = threadIdx.x & 63;
    const int64_t first = wide ? (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6) : (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * (wide ? 4 : 256);
    for (int64_t i = first; i < total; i += step) {
        const int r = (int)(i / jb.cols), c = (int)(i - (int64_t)r * jb.cols);
        float v = 0.f;
        for (int t = 0; t < jb.n_terms; t++) {
            const hsAsmTerm &tm = jb.term[t];
            const float *p = tm.src + (int64_t)r * tm.ld + (tm.col_map ? tm.col_map[c] : tm.col0 + c);
            float s = 0.f;
            if (wide) {       // eight independent loads in flight per lane (a plain loop is one dependent-latency load at a time: 22 us for 3 136 blocks)
                float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                int k = lane;
                for (; k + 7 * 64 < tm.red; k += 8 * 64) {
#pragma unroll
                    for (int u = 0; u < 8; u++) s8[u] += p[(int64_t)(k + 64 * u) * tm.red_stride];
                }
                for (; k < tm.red; k += 64) s8[0] += p[(int64_t)k * tm.red_stride];
                s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
            } else {
                for (int k = 0; k < tm.red; k++) s += p[(int64_t)k * tm.red_stride];
            }
            v += s;
        }
        if (wide) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane != 0) continue;
        }
        jb.dst[(int64_t)r * jb.dst_ld + c] = v;
    }
}

__global__ void k_abs_shift(const float *__restrict__ x, const float *__restrict__ shift, float *__restrict__ y, const float *__restrict__ gy,
                            float *__restrict__ gx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    if (y) y[i] = fabsf(v) + shift[0];
    if (gx) gx[i] = gy[i] * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));      // torch.sgn: 0 at 0
}

}  // namespace

extern "C" {

int hs_assemble(const hsAsmJob *jobs, int32_t n_jobs, void *stream) {
    if (n_jobs < 0 || n_jobs > HS_ASM_MAX_JOBS) return HS_ERR_ARG;
    if (n_jobs == 0) return HS_OK;
    if (!jobs) return HS_ERR_NULL;
    AsmJobs aj;
    int64_t most = 1;
    for (int i = 0; i < n_jobs; i++) {
        const hsAsmJob &j = jobs[i];
        if (j.rows < 1 || j.cols < 1 || j.n_terms < 1 || j.n_terms > HS_ASM_MAX_TERMS || j.dst_ld < j.cols) return HS_ERR_ARG;
        if (!j.dst) return HS_ERR_NULL;
        for (int t = 0; t < j.n_terms; t++) {
            if (!j.term[t].src) return HS_ERR_NULL;
            if (j.term[t].red < 1) return HS_ERR_ARG;
        }
        aj.j[i] = j;
        int64_t n = (int64_t)j.rows * j.cols;
        for (int t = 0; t < j.n_terms; t++)
            if (j.term[t].red >= 16) { n *= 64; break; }       // a wave per element (k_assemble)
        most = n > most ? n : most;
    }
    const int64_t want = (most + 255) / 256;
    k_assemble<<<dim3((unsigned)(want < 256 ? want : 256), n_jobs), 256, 0, (hipStream_t)stream>>>(aj);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

int hs_abs_shift(const float *x, const float *shift, float *y, const float *gy, float *gx, int32_t n, void *stream) {
    if (n < 0) return HS_ERR_ARG;
    if (n == 0) return HS_OK;
    if (!x || (y && !shift) || (gx && !gy) || (!y && !gx)) return HS_ERR_NULL;
    k_abs_shift<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(x, shift, y, gy, gx, n);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

}  // extern "C"
