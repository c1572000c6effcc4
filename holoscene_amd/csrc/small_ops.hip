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
#include "assemble.h"

namespace {

__global__ __launch_bounds__(256) void k_assemble(AsmJobs jobs) { assemble_body(jobs, (int)blockIdx.x); }

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
    AsmJobs aj;
    const int rc = fill_asm_jobs(jobs, n_jobs, aj);
    if (rc != HS_OK) return rc;
    if (n_jobs == 0) return HS_OK;
    k_assemble<<<aj.first[n_jobs], 256, 0, (hipStream_t)stream>>>(aj);
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
