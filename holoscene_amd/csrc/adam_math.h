// holoscene_amd/csrc/adam_math.h -- the per-element Adam update shared by the flat optimiser kernel (optim.hip) and the
// reduce-and-step kernel of the binned scatter (hash_encode.hip).  Operation for operation torch.optim.Adam's single-tensor
// path (training/holoscene_train.py:156-169 builds it with betas (0.9, 0.99), eps 1e-15).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float step_size, float bc2_sqrt, float beta1, float beta2, float eps,
                                      float gscale) {
    g *= gscale;
    m = m + (1.f - beta1) * (g - m);                 // exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + (1.f - beta2) * g * g;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;   // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - step_size * (m / denom);                 // param.addcdiv_(exp_avg, denom, value=-step_size)
}
