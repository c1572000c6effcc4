// holoscene_amd/csrc/mlp_ops.hip -- elementwise stages of the value+Jacobian SDF trunk for gfx950.
//
// The trunk pushes 4 rows per point through every Linear layer: row 0 = value, rows 1..3 = the three
// input tangents (d/dx, d/dy, d/dz).  Between Linears the reference applies Softplus(beta=100)
// (model/network.py:163, 205-206); for the tangent rows the chain rule turns that into a scaling by
// softplus'(a) = sigmoid(100 a).  PyTorch would run ~10 separate elementwise kernels over the
// [B,4,W] activations per layer (hundreds of MB each at B = 100 352); here it is one pass forward and
// one pass backward, 16 B per lane, with the bias gradient reduced in-kernel.
//
//   forward :  v = A[b,0,:] + bias ; out[b,0,:] = softplus100(v) ; out[b,d,:] = sigmoid(100 v) * A[b,d,:]
//   backward:  s = sigmoid(100 v), s' = 100 s (1-s)
//              gA[b,0,:] = s*g[b,0,:] + s' * sum_d A[b,d,:]*g[b,d,:] ;  gA[b,d,:] = s*g[b,d,:] ;  gbias += sum_b gA[b,0,:]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"

#include <hip/hip_bf16.h>

namespace {

constexpr int kThreads = 256;

// 4 consecutive features as fp32, whatever the storage type (fp32: 16 B, bf16: 8 B per lane)
template <class T> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ float4 load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void store(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <> struct Quad<__hip_bfloat16> {
    static __device__ __forceinline__ float4 load(const __hip_bfloat16 *p) {
        const uint2 r = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    }
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float float2_t __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ uint32_t pack(float a, float b) {  // one v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-safe)
        const float2_t v = {a, b};
        const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
        return *reinterpret_cast<const uint32_t *>(&r);
    }
    static __device__ __forceinline__ void store(__hip_bfloat16 *p, float4 v) {
        uint2 r;
        r.x = pack(v.x, v.y);
        r.y = pack(v.z, v.w);
        *reinterpret_cast<uint2 *>(p) = r;
    }
};
constexpr int kPointsPerBlock = 64;

// Softplus(beta=100, threshold=20) and its derivative from ONE exponential:
//   e = exp(100 v);  softplus = log(1+e)/100 (v itself above the threshold);  softplus' = e/(1+e) = sigmoid(100 v).
// Hardware exp/log (v_exp_f32 / v_log_f32): absolute error of the value <= 1e-9 (the log argument is >= 1), of the
// derivative <= 1e-7 -- below fp32 round-off of the GEMMs on either side.  The libm log1pf/expf pair made this
// stage ALU-bound (133 us for 131 072 x 256 bf16 values vs 34 us of HBM time).
struct SpPair { float sp, ds; };
__device__ __forceinline__ SpPair softplus100_pair(float v) {  // branch-free; raw v_exp_f32 / v_log_f32 (base 2)
    const float t = v * 100.f;
    const float e = __builtin_amdgcn_exp2f(fminf(t, 20.f) * 1.44269504f);
    const float one_e = 1.f + e;
    SpPair r;
    const bool lin = t > 20.f;
    r.sp = lin ? v : __builtin_amdgcn_logf(one_e) * (0.69314718f * 0.01f);
    r.ds = lin ? 1.f : e * __builtin_amdgcn_rcpf(one_e);
    return r;
}

// A, out: [B, rows, W]; one thread = one (point, 4 consecutive features); rows = 1 + number of tangents (1..4)
template <int ROWS, class T>
__global__ __launch_bounds__(kThreads) void k_softplus_tangent_fwd(const T *__restrict__ A, const float *__restrict__ bias,
                                                                    T *__restrict__ out, int64_t B, int W) {
    const int quads = W >> 2;
    const int64_t total = B * quads;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int64_t b = i / quads;
        const int q = (int)(i - b * quads);
        const float4 bi = reinterpret_cast<const float4 *>(bias)[q];
        const T *src = A + b * ROWS * W + 4 * q;
        T *dst = out + b * ROWS * W + 4 * q;
        float4 a0 = Quad<T>::load(src);
        a0.x += bi.x; a0.y += bi.y; a0.z += bi.z; a0.w += bi.w;
        const SpPair px = softplus100_pair(a0.x), py = softplus100_pair(a0.y), pz = softplus100_pair(a0.z), pw = softplus100_pair(a0.w);
        Quad<T>::store(dst, make_float4(px.sp, py.sp, pz.sp, pw.sp));
        const float4 s = make_float4(px.ds, py.ds, pz.ds, pw.ds);
#pragma unroll
        for (int r = 1; r < ROWS; r++) {
            const float4 t = Quad<T>::load(src + (size_t)r * W);
            Quad<T>::store(dst + (size_t)r * W, make_float4(s.x * t.x, s.y * t.y, s.z * t.z, s.w * t.w));
        }
    }
}

// grid: ceil(B / kPointsPerBlock) blocks; thread = (point slot tid / quads_per_pass ..., quad)
// FROM_H: `A` holds the layer's OUTPUT instead (value row h = softplus100(v), tangent rows s*A_d) -- what the fused
// matrix-core trunk keeps (sdf_mlp.hip, k_trunk_fwd).  Then s = 1 - exp(-100 h) and s' * A_d = 100 (1-s) * H_d, so the
// pre-activations are never needed (and never stored).
template <int ROWS, class T, bool FROM_H = false>
__global__ __launch_bounds__(kThreads) void k_softplus_tangent_bwd(const T *__restrict__ A, const float *__restrict__ bias,
                                                                    const T *__restrict__ G, T *__restrict__ gA,
                                                                    float *__restrict__ gbias, int64_t B, int W) {
    extern __shared__ float red[];  // [kThreads][4] partial bias sums
    const int quads = W >> 2;
    const int64_t b0 = (int64_t)blockIdx.x * kPointsPerBlock;
    const int64_t b1 = min(b0 + (int64_t)kPointsPerBlock, B);
    const int64_t work = (b1 - b0) * quads;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // thread t always sees the same quad when kThreads % quads == 0 (W = 256 -> quads = 64): then acc is a per-feature partial sum
    for (int64_t i = threadIdx.x; i < work; i += kThreads) {
        const int64_t b = b0 + i / quads;
        const int q = (int)(i % quads);
        const T *a = A + b * ROWS * W + 4 * q;
        const T *g = G + b * ROWS * W + 4 * q;
        T *o = gA + b * ROWS * W + 4 * q;
        float4 v = Quad<T>::load(a);
        float4 s, c;   // s = sigmoid(100 v); c = the factor of sum_d A_d g_d in the value-row gradient
        if (FROM_H) {
            const float k = -100.f * 1.44269504f;
            const float4 e = make_float4(__builtin_amdgcn_exp2f(v.x * k), __builtin_amdgcn_exp2f(v.y * k), __builtin_amdgcn_exp2f(v.z * k),
                                         __builtin_amdgcn_exp2f(v.w * k));
            s = make_float4(1.f - e.x, 1.f - e.y, 1.f - e.z, 1.f - e.w);
            c = make_float4(100.f * e.x, 100.f * e.y, 100.f * e.z, 100.f * e.w);
        } else {
            const float4 bi = reinterpret_cast<const float4 *>(bias)[q];
            v.x += bi.x; v.y += bi.y; v.z += bi.z; v.w += bi.w;
            s = make_float4(softplus100_pair(v.x).ds, softplus100_pair(v.y).ds, softplus100_pair(v.z).ds, softplus100_pair(v.w).ds);
            c = make_float4(100.f * s.x * (1.f - s.x), 100.f * s.y * (1.f - s.y), 100.f * s.z * (1.f - s.z), 100.f * s.w * (1.f - s.w));
        }
        const float4 g0 = Quad<T>::load(g);
        float4 dot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 1; r < ROWS; r++) {
            const float4 t = Quad<T>::load(a + (size_t)r * W), gr = Quad<T>::load(g + (size_t)r * W);
            dot.x += t.x * gr.x; dot.y += t.y * gr.y; dot.z += t.z * gr.z; dot.w += t.w * gr.w;
            Quad<T>::store(o + (size_t)r * W, make_float4(s.x * gr.x, s.y * gr.y, s.z * gr.z, s.w * gr.w));
        }
        float4 r0;
        r0.x = s.x * g0.x + c.x * dot.x;
        r0.y = s.y * g0.y + c.y * dot.y;
        r0.z = s.z * g0.z + c.z * dot.z;
        r0.w = s.w * g0.w + c.w * dot.w;
        Quad<T>::store(o, r0);
        if (kThreads % quads == 0) {
            acc.x += r0.x; acc.y += r0.y; acc.z += r0.z; acc.w += r0.w;
        } else if (gbias) {
            unsafeAtomicAdd(gbias + 4 * q + 0, r0.x); unsafeAtomicAdd(gbias + 4 * q + 1, r0.y);
            unsafeAtomicAdd(gbias + 4 * q + 2, r0.z); unsafeAtomicAdd(gbias + 4 * q + 3, r0.w);
        }
    }
    if (gbias && kThreads % quads == 0) {
        reinterpret_cast<float4 *>(red)[threadIdx.x] = acc;
        __syncthreads();
        if ((int)threadIdx.x < quads) {
            float4 s = acc;
            for (int t = threadIdx.x + quads; t < kThreads; t += quads) {
                const float4 o = reinterpret_cast<float4 *>(red)[t];
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            const int q = threadIdx.x;
            unsafeAtomicAdd(gbias + 4 * q + 0, s.x); unsafeAtomicAdd(gbias + 4 * q + 1, s.y);
            unsafeAtomicAdd(gbias + 4 * q + 2, s.z); unsafeAtomicAdd(gbias + 4 * q + 3, s.w);
        }
    }
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

template <class T>
int launch_fwd(const T *A, const float *bias, T *out, int64_t B, int rows, int W, hipStream_t st) {
    const int64_t total = B * (W >> 2);
    const int64_t want = (total + kThreads - 1) / kThreads;
    const int grid = (int)(want < 256 * 16 ? want : 256 * 16);
    switch (rows) {
        case 1: k_softplus_tangent_fwd<1, T><<<grid, kThreads, 0, st>>>(A, bias, out, B, W); break;
        case 2: k_softplus_tangent_fwd<2, T><<<grid, kThreads, 0, st>>>(A, bias, out, B, W); break;
        case 3: k_softplus_tangent_fwd<3, T><<<grid, kThreads, 0, st>>>(A, bias, out, B, W); break;
        default: k_softplus_tangent_fwd<4, T><<<grid, kThreads, 0, st>>>(A, bias, out, B, W); break;
    }
    return check_launch();
}

template <class T>
int launch_bwd(const T *A, const float *bias, const T *G, T *gA, float *gbias, int64_t B, int rows, int W, hipStream_t st) {
    const int grid = (int)((B + kPointsPerBlock - 1) / kPointsPerBlock);
    const size_t lds = kThreads * 4 * sizeof(float);
    if (!bias) {  // activations-only form (rows == 4, checked by the caller)
        k_softplus_tangent_bwd<4, T, true><<<grid, kThreads, lds, st>>>(A, bias, G, gA, gbias, B, W);
        return check_launch();
    }
    switch (rows) {
        case 1: k_softplus_tangent_bwd<1, T><<<grid, kThreads, lds, st>>>(A, bias, G, gA, gbias, B, W); break;
        case 2: k_softplus_tangent_bwd<2, T><<<grid, kThreads, lds, st>>>(A, bias, G, gA, gbias, B, W); break;
        case 3: k_softplus_tangent_bwd<3, T><<<grid, kThreads, lds, st>>>(A, bias, G, gA, gbias, B, W); break;
        default: k_softplus_tangent_bwd<4, T><<<grid, kThreads, lds, st>>>(A, bias, G, gA, gbias, B, W); break;
    }
    return check_launch();
}

}  // namespace

extern "C" {

int hs_softplus_tangent_fwd(const void *A, const float *bias, void *out, int64_t B, int32_t rows, int32_t W, int32_t dtype, void *stream) {
    if (rows < 1 || rows > 4 || W <= 0 || (W & 3) || (dtype != HS_F32 && dtype != HS_BF16)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!A || !bias || !out) return HS_ERR_NULL;
    if (dtype == HS_F32) return launch_fwd<float>((const float *)A, bias, (float *)out, B, rows, W, (hipStream_t)stream);
    return launch_fwd<__hip_bfloat16>((const __hip_bfloat16 *)A, bias, (__hip_bfloat16 *)out, B, rows, W, (hipStream_t)stream);
}

int hs_softplus_tangent_bwd(const void *A, const float *bias, const void *G, void *gA, float *gbias, int64_t B, int32_t rows, int32_t W,
                            int32_t dtype, void *stream) {
    if (rows < 1 || rows > 4 || W <= 0 || (W & 3) || (dtype != HS_F32 && dtype != HS_BF16)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!A || !bias || !G || !gA) return HS_ERR_NULL;
    if (dtype == HS_F32) return launch_bwd<float>((const float *)A, bias, (const float *)G, (float *)gA, gbias, B, rows, W, (hipStream_t)stream);
    return launch_bwd<__hip_bfloat16>((const __hip_bfloat16 *)A, bias, (const __hip_bfloat16 *)G, (__hip_bfloat16 *)gA, gbias, B, rows, W,
                                      (hipStream_t)stream);
}

int hs_softplus_tangent_bwd_h(const void *H, const void *G, void *gA, float *gbias, int64_t B, int32_t W, int32_t dtype, void *stream) {
    if (W <= 0 || (W & 3) || (dtype != HS_F32 && dtype != HS_BF16)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!H || !G || !gA) return HS_ERR_NULL;
    if (dtype == HS_F32) return launch_bwd<float>((const float *)H, nullptr, (const float *)G, (float *)gA, gbias, B, 4, W, (hipStream_t)stream);
    return launch_bwd<__hip_bfloat16>((const __hip_bfloat16 *)H, nullptr, (const __hip_bfloat16 *)G, (__hip_bfloat16 *)gA, gbias, B, 4, W,
                                      (hipStream_t)stream);
}

}  // extern "C"
