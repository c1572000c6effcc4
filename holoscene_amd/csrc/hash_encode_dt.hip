// holoscene_amd/csrc/hash_encode_dt.hip -- the hash-grid encoder in the reference's OTHER scalar types (gfx950).
//
// The reference instantiates its five kernels for double, float and half (AT_DISPATCH_FLOATING_TYPES_AND_HALF, hashencoder/src/hashencoder.cu:
// 747, 778, 817); HoloScene's Stage 1 runs them in float (no autocast: training/holoscene_train.py:45), and that is the path csrc/hash_encode.hip
// is built around.  This file serves the two other instantiations behind the same three entry points (bindings.cpp:5-9), as plain
// one-thread-per-(point, level) kernels in the reference's order of operations -- they exist for completeness of the native boundary (a double
// gradcheck, a caller under autocast), not for speed.  What a scalar type changes, restated from the reference's expressions under C++'s
// promotion rules and at::Half's operators (every Half operation = the float operation, rounded to half):
//   * positions: (float)inputs[d] * scale -- located in float for every type (:158, :295, :487); the range test is on the stored value;
//   * forward sum `results[ch] += w * grid[..]` (:172-204): double: double product and sum; half: float product, sum rounded to half per corner;
//   * dy_dx `results_grad[ch] += w * (grid[r] - grid[l]) * pos_derivative` (:217-246): the difference in T (half: rounded), the products in
//     double / float, the sum in T;
//   * first backward `atomicAdd(grad_grid, w * grad_cur[c])` (:301-341): double atomics; half: the product rounded to half, added by a half (x2) atomic;
//   * input backward and grad_grad `result += grad * dy_dx` (:361-369, :400-409): product AND sum in T;
//   * second backward, embedding part (:507-592): the eight-corner cache in T, `+=` / `-=` of w * grad * g2 * pos_derivative (double / float).
// Results: bit-identical to oracle/hash_oracle_dt.c for everything but the two scatters (atomic order).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "holoscene_hip.h"

namespace {

constexpr int kT = 256;

__device__ __forceinline__ float sstep(float t) { return t * t * (3.0f - 2.0f * t); }
__device__ __forceinline__ float sstep_d(float t) { return 6 * t * (1.0f - t); }

struct Lvl { float scale; uint32_t res, table, offset; };
__device__ __forceinline__ Lvl level_of(const int32_t *__restrict__ offsets, uint32_t level, float scale) {
    Lvl l;
    l.offset = (uint32_t)offsets[level];
    l.table = (uint32_t)offsets[level + 1] - l.offset;
    l.scale = scale;
    l.res = (uint32_t)ceilf(scale) + 1u;
    return l;
}
template <int D>
__device__ __forceinline__ uint32_t cell_of(const Lvl &l, const uint32_t g[D]) {      // hashencoder.cu:36-72
    const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (int d = 0; d < D; d++) {
        if (stride <= l.table) {
            index += g[d] * stride;
            stride *= l.res;
        }
    }
    if (stride > l.table) {
        index = 0;
#pragma unroll
        for (int d = 0; d < D; d++) index ^= g[d] * primes[d];
    }
    return index % l.table;
}

// ---- scalar-type arithmetic as the reference's expressions evaluate it
template <typename T> struct Num;
template <> struct Num<double> {
    typedef double P;                                   // type of `float * T`
    static __device__ __forceinline__ float to_f(double v) { return (float)v; }
    static __device__ __forceinline__ bool oob(double v) { return v < 0 || v > 1; }
    static __device__ __forceinline__ double zero() { return 0.0; }
    static __device__ __forceinline__ P mul_f(float w, double v) { return (double)w * v; }
    static __device__ __forceinline__ P mul_p(P a, double v) { return a * v; }
    static __device__ __forceinline__ P mul_pf(P a, float f) { return a * (double)f; }
    static __device__ __forceinline__ void acc(double &r, P x) { r += x; }
    static __device__ __forceinline__ void dec(double &r, P x) { r -= x; }
    static __device__ __forceinline__ double sub(double a, double b) { return a - b; }
    static __device__ __forceinline__ double mul(double a, double b) { return a * b; }
    static __device__ __forceinline__ double add(double a, double b) { return a + b; }
    static __device__ __forceinline__ void atomic_add(double *p, P x) { atomicAdd(p, x); }
};
template <> struct Num<__half> {
    typedef float P;
    static __device__ __forceinline__ float to_f(__half v) { return __half2float(v); }
    static __device__ __forceinline__ bool oob(__half v) { const float f = __half2float(v); return f < 0.f || f > 1.f; }
    static __device__ __forceinline__ __half zero() { return __float2half_rn(0.f); }
    static __device__ __forceinline__ P mul_f(float w, __half v) { return w * __half2float(v); }
    static __device__ __forceinline__ P mul_p(P a, __half v) { return a * __half2float(v); }
    static __device__ __forceinline__ P mul_pf(P a, float f) { return a * f; }
    static __device__ __forceinline__ void acc(__half &r, P x) { r = __float2half_rn(__half2float(r) + x); }
    static __device__ __forceinline__ void dec(__half &r, P x) { r = __float2half_rn(__half2float(r) - x); }
    static __device__ __forceinline__ __half sub(__half a, __half b) { return __float2half_rn(__half2float(a) - __half2float(b)); }
    static __device__ __forceinline__ __half mul(__half a, __half b) { return __float2half_rn(__half2float(a) * __half2float(b)); }
    static __device__ __forceinline__ __half add(__half a, __half b) { return __float2half_rn(__half2float(a) + __half2float(b)); }
    // half += half(x) on the 32-bit word that holds it (compare-and-swap: per-lane round-to-nearest add, as the hardware's packed atomic does)
    static __device__ __forceinline__ void atomic_add(__half *p, P x) {
        const __half v = __float2half_rn(x);
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        unsigned int *word = reinterpret_cast<unsigned int *>(a & ~(uintptr_t)3);
        const bool hi = (a & 2) != 0;
        unsigned int old = *word, assumed;
        do {
            assumed = old;
            const unsigned short cur = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
            const __half s = __float2half_rn(__half2float(__ushort_as_half(cur)) + __half2float(v));
            const unsigned int nb = __half_as_ushort(s);
            const unsigned int next = hi ? ((assumed & 0x0000ffffu) | (nb << 16)) : ((assumed & 0xffff0000u) | nb);
            old = atomicCAS(word, assumed, next);
        } while (old != assumed);
    }
};

template <typename T, int D>
__device__ __forceinline__ bool locate_t(const T *__restrict__ x, const Lvl &l, uint32_t g[D], float w[D], float dw[D]) {
#pragma unroll
    for (int d = 0; d < D; d++)
        if (Num<T>::oob(x[d])) return false;
#pragma unroll
    for (int d = 0; d < D; d++) {
        float pos = Num<T>::to_f(x[d]) * l.scale;
        g[d] = (uint32_t)floorf(pos);
        pos -= (float)g[d];
        dw[d] = sstep_d(pos);
        w[d] = sstep(pos);
    }
    return true;
}

struct Scales { float v[HS_MAX_LEVELS]; };

// ------------------------------------------------------------------------------------ forward (+ dy_dx [B, L, D, C])
template <typename T, int D, int C>
__global__ __launch_bounds__(kT) void k_fwd_t(const T *__restrict__ x, const T *__restrict__ emb, const int32_t *__restrict__ offsets, T *__restrict__ out,
                                              T *__restrict__ dydx, uint32_t B, uint32_t L, Scales sc) {
    typedef Num<T> N;
    const uint32_t b = blockIdx.x * kT + threadIdx.x, level = blockIdx.y;
    if (b >= B) return;
    const Lvl l = level_of(offsets, level, sc.v[level]);
    T *o = out + ((size_t)level * B + b) * C;
    T *j = dydx ? dydx + ((size_t)b * L + level) * D * C : nullptr;
    uint32_t g[D];
    float w[D], dw[D];
    if (l.table == 0u || !locate_t<T, D>(x + (size_t)b * D, l, g, w, dw)) {
#pragma unroll
        for (int c = 0; c < C; c++) o[c] = N::zero();
        if (j)
            for (int i = 0; i < D * C; i++) j[i] = N::zero();
        return;
    }
    const T *grid = emb + (size_t)l.offset * C;
    T res[C];
#pragma unroll
    for (int c = 0; c < C; c++) res[c] = N::zero();
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float wt = 1;
        uint32_t gl[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            if ((idx & (1 << d)) == 0) { wt *= 1 - w[d]; gl[d] = g[d]; }
            else { wt *= w[d]; gl[d] = g[d] + 1; }
        }
        const uint32_t cell = cell_of<D>(l, gl);
#pragma unroll
        for (int c = 0; c < C; c++) N::acc(res[c], N::mul_f(wt, grid[(size_t)cell * C + c]));
    }
#pragma unroll
    for (int c = 0; c < C; c++) o[c] = res[c];
    if (!j) return;
#pragma unroll
    for (int gd = 0; gd < D; gd++) {
        T rg[C];
#pragma unroll
        for (int c = 0; c < C; c++) rg[c] = N::zero();
#pragma unroll
        for (int idx = 0; idx < (1 << (D - 1)); idx++) {
            float wt = l.scale;
            uint32_t gl[D];
#pragma unroll
            for (int nd = 0; nd < D - 1; nd++) {
                const int d = (nd >= gd) ? nd + 1 : nd;
                if ((idx & (1 << nd)) == 0) { wt *= 1 - w[d]; gl[d] = g[d]; }
                else { wt *= w[d]; gl[d] = g[d] + 1; }
            }
            gl[gd] = g[gd];
            const uint32_t left = cell_of<D>(l, gl);
            gl[gd] = g[gd] + 1;
            const uint32_t right = cell_of<D>(l, gl);
#pragma unroll
            for (int c = 0; c < C; c++)
                N::acc(rg[c], N::mul_pf(N::mul_f(wt, N::sub(grid[(size_t)right * C + c], grid[(size_t)left * C + c])), dw[gd]));
        }
#pragma unroll
        for (int c = 0; c < C; c++) j[gd * C + c] = rg[c];
    }
}

// ------------------------------------------------------------------------------------ first backward: scatter
template <typename T, int D, int C>
__global__ __launch_bounds__(kT) void k_bwd_t(const T *__restrict__ grad, const T *__restrict__ x, const int32_t *__restrict__ offsets, T *__restrict__ gemb,
                                              uint32_t B, uint32_t L, Scales sc) {
    typedef Num<T> N;
    const uint32_t b = blockIdx.x * kT + threadIdx.x, level = blockIdx.y;
    if (b >= B) return;
    const Lvl l = level_of(offsets, level, sc.v[level]);
    if (l.table == 0u) return;
    uint32_t g[D];
    float w[D], dw[D];
    if (!locate_t<T, D>(x + (size_t)b * D, l, g, w, dw)) return;
    const T *gr = grad + ((size_t)level * B + b) * C;
    T *gg = gemb + (size_t)l.offset * C;
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float wt = 1;
        uint32_t gl[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            if ((idx & (1 << d)) == 0) { wt *= 1 - w[d]; gl[d] = g[d]; }
            else { wt *= w[d]; gl[d] = g[d] + 1; }
        }
        const uint32_t cell = cell_of<D>(l, gl);
#pragma unroll
        for (int c = 0; c < C; c++) N::atomic_add(gg + (size_t)cell * C + c, N::mul_f(wt, gr[c]));
    }
}

// ------------------------------------------------------------------------------------ input backward (:347-372) and grad_grad (:376-428)
template <typename T, int D, int C>
__global__ __launch_bounds__(kT) void k_input_bwd_t(const T *__restrict__ grad, const T *__restrict__ dydx, T *__restrict__ gx, uint32_t B, uint32_t L) {
    typedef Num<T> N;
    const uint32_t t = blockIdx.x * kT + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const T *j = dydx + (size_t)b * L * D * C;
    T r = N::zero();
    for (uint32_t l = 0; l < L; l++)
#pragma unroll
        for (int c = 0; c < C; c++) r = N::add(r, N::mul(grad[((size_t)l * B + b) * C + c], j[(size_t)l * D * C + d * C + c]));
    gx[t] = r;
}

template <typename T, int D, int C>
__global__ __launch_bounds__(kT) void k_gg_t(const T *__restrict__ ggx, const T *__restrict__ dydx, T *__restrict__ gg, uint32_t B, uint32_t L) {
    typedef Num<T> N;
    const uint32_t b = blockIdx.x * kT + threadIdx.x, level = blockIdx.y;
    if (b >= B) return;
    const T *j = dydx + ((size_t)b * L + level) * D * C;
    T r[C];
#pragma unroll
    for (int c = 0; c < C; c++) r[c] = N::zero();
#pragma unroll
    for (int d = 0; d < D; d++)
#pragma unroll
        for (int c = 0; c < C; c++) r[c] = N::add(r[c], N::mul(ggx[(size_t)b * D + d], j[d * C + c]));
#pragma unroll
    for (int c = 0; c < C; c++) gg[((size_t)level * B + b) * C + c] = r[c];
}

// ------------------------------------------------------------------------------------ second backward, embedding part (:432-595)
template <typename T, int D, int C>
__global__ __launch_bounds__(kT) void k_bwd2_t(const T *__restrict__ grad, const T *__restrict__ x, const int32_t *__restrict__ offsets, const T *__restrict__ ggx,
                                               T *__restrict__ g2emb, uint32_t B, uint32_t L, Scales sc) {
    typedef Num<T> N;
    const uint32_t b = blockIdx.x * kT + threadIdx.x, level = blockIdx.y;
    if (b >= B) return;
    const Lvl l = level_of(offsets, level, sc.v[level]);
    if (l.table == 0u) return;
    uint32_t g[D];
    float w[D], dw[D];
    if (!locate_t<T, D>(x + (size_t)b * D, l, g, w, dw)) return;
    const T *gr = grad + ((size_t)level * B + b) * C;
    T cache[(1 << D) * C];
#pragma unroll
    for (int i = 0; i < (1 << D) * C; i++) cache[i] = N::zero();
#pragma unroll
    for (int gd = 0; gd < D; gd++) {
        const T g2 = ggx[(size_t)b * D + gd];
#pragma unroll
        for (int idx = 0; idx < (1 << (D - 1)); idx++) {
            float wt = l.scale;
            int bits = 0;
#pragma unroll
            for (int nd = 0; nd < D - 1; nd++) {
                const int d = (nd >= gd) ? nd + 1 : nd;
                if ((idx & (1 << nd)) == 0) wt *= 1 - w[d];
                else { wt *= w[d]; bits |= 1 << d; }
            }
            const int left = bits, right = bits | (1 << gd);
#pragma unroll
            for (int c = 0; c < C; c++) {
                const typename N::P v = N::mul_pf(N::mul_p(N::mul_f(wt, gr[c]), g2), dw[gd]);
                N::acc(cache[right * C + c], v);
                N::dec(cache[left * C + c], v);
            }
        }
    }
    T *gg = g2emb + (size_t)l.offset * C;
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        uint32_t gl[D];
#pragma unroll
        for (int d = 0; d < D; d++) gl[d] = g[d] + ((idx >> d) & 1);
        const uint32_t cell = cell_of<D>(l, gl);
#pragma unroll
        for (int c = 0; c < C; c++) N::atomic_add(gg + (size_t)cell * C + c, (typename N::P)N::mul_f(1.0f, cache[idx * C + c]));
    }
}

Scales scales_of(uint32_t L, float S, uint32_t H) {
    Scales s;
    for (uint32_t l = 0; l < HS_MAX_LEVELS; l++) s.v[l] = l < L ? exp2f((float)l * S) * (float)H - 1.0f : 0.f;      // host exp2f, as hash_encode.hip
    return s;
}

bool ok_dims(uint32_t D, uint32_t C, uint32_t L) { return (D == 2 || D == 3) && (C == 1 || C == 2 || C == 4 || C == 8) && L >= 1 && L <= HS_MAX_LEVELS; }

template <typename T, class F>
void with_dc(uint32_t D, uint32_t C, F &&f) {
#define HS_DC(d_, c_) if (D == d_ && C == c_) { f(std::integral_constant<int, d_>{}, std::integral_constant<int, c_>{}); return; }
    HS_DC(2, 1) HS_DC(2, 2) HS_DC(2, 4) HS_DC(2, 8) HS_DC(3, 1) HS_DC(3, 2) HS_DC(3, 4) HS_DC(3, 8)
#undef HS_DC
}

template <typename T>
int fwd_t(const void *inputs, const void *embeddings, const int32_t *offsets, void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
          void *dy_dx, hipStream_t st) {
    const Scales sc = scales_of(L, S, H);
    with_dc<T>(D, C, [&](auto d, auto c) {
        k_fwd_t<T, decltype(d)::value, decltype(c)::value><<<dim3((B + kT - 1) / kT, L), kT, 0, st>>>((const T *)inputs, (const T *)embeddings, offsets, (T *)outputs,
                                                                                                      (T *)dy_dx, B, L, sc);
    });
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

template <typename T>
int bwd_t(const void *grad, const void *inputs, const int32_t *offsets, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
          const void *dy_dx, void *grad_inputs, hipStream_t st) {
    const Scales sc = scales_of(L, S, H);
    with_dc<T>(D, C, [&](auto d, auto c) {
        constexpr int D_ = decltype(d)::value, C_ = decltype(c)::value;
        if (grad_embeddings) k_bwd_t<T, D_, C_><<<dim3((B + kT - 1) / kT, L), kT, 0, st>>>((const T *)grad, (const T *)inputs, offsets, (T *)grad_embeddings, B, L, sc);
        if (grad_inputs) k_input_bwd_t<T, D_, C_><<<(B * D_ + kT - 1) / kT, kT, 0, st>>>((const T *)grad, (const T *)dy_dx, (T *)grad_inputs, B, L);
    });
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

template <typename T>
int bwd2_t(const void *grad, const void *inputs, const int32_t *offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const void *dy_dx,
           const void *ggx, void *grad_grad, void *g2emb, hipStream_t st) {
    const Scales sc = scales_of(L, S, H);
    with_dc<T>(D, C, [&](auto d, auto c) {
        constexpr int D_ = decltype(d)::value, C_ = decltype(c)::value;
        if (grad_grad) k_gg_t<T, D_, C_><<<dim3((B + kT - 1) / kT, L), kT, 0, st>>>((const T *)ggx, (const T *)dy_dx, (T *)grad_grad, B, L);
        if (g2emb) k_bwd2_t<T, D_, C_><<<dim3((B + kT - 1) / kT, L), kT, 0, st>>>((const T *)grad, (const T *)inputs, offsets, (const T *)ggx, (T *)g2emb, B, L, sc);
    });
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int hs_hash_encode_forward_dt(int32_t dtype, const void *inputs, const void *embeddings, const int32_t *offsets, void *outputs, uint32_t B, uint32_t D,
                              uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, void *dy_dx, void *stream) {
    if (dtype == HS_DTYPE_F32)
        return hs_hash_encode_forward((const float *)inputs, (const float *)embeddings, offsets, (float *)outputs, B, D, C, L, S, H, calc_grad_inputs, (float *)dy_dx, stream);
    if (dtype != HS_DTYPE_F64 && dtype != HS_DTYPE_F16) return HS_ERR_ARG;
    if (!ok_dims(D, C, L)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!inputs || !embeddings || !offsets || !outputs || (calc_grad_inputs && !dy_dx)) return HS_ERR_NULL;
    void *j = calc_grad_inputs ? dy_dx : nullptr;
    return dtype == HS_DTYPE_F64 ? fwd_t<double>(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, j, (hipStream_t)stream)
                                 : fwd_t<__half>(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, j, (hipStream_t)stream);
}

int hs_hash_encode_backward_dt(int32_t dtype, const void *grad, const void *inputs, const void *embeddings, const int32_t *offsets, void *grad_embeddings,
                               uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void *dy_dx, void *grad_inputs,
                               void *stream) {
    if (dtype == HS_DTYPE_F32)
        return hs_hash_encode_backward((const float *)grad, (const float *)inputs, (const float *)embeddings, offsets, (float *)grad_embeddings, B, D, C, L, S, H,
                                       calc_grad_inputs, (const float *)dy_dx, (float *)grad_inputs, stream);
    (void)embeddings;
    if (dtype != HS_DTYPE_F64 && dtype != HS_DTYPE_F16) return HS_ERR_ARG;
    if (!ok_dims(D, C, L)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!grad || !inputs || !offsets || !grad_embeddings || (calc_grad_inputs && (!dy_dx || !grad_inputs))) return HS_ERR_NULL;
    void *gx = calc_grad_inputs ? grad_inputs : nullptr;
    return dtype == HS_DTYPE_F64 ? bwd_t<double>(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, gx, (hipStream_t)stream)
                                 : bwd_t<__half>(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, gx, (hipStream_t)stream);
}

int hs_hash_encode_second_backward_dt(int32_t dtype, const void *grad, const void *inputs, const void *embeddings, const int32_t *offsets, uint32_t B, uint32_t D,
                                      uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void *dy_dx, const void *grad_grad_inputs,
                                      void *grad_grad, void *grad2_embeddings, void *stream) {
    if (dtype == HS_DTYPE_F32)
        return hs_hash_encode_second_backward((const float *)grad, (const float *)inputs, (const float *)embeddings, offsets, B, D, C, L, S, H, calc_grad_inputs,
                                              (const float *)dy_dx, (const float *)grad_grad_inputs, (float *)grad_grad, (float *)grad2_embeddings, stream);
    (void)embeddings;
    (void)calc_grad_inputs;
    if (dtype != HS_DTYPE_F64 && dtype != HS_DTYPE_F16) return HS_ERR_ARG;
    if (!ok_dims(D, C, L)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!grad || !inputs || !offsets || !dy_dx || !grad_grad_inputs || !grad_grad || !grad2_embeddings) return HS_ERR_NULL;
    return dtype == HS_DTYPE_F64 ? bwd2_t<double>(grad, inputs, offsets, B, D, C, L, S, H, dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings, (hipStream_t)stream)
                                 : bwd2_t<__half>(grad, inputs, offsets, B, D, C, L, S, H, dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings, (hipStream_t)stream);
}

}  // extern "C"
