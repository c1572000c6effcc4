// holoscene_amd/csrc/sampler.hip -- per-ray kernels of the error-bounded sampler (VolSDF Algorithm 1) for gfx950.
//
// Replaces the ~540 tiny PyTorch kernels per sampler round of the reference's
// ErrorBoundSampler.get_z_vals (model/ray_sampler.py:130-287, get_error_bound :450-458).
//
// One workgroup owns one ray (four waves in the update kernel, two in the draw kernel, one in
// the rest).  The ray's sorted sample depths, SDF values and derived per-section quantities live
// in LDS (<= 6 arrays x M floats, M <= 1024); every cumulative sum is a three-level scan (serial
// inside a thread's contiguous chunk, DPP scan across the lanes of a wave, prefix over the waves
// through LDS), every max likewise, the 10-step beta bisection runs entirely in registers/LDS
// with no global traffic.
//
//   k_sampler_update  merge the round's new samples+SDFs into the sorted set (merge by
//                     rank: both inputs are sorted), d* (Heron bound, :165-178), beta line
//                     search (:181-190), global max(beta) for the convergence test (:204)
//   k_sampler_draw    pdf/cdf from the error bound (:209-219) or from the rendering
//                     weights (:224-232), inverse-CDF sampling (:241-253)
//   k_sampler_final   near/far/extra samples + final sort (:261-276), Eikonal pick (:279-280)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "holoscene_hip.h"
#include "wave_ops.h"

namespace {

__device__ __forceinline__ bool gate_closed(const hsGate &g) { return g.a != nullptr && !(*g.a > *g.b); }

constexpr int kWave = 64;

// wave-level scans and reductions on DPP operands (wave_ops.h)
__device__ __forceinline__ float wave_incl_scan(float v, int) { return hs_wave::incl_scan(v); }
__device__ __forceinline__ float wave_max(float v) { return hs_wave::max(v); }
__device__ __forceinline__ float wave_sum(float v) { return hs_wave::sum(v); }

// Laplace density sigma(s; beta) (model/density.py:21-26)
__device__ __forceinline__ float laplace_sigma(float s, float beta) {
    const float sgn = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
    return (1.f / beta) * (0.5f + 0.5f * sgn * expm1f(-fabsf(s) / beta));
}

// The line search evaluates the bound 11 times per ray and round; with libm expf/expm1f (15-20 instructions each, four per
// section) the update kernel is ALU-bound.  Hardware exponentials (v_exp_f32, 1-2 ulp) change the bound by ~1e-6 relative: a
// bisection decision flips only if the bound lies that close to eps.  The draw kernels keep libm: they place the samples.
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float laplace_sigma_fast(float s, float beta) {
    const float sgn = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
    const float x = -fabsf(s) / beta;
    const float em1 = x > -1e-3f ? x + 0.5f * x * x : __expf(x) - 1.f;   // expm1 near 0 by its series
    return (1.f / beta) * (0.5f + 0.5f * sgn * em1);
}

// sections [lo, hi) handled by this lane
__device__ __forceinline__ void lane_chunk(int n, int lane, int &lo, int &hi) {
    const int ch = (n + kWave - 1) / kWave;
    lo = min(lane * ch, n);
    hi = min(lo + ch, n);
}

// ---- workgroup-wide (kUpd threads = 4 waves per ray) scan / max for the beta line search.  One wave per ray left the
// chip at one wave per SIMD with every lane walking ~10 sections serially through libm exp/expm1: 100 us at 640 sections.
constexpr int kUpd = 256;       // default threads per ray; HOLOSCENE_SAMPLER_UPDATE_THREADS = 64 | 128 | 256 (A/B switch)

__device__ __forceinline__ void chunk_of(int n, int tid, int nthreads, int &lo, int &hi) {
    const int ch = (n + nthreads - 1) / nthreads;
    lo = min(tid * ch, n);
    hi = min(lo + ch, n);
}

// exclusive prefix sums of (a, b) over the workgroup's threads; sc: 2*kUpdWaves floats
template <int NT>
__device__ __forceinline__ void block_excl_scan2(float &a, float &b, float *sc, int tid) {
    constexpr int kUpdWaves = NT / kWave;
    const int lane = tid & 63, w = tid >> 6;
    float ai = a, bi = b;
    hs_wave::incl_scan2(ai, bi);          // (same association as two hs_wave::incl_scan: bit-identical; ~half the instructions)
    float pa = 0.f, pb = 0.f;
    if constexpr (kUpdWaves > 1) {
        if (lane == 63) { sc[w] = ai; sc[kUpdWaves + w] = bi; }
        __syncthreads();
        // prefix over the waves before this one: the left-to-right sums of a loop j < w, written as selects on the wave-uniform w
        // (the loop compiled to ~80 instructions of scalar control flow per evaluation of an issue-bound kernel)
        float va[kUpdWaves], vb[kUpdWaves];
#pragma unroll
        for (int j = 0; j < kUpdWaves; j++) { va[j] = sc[j]; vb[j] = sc[kUpdWaves + j]; }
#pragma unroll
        for (int j = 0; j < kUpdWaves - 1; j++) {
            pa = j < w ? pa + va[j] : pa;
            pb = j < w ? pb + vb[j] : pb;
        }
    }
    a = pa + (ai - a);
    b = pb + (bi - b);
}

template <int NT>
__device__ __forceinline__ float block_max(float v, float *sc, int tid) {   // sc: kUpdWaves floats, distinct from the scan's
    constexpr int kUpdWaves = NT / kWave;
    v = wave_max(v);
    if constexpr (kUpdWaves == 1) return v;
    if ((tid & 63) == 0) sc[tid >> 6] = v;
    __syncthreads();
    float r = sc[0];
#pragma unroll
    for (int j = 1; j < kUpdWaves; j++) r = fmaxf(r, sc[j]);
    return r;
}

template <int NT>
__device__ __forceinline__ float block_sum(float v, float *sc, int tid) {   // sc: NT / 64 floats, distinct from the scan's
    constexpr int kW = NT / kWave;
    v = wave_sum(v);
    if constexpr (kW == 1) return v;
    if ((tid & 63) == 0) sc[tid >> 6] = v;
    __syncthreads();
    float r = sc[0];
#pragma unroll
    for (int j = 1; j < kW; j++) r += sc[j];
    return r;
}

// max_i (min(exp(E_i), 1e6) - 1) * exp(-F_i), E inclusive cumsum of err terms, F exclusive cumsum of free energy
// (ray_sampler.py:450-458).  sdf[0..n], dists/dstar[0..n); fe/ee: per-section scratch (each thread touches only its chunk)
template <int NT>
__device__ float error_bound(const float *__restrict__ sdf, const float *__restrict__ dists, const float *__restrict__ dstar,
                             float *__restrict__ fe, float *__restrict__ ee, int n, float beta, int tid, float *sc) {
    int lo, hi;
    chunk_of(n, tid, NT, lo, hi);
    // The kernel is instruction-issue-bound (1 024 rays x 4 waves x 11 evaluations of this function; tools/exp/sampler_prof.hip: an
    // evaluation takes ~2 650 cycles with four waves sharing each SIMD = ~660 issue cycles per wave = ~165 wave instructions at 4
    // cycles each.  A speculative search that resolves two bisection levels per pass -- mid and both candidates for the next
    // midpoint in one evaluation, bit-identical results -- does 1.5x the work in 6 instead of 11 passes and was SLOWER, 27.5 ->
    // 34.8 us: there is no latency chain to shorten): the four IEEE divisions per
    // section of the textbook form (|s| / beta, d* / beta, 1 / beta, 1 / (4 beta^2); ~10 instructions each) become multiplications by
    // ONE reciprocal per evaluation.  That moves the bound by <= 1 ulp per factor -- the same order as the hardware exponentials
    // above, and like them it can flip a bisection decision only if the bound lies that close to eps.
    const float inv_b = 1.f / beta, q = 0.25f * inv_b * inv_b, half_inv_b = 0.5f * inv_b;
    float fsum = 0.f, esum = 0.f;
    for (int i = lo; i < hi; i++) {
        const float s = sdf[i], x = -fabsf(s) * inv_b, d = dists[i];
        const float ser = x + 0.5f * x * x, ex = __expf(x) - 1.f;                 // both sides evaluated: a select, not a divergent branch
        const float em1 = x > -1e-3f ? ser : ex;                                  // expm1 near 0 by its series
        const float sig = half_inv_b + half_inv_b * (s > 0.f ? em1 : (s < 0.f ? -em1 : 0.f));   // Laplace density (model/density.py:21-26)
        const float f_i = d * sig;
        const float e_i = fast_exp(-dstar[i] * inv_b) * (d * d) * q;
        fe[i] = f_i; ee[i] = e_i;
        fsum += f_i; esum += e_i;
    }
    float f = fsum, e = esum;
    block_excl_scan2<NT>(f, e, sc, tid);
    float best = -INFINITY;
    for (int i = lo; i < hi; i++) {
        e += ee[i];
        const float b = (fminf(fast_exp(e), 1.0e6f) - 1.0f) * fast_exp(-f);
        best = fmaxf(best, b);
        f += fe[i];
    }
    return block_max<NT>(best, sc + 2 * (NT / kWave), tid);
}

// error_bound with the thread's sections held in REGISTERS across the eleven evaluations of a line search (HOLOSCENE_SAMPLER_SEARCH=lds restores
// the form above).  A thread owns the same <= CH sections in both sweeps of every evaluation, so what the LDS form re-reads per evaluation (s, d, d*)
// and parks between its sweeps (fe, ee) never has to leave the lane; what does not depend on beta is formed once (-|s|, sign(s), d^2, -d*); the
// loops are unrolled over CH with the tail slots padded by d = 0 (their terms are exact zeros: the sums are untouched) and masked out of the
// maximum; the scans and the maximum take their DPP operand inside the add / max (wave_ops.h: incl_scan2, max_dpp).  Same operations on the same
// values in the same order as the form above -- sign(s) * em1 for its select (an exact product; the zero it yields for s = 0 may carry a sign,
// which the following addition of half_inv_b absorbs) -- hence bit-identical bounds for finite SDF values (a NaN value gives 0 * NaN = NaN
// here where the select gives 0: both forms then drop that section's bound in fmaxf, the sums differ); ~315 -> ~200 instructions per evaluation at three sections.
template <int CH>
struct SecRegs {
    float nas[CH], sg[CH], d[CH], dd[CH], nds[CH];
    int cnt;
};

template <int NT, int CH>
__device__ __forceinline__ SecRegs<CH> load_sections(const float *__restrict__ sdf, const float *__restrict__ dists, const float *__restrict__ dstar,
                                                     int n, int tid) {
    SecRegs<CH> r;
    int lo, hi;
    chunk_of(n, tid, NT, lo, hi);
    r.cnt = hi - lo;
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const bool in = k < r.cnt;
        const float s = in ? sdf[lo + k] : 0.f, d = in ? dists[lo + k] : 0.f, ds = in ? dstar[lo + k] : 0.f;
        r.nas[k] = -fabsf(s);
        r.sg[k] = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
        r.d[k] = d;
        r.dd[k] = d * d;
        r.nds[k] = -ds;
    }
    return r;
}

template <int NT, int CH>
__device__ __forceinline__ float error_bound_regs(const SecRegs<CH> &r, float beta, int tid, float *sc) {
    constexpr int kW = NT / kWave;
    const float inv_b = 1.f / beta, q = 0.25f * inv_b * inv_b, half_inv_b = 0.5f * inv_b;
    float fi[CH], ei[CH], fsum = 0.f, esum = 0.f;
#pragma unroll
    for (int k = 0; k < CH; k++) {
        const float x = r.nas[k] * inv_b;
        const float ser = x + 0.5f * x * x, ex = __expf(x) - 1.f;
        const float em1 = x > -1e-3f ? ser : ex;
        const float sig = half_inv_b + half_inv_b * (r.sg[k] * em1);
        fi[k] = r.d[k] * sig;
        ei[k] = fast_exp(r.nds[k] * inv_b) * r.dd[k] * q;
        fsum += fi[k]; esum += ei[k];
    }
    // exclusive prefixes over the workgroup's threads (block_excl_scan2, with the folded-DPP wave scan)
    float f = fsum, e = esum;
    {
        const int lane = tid & 63, w = tid >> 6;
        float ai = f, bi = e;
        hs_wave::incl_scan2(ai, bi);
        float pa = 0.f, pb = 0.f;
        if constexpr (kW > 1) {
            if (lane == 63) { sc[w] = ai; sc[kW + w] = bi; }
            __syncthreads();
            float va[kW], vb[kW];
#pragma unroll
            for (int j = 0; j < kW; j++) { va[j] = sc[j]; vb[j] = sc[kW + j]; }
#pragma unroll
            for (int j = 0; j < kW - 1; j++) {
                pa = j < w ? pa + va[j] : pa;
                pb = j < w ? pb + vb[j] : pb;
            }
        }
        f = pa + (ai - f);
        e = pb + (bi - e);
    }
    float best = -INFINITY;
#pragma unroll
    for (int k = 0; k < CH; k++) {
        e += ei[k];
        const float b = (fminf(fast_exp(e), 1.0e6f) - 1.0f) * fast_exp(-f);
        best = k < r.cnt ? fmaxf(best, b) : best;
        f += fi[k];
    }
    // block_max; a NaN bound (NaN inputs) is dropped by fmaxf above as in the LDS form, so no lane passes a NaN on
    float *scm = sc + 2 * kW;
    best = hs_wave::max_dpp(best);
    if constexpr (kW == 1) return best;
    if ((tid & 63) == 0) scm[tid >> 6] = best;
    __syncthreads();
    float m = scm[0];
#pragma unroll
    for (int j = 1; j < kW; j++) m = fmaxf(m, scm[j]);
    return m;
}

__device__ __forceinline__ int lower_bound(const float *a, int n, float v) {  // # elements < v
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int upper_bound(const float *a, int n, float v) {  // # elements <= v
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void atomic_max_float(float *addr, float v) {  // v >= 0
    atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

// ------------------------------------------------------------------------------------ draw
// mode 0: pdf ~ error-bound opacity (+tiny); mode 1: pdf ~ rendering weights (+1e-5).  u: explicit [R,n_out] or NULL = linspace(0,1,n_out)
// Optional fused duties of a draw launch in the device-controlled loop (hs_sampler_draw_step):
//  * the loop-control step that used to be its own 1-thread launch: every workgroup derives the post-round state from ctl_in
//    (read-only here) + the round's beta_max, workgroup 0 publishes it to ctl_out (a different slot: no intra-launch race);
//  * the positions of the drawn depths (k_ray_points' arithmetic, same roundings) for the next SDF sweep.
struct DrawExt {
    const hsSamplerCtl *ctl_in;
    hsSamplerCtl *ctl_out;
    const float *beta_max, *beta0;
    int s_new, max_rounds;
    int n_steps;            // how many times the step rule is applied, on beta_max[0 .. n_steps) (hs_sampler_draw_steps); 0 means 1
    const float *o, *d;
    float *x, *x01;
    float divide_factor;
};

// The tail of Algorithm 1 in the SAME launch as the final draw (hs_sampler_tail): the extra-sample indices (k_sampler_pick's partial
// Fisher-Yates shuffle, recomputed by every workgroup from the same draws: 32 dependent LDS steps against a 5 us launch) and the merge of
// [drawn samples | near | far | z[pick]] into the sorted output row (k_sampler_final).  z_out == NULL: a plain draw.
struct TailExt {
    const float *u_pick;          // NULL: eval mode (torch.linspace(0, m - 1, n_extra).long()) unless pick_in
    const int64_t *pick_in;       // explicit indices (parity tests inject the reference's permutation)
    int n_extra;
    float near, far;
    const float *near_r, *far_r;
    const int64_t *eik_idx;
    const float *eik_u;
    float *z_out, *z_eik;
};

// The draw itself, on a ray whose merged depths / SDF values are in LDS (z, sdf: m entries; cdf, pdf: scratch of m floats each; sc: 3 x
// NT / 64 floats): shared by k_sampler_draw (which stages the ray first) and by the fused update + draw kernel (whose update phase
// has just produced them).  All NT threads of the workgroup call it.
template <int NT>
__device__ __forceinline__ void draw_phase(const float *__restrict__ z, const float *__restrict__ sdf, float *__restrict__ cdf, float *__restrict__ pdf,
                                           float *__restrict__ sc, int m, float beta, int mode, float add_tiny, const float *__restrict__ u_in, int n_out,
                                           float *__restrict__ out, int r, int lane, const DrawExt &ext) {
    constexpr int kDraw = NT;
    const int n = m - 1;
    int lo, hi;
    chunk_of(n, lane, kDraw, lo, hi);
    // transmittance at the start of each section (exclusive scan of free energy)
    float fsum = 0.f, esum = 0.f;
    for (int i = lo; i < hi; i++) {
        const float d = z[i + 1] - z[i];
        fsum += d * laplace_sigma(sdf[i], beta);
        if (mode == 0) {
            const float a = d, b = fabsf(sdf[i]), c = fabsf(sdf[i + 1]);
            const bool first = a * a + b * b <= c * c, second = a * a + c * c <= b * b;
            float ds = 0.f;
            if (first) ds = b;
            if (second) ds = c;
            if (!first && !second && (b + c - a > 0.f)) {
                const float s = (a + b + c) / 2.0f;
                ds = (2.0f * sqrtf(s * (s - a) * (s - b) * (s - c))) / a;
            }
            const float sa = (sdf[i] > 0.f) - (sdf[i] < 0.f), sb = (sdf[i + 1] > 0.f) - (sdf[i + 1] < 0.f);
            if (sa * sb != 1.f) ds = 0.f;
            esum += expf(-ds / beta) * (d * d) / (4.f * beta * beta);
            pdf[i] = ds;  // park d* for the second sweep
        }
    }
    float f = fsum, e = esum;
    block_excl_scan2<kDraw>(f, e, sc, lane);      // exclusive prefixes over the ray's threads
    if (mode != 0) e = 0.f;
    float psum = 0.f;
    for (int i = lo; i < hi; i++) {
        const float d = z[i + 1] - z[i];
        const float fe = d * laplace_sigma(sdf[i], beta);
        const float T = expf(-f);
        float p;
        if (mode == 0) {
            e += expf(-pdf[i] / beta) * (d * d) / (4.f * beta * beta);
            p = (fminf(expf(e), 1.0e6f) - 1.0f) * T + add_tiny;
        } else {
            p = (1.f - expf(-fe)) * T + 1e-5f;
        }
        pdf[i] = p;
        psum += p;
        f += fe;
    }
    const float total = block_sum<kDraw>(psum, sc + 2 * (kDraw / kWave), lane);
    // cdf[0] = 0, cdf[i+1] = cumsum(pdf/total)
    float csum = 0.f;
    for (int i = lo; i < hi; i++) { pdf[i] = pdf[i] / total; csum += pdf[i]; }
    float c = csum, unused = 0.f;
    block_excl_scan2<kDraw>(c, unused, sc, lane);
    for (int i = lo; i < hi; i++) { c += pdf[i]; cdf[i + 1] = c; }
    if (lane == 0) cdf[0] = 0.f;
    __syncthreads();
    for (int j = lane; j < n_out; j += kDraw) {
        float u;
        if (u_in) {
            u = u_in[(size_t)r * n_out + j];
        } else {  // torch.linspace(0, 1, n_out): symmetric evaluation around the midpoint
            const float step = 1.0f / (float)(n_out - 1);
            u = (j < n_out / 2) ? step * (float)j : 1.0f - step * (float)(n_out - 1 - j);
        }
        const int inds = upper_bound(cdf, m, u);  // searchsorted(right=True)
        const int below = max(inds - 1, 0), above = min(inds, m - 1);
        const float c0 = cdf[below], c1 = cdf[above], b0 = z[below], b1 = z[above];
        float den = c1 - c0;
        if (den < 1e-5f) den = 1.f;
        const float zj = b0 + (u - c0) / den * (b1 - b0);
        out[(size_t)r * n_out + j] = zj;
        if (ext.x) {
            const float inv_df = __fdiv_rn(1.0f, ext.divide_factor);
            const size_t p = ((size_t)r * n_out + j) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float v = __fadd_rn(ext.o[r * 3 + c], __fmul_rn(zj, ext.d[r * 3 + c]));
                ext.x[p + c] = v;
                ext.x01[p + c] = __fmul_rn(__fadd_rn(__fmul_rn(v, inv_df), 1.0f), 0.5f);
            }
        }
    }
}

#ifdef HS_SAMPLER_PROFILE      // tools/exp/sampler_prof.hip: s_memtime stamps of the phases of one ray's update
__device__ unsigned long long g_sampler_prof[1024 * 8];
#define HS_SSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_sampler_prof[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HS_SSTAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------ update
// The draw of the NEXT round's depths fused into the update launch (hs_sampler_update_draw): the ray's merged set is in LDS already and
// beta has just been found, so the separate draw launch's staging and its launch go away.  Speculative -- whether another round
// runs is only known once every ray has reported its beta -- but the drawn depths and positions are only ever read by the next
// round's kernels, which are gated on this round's max beta.
struct UpdDraw {
    float *out;             // [R, n_out] next depths, or NULL: plain update
    int n_out;
    float add_tiny;
    DrawExt ext;            // only the position outputs (o, d, x, x01, divide_factor) are used
};

template <int kUpd>
__global__ __launch_bounds__(kUpd) void k_sampler_update(float *__restrict__ z_io, float *__restrict__ sdf_io, int ld, int m_old,
                                                           const float *__restrict__ samples, const float *__restrict__ new_sdf, int s_new,
                                                           float *__restrict__ beta_io, const float *__restrict__ beta0_p, float eps,
                                                           int beta_iters, float *__restrict__ beta_max, int R, hsGate gate, const int32_t *__restrict__ m_dev,
                                                           UpdDraw dr, bool search_in_regs) {
    extern __shared__ float lds[];
    if (gate_closed(gate)) return;
    const int r = blockIdx.x, lane = threadIdx.x;   // lane = thread index inside the ray's workgroup (4 waves)
    if (r >= R) return;
    if (m_dev) m_old = *m_dev;                      // device-controlled rounds: the merged count lives in hsSamplerCtl
    HS_SSTAMP(0);
    const int m = m_old + s_new;
    float *z = lds, *sdf = lds + m, *dists = lds + 2 * m, *dstar = lds + 3 * m, *tz = lds + 4 * m, *ts = lds + 5 * m, *sc = lds + 6 * m;
    float *zr = z_io + (size_t)r * ld, *sr = sdf_io + (size_t)r * ld;
    const float *nz = samples + (size_t)r * s_new, *ns = new_sdf + (size_t)r * s_new;
    // stage old set (tz[0..m_old)) and new samples (tz[m_old..m))
    for (int i = lane; i < m_old; i += kUpd) { tz[i] = zr[i]; ts[i] = sr[i]; }
    for (int i = lane; i < s_new; i += kUpd) { tz[m_old + i] = nz[i]; ts[m_old + i] = ns[i]; }
    __syncthreads();
    HS_SSTAMP(1);
    // stable merge by rank (old before new on ties).  (Measured and rejected, round 5: the thread's searches as fixed-step descents advancing
    // together -- one LDS latency per step for all of them -- took 8.1 k ticks against these loops' 4.3 k: four waves share a SIMD, the phase is
    // bound by its instruction count like the rest of the kernel, not by the latency of its dependent reads.)
    for (int i = lane; i < m_old; i += kUpd) {
        const int p = i + lower_bound(tz + m_old, s_new, tz[i]);
        z[p] = tz[i]; sdf[p] = ts[i];
    }
    for (int i = lane; i < s_new; i += kUpd) {
        const int p = i + upper_bound(tz, m_old, tz[m_old + i]);
        z[p] = tz[m_old + i]; sdf[p] = ts[m_old + i];
    }
    __syncthreads();
    HS_SSTAMP(2);
    for (int i = lane; i < m; i += kUpd) { zr[i] = z[i]; sr[i] = sdf[i]; }
    const int n = m - 1;
    for (int i = lane; i < n; i += kUpd) {  // Theorem 1 bound d* per section
        const float a = z[i + 1] - z[i], b = fabsf(sdf[i]), c = fabsf(sdf[i + 1]);
        const bool first = a * a + b * b <= c * c, second = a * a + c * c <= b * b;
        float d = 0.f;
        if (first) d = b;
        if (second) d = c;
        if (!first && !second && (b + c - a > 0.f)) {
            const float s = (a + b + c) / 2.0f;
            d = (2.0f * sqrtf(s * (s - a) * (s - b) * (s - c))) / a;
        }
        const float sa = (sdf[i] > 0.f) - (sdf[i] < 0.f), sb = (sdf[i + 1] > 0.f) - (sdf[i + 1] < 0.f);
        dists[i] = a;
        dstar[i] = (sa * sb == 1.f) ? d : 0.f;
    }
    __syncthreads();
    HS_SSTAMP(3);
    const float beta0 = *beta0_p;
    float hi = beta_io[r];
    // the line search over an evaluator of the bound: eb(beta)
    auto search = [&](auto &&eb) {
        if (eb(beta0) <= eps) hi = beta0;
        HS_SSTAMP(4);
        float lo = beta0;
        // a ray whose bound already holds at beta0 has hi == lo == beta0: every midpoint is beta0 and neither end can move, so the
        // line search is skipped (exactly the reference's result; these workgroups retire ~11x sooner)
        if (hi != lo)
        for (int it = 0; it < beta_iters; it++) {
            const float mid = (lo + hi) / 2.f;
            const float err = eb(mid);
            if (err <= eps) hi = mid;
            else if (err > eps) lo = mid;  // (a NaN bound moves neither end, as in the reference's masked assignments)
        }
    };
    // (Measured and rejected, round 5: the fused draw on the same section registers -- no second Heron bound, density and error term of a section
    // evaluated once instead of twice.  The draw's share of a ray fell from 8.3 k to 6.2 k ticks in tools/exp/sampler_prof.hip, the launch inside
    // the replayed iteration did not move: 21.6 -> 21.9 us, rocprofv3 on one box.  Removed again.)
    auto search_regs = [&](auto chc) {
        constexpr int CH = decltype(chc)::value;
        const SecRegs<CH> regs = load_sections<kUpd, CH>(sdf, dists, dstar, n, lane);
        search([&](float beta) { return error_bound_regs<kUpd, CH>(regs, beta, lane, sc); });
    };
    const int ch = (n + kUpd - 1) / kUpd;       // sections per thread (workgroup-uniform)
    if (search_in_regs && ch == 1) search_regs(std::integral_constant<int, 1>{});
    else if (search_in_regs && ch == 2) search_regs(std::integral_constant<int, 2>{});
    else if (search_in_regs && ch == 3) search_regs(std::integral_constant<int, 3>{});
    else if (search_in_regs && ch == 4) search_regs(std::integral_constant<int, 4>{});
    else search([&](float beta) { return error_bound<kUpd>(sdf, dists, dstar, tz, ts, n, beta, lane, sc); });
    HS_SSTAMP(5);
    if (lane == 0) {
        beta_io[r] = hi;
        // only rays still above beta0 can make the round's test (max beta > beta0, ray_sampler.py:204) true; the others skip the
        // same-address atomic (1 024 of them per launch serialise in the L2)
        if (hi > beta0) atomic_max_float(beta_max, hi);
    }
    if (dr.out) {           // workgroup-uniform
        __syncthreads();    // the line search's scratch (tz, ts, sc) is free from here on
        draw_phase<kUpd>(z, sdf, tz, ts, sc, m, hi, 0, dr.add_tiny, nullptr, dr.n_out, dr.out, r, lane, dr.ext);
        HS_SSTAMP(6);
    }
}


// ------------------------------------------------------------------------------------ device-side round control
// Algorithm 1's loop test (ray_sampler.py:204, 130-287) without the host: after round r's update kernel, one thread
// advances the counters and clears `running` when max beta no longer exceeds beta0 or the round budget is spent.
// Every later kernel of the (fully unrolled) loop is gated on `running` and returns at once.
__global__ void k_sampler_step(hsSamplerCtl *ctl, const float *__restrict__ beta_max, const float *__restrict__ beta0, int s_new, int max_rounds) {
    if (!(ctl->running > ctl->half)) return;
    ctl->m += s_new;
    ctl->rounds += 1;
    if (!(*beta_max > *beta0) || ctl->rounds >= max_rounds) ctl->running = 0.f;
}

// n_extra DISTINCT indices uniformly from [0, m): the first n_extra entries of a random permutation (ray_sampler.py:269,
// torch.randperm(m)[:n_extra]) by a partial Fisher-Yates shuffle driven by u[j] ~ U[0,1); u == NULL: eval mode,
// torch.linspace(0, m-1, n_extra).long() (:271).
__global__ void k_sampler_pick(const hsSamplerCtl *__restrict__ ctl, const float *__restrict__ u, int n_extra, int64_t *__restrict__ pick) {
    __shared__ int idx[HS_SAMPLER_MAX_M];
    const int m = ctl->m;
    if (u == nullptr) {
        for (int j = threadIdx.x; j < n_extra; j += blockDim.x) {
            const float step = (float)(m - 1) / (float)(n_extra - 1 > 0 ? n_extra - 1 : 1);
            const float v = j < n_extra / 2 ? step * (float)j : (float)(m - 1) - step * (float)(n_extra - 1 - j);
            pick[j] = (int64_t)v;
        }
        return;
    }
    for (int i = threadIdx.x; i < m; i += blockDim.x) idx[i] = i;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int j = 0; j < n_extra && j < m; j++) {
            int k = j + (int)(u[j] * (float)(m - j));
            k = k > m - 1 ? m - 1 : k;
            const int t = idx[j]; idx[j] = idx[k]; idx[k] = t;
            pick[j] = idx[j];
        }
    }
}

// ------------------------------------------------------------------------------------ draw (own launch)
// kDraw threads per ray (default 128: one wave per ray leaves a SIMD with ONE wave walking 6-7 sections through libm exp / expm1
// with every LDS and transcendental latency exposed -- 15.9-17.6 us per launch at 64 threads, 11.9-12.4 at 128, 11.6-15.3 at 256 in
// the iteration; HOLOSCENE_SAMPLER_DRAW_THREADS = 64 | 128 | 256 for A/B).
// The cumulative sums are chunk sums in section order + a scan over the threads, so their rounding depends on the chunking -- as
// it does against torch.cumsum in any case; the parity tests bound the drawn depths, not the bit pattern.
template <int kDraw>
__global__ __launch_bounds__(kDraw) void k_sampler_draw(const float *__restrict__ z_in, const float *__restrict__ sdf_in, int ld, int m,
                                                         const float *__restrict__ beta_in, int mode, float add_tiny, const float *__restrict__ u_in,
                                                         int n_out, float *__restrict__ out, int R, hsGate gate, const int32_t *__restrict__ m_dev,
                                                         DrawExt ext, TailExt tail) {
    extern __shared__ float lds[];
    if (ext.ctl_in) {
        hsSamplerCtl c = *ext.ctl_in;
        const int steps = ext.n_steps > 0 ? ext.n_steps : 1;
        for (int k = 0; k < steps; k++)
            if (c.running > c.half) {   // k_sampler_step, once per round that has run
                c.m += ext.s_new;
                c.rounds += 1;
                if (!(ext.beta_max[k] > *ext.beta0) || c.rounds >= ext.max_rounds) c.running = 0.f;
            }
        if (blockIdx.x == 0 && threadIdx.x == 0) *ext.ctl_out = c;
        if (mode == 0 && !(c.running > c.half)) return;
        m = c.m;
    } else {
        if (gate_closed(gate)) return;
        if (m_dev) m = *m_dev;
    }
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= R) return;
    float *z = lds, *cdf = lds + m, *pdf = lds + 2 * m;
    const float *zr = z_in + (size_t)r * ld, *sr = sdf_in + (size_t)r * ld;
    float *sdf = lds + 3 * m, *sc = lds + 4 * m;      // sc: 2 * waves (scan) + waves (sum) floats
    for (int i = lane; i < m; i += kDraw) { z[i] = zr[i]; sdf[i] = sr[i]; }
    __syncthreads();
    draw_phase<kDraw>(z, sdf, cdf, pdf, sc, m, beta_in[r], mode, add_tiny, u_in, n_out, out, r, lane, ext);
    if (tail.z_out == nullptr) return;
    // ---- tail: pick, merge, sort (the ray's merged depths are still in z; cdf / pdf / sdf are scratch from here on)
    __syncthreads();          // every inverse-CDF read is done; the drawn depths (global, written by this workgroup) are visible to it
    const int n_extra = tail.n_extra, n = n_out + 2 + n_extra;      // n <= s_new (launcher) <= m (the merged set: a device-side count)
    if (n > m || n_extra > m) {       // a merged set smaller than the tail's n-float scratch rows (pdf, sdf: m floats each) or than the pick: the
                                      // contract of hs_sampler_tail is broken -- write NaN depths (loud downstream), nothing out of bounds
        for (int i = lane; i < n; i += kDraw) tail.z_out[(size_t)r * n + i] = NAN;
        if (lane == 0 && tail.z_eik) tail.z_eik[r] = NAN;
        return;
    }
    int *idx = reinterpret_cast<int *>(cdf);                          // m ints
    float *v = pdf, *sorted = sdf;                                    // n floats each
    int *pk = reinterpret_cast<int *>(lds + 4 * m + 3 * 4);           // n_extra ints behind the draw's scratch (launcher sizes it)
    if (n_extra > 0) {
        if (tail.pick_in) {
            for (int j = lane; j < n_extra; j += kDraw) pk[j] = (int)tail.pick_in[j];
        } else if (tail.u_pick == nullptr) {
            for (int j = lane; j < n_extra; j += kDraw) {
                const float step = (float)(m - 1) / (float)(n_extra - 1 > 0 ? n_extra - 1 : 1);
                const float t = j < n_extra / 2 ? step * (float)j : (float)(m - 1) - step * (float)(n_extra - 1 - j);
                pk[j] = (int)t;
            }
        } else {
            for (int i = lane; i < m; i += kDraw) idx[i] = i;
            float *up = v;            // the draws through LDS: the shuffle below is a serial chain, a global load per step would be its whole cost
            for (int j = lane; j < n_extra; j += kDraw) up[j] = tail.u_pick[j];
            __syncthreads();
            if (lane == 0) {
                for (int j = 0; j < n_extra && j < m; j++) {
                    int k = j + (int)(up[j] * (float)(m - j));
                    k = k > m - 1 ? m - 1 : k;
                    const int t = idx[j]; idx[j] = idx[k]; idx[k] = t;
                    pk[j] = idx[j];
                }
            }
        }
        __syncthreads();
    }
    for (int i = lane; i < n_out; i += kDraw) v[i] = out[(size_t)r * n_out + i];
    if (lane == 0) { v[n_out] = tail.near_r ? tail.near_r[r] : tail.near; v[n_out + 1] = tail.far_r ? tail.far_r[r] : tail.far; }
    for (int i = lane; i < n_extra; i += kDraw) v[n_out + 2 + i] = z[pk[i]];
    __syncthreads();
    for (int i = lane; i < n; i += kDraw) {  // rank sort (n ~ 100), k_sampler_final's
        const float x = v[i];
        int rank = 0;
        for (int j = 0; j < n; j++) rank += (v[j] < x) || (v[j] == x && j < i);
        sorted[rank] = x;
    }
    __syncthreads();
    for (int i = lane; i < n; i += kDraw) tail.z_out[(size_t)r * n + i] = sorted[i];
    if (lane == 0 && tail.z_eik) tail.z_eik[r] = sorted[tail.eik_u ? min((int)(tail.eik_u[r] * (float)n), n - 1) : (int)tail.eik_idx[r]];
}

// ------------------------------------------------------------------------------------ final
// z_out = sort(cat(z_samples, near, far, z[:, pick])); z_eik = z_out[eik_idx]
__global__ __launch_bounds__(kWave) void k_sampler_final(const float *__restrict__ z_samples, int n_s, const float *__restrict__ z_in, int ld,
                                                          const int64_t *__restrict__ pick, int n_extra, float near, float far,
                                                          const int64_t *__restrict__ eik_idx, float *__restrict__ z_out, float *__restrict__ z_eik,
                                                          int R, const float *__restrict__ near_r, const float *__restrict__ far_r, const float *__restrict__ eik_u) {
    extern __shared__ float lds[];
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= R) return;
    const int n = n_s + 2 + n_extra;
    float *v = lds, *sorted = lds + n;
    for (int i = lane; i < n_s; i += kWave) v[i] = z_samples[(size_t)r * n_s + i];
    if (lane == 0) { v[n_s] = near_r ? near_r[r] : near; v[n_s + 1] = far_r ? far_r[r] : far; }   // per-ray bounds: get_z_vals_near_far
    for (int i = lane; i < n_extra; i += kWave) v[n_s + 2 + i] = z_in[(size_t)r * ld + pick[i]];
    __syncthreads();
    for (int i = lane; i < n; i += kWave) {  // rank sort (n ~ 100)
        const float x = v[i];
        int rank = 0;
        for (int j = 0; j < n; j++) rank += (v[j] < x) || (v[j] == x && j < i);
        sorted[rank] = x;
    }
    __syncthreads();
    for (int i = lane; i < n; i += kWave) z_out[(size_t)r * n + i] = sorted[i];
    if (lane == 0 && z_eik) z_eik[r] = sorted[eik_u ? min((int)(eik_u[r] * (float)n), n - 1) : (int)eik_idx[r]];
}


// ------------------------------------------------------------------------------------ ray setup
// Camera rays + stratified uniform depths + initial beta in ONE launch (one wave per ray), replacing ~70 ATen launches:
// rend_util.get_camera_params x2 (utils/rend_util.py:56-125, incl. quirk Q1: the depth-scale rays see twice the pixel
// offset), UniformSampler.get_z_vals with the cube exit (model/ray_sampler.py:48-83), Lemma-2 beta (:136-140).
__global__ __launch_bounds__(kWave) void k_ray_setup(const float *__restrict__ uv, const float *__restrict__ offset, const float *__restrict__ pose,
                                                      const float *__restrict__ intr, const float *__restrict__ t_rand, int S, float near,
                                                      float far_cap, float bound, float eps, float *__restrict__ ray_dirs,
                                                      float *__restrict__ cam_loc, float *__restrict__ depth_scale, float *__restrict__ z0,
                                                      float *__restrict__ beta_init, int R, float divide_factor, float *__restrict__ x,
                                                      float *__restrict__ x01, float offset_shift, float *__restrict__ rot_out,
                                                      float *__restrict__ beta_work, const float *__restrict__ patch_u, int patch) {
    extern __shared__ float lds[];  // [S] stratified depths of this ray
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= R) return;
    if (rot_out && r == 0 && lane < 9) rot_out[lane] = pose[4 * (lane % 3) + lane / 3];   // world -> camera: transpose of the pose rotation
    const float fx = intr[0], sk = intr[1], cx = intr[2], fy = intr[5], cy = intr[6];
    const float ox = offset ? offset[2 * r] + offset_shift : 0.f, oy = offset ? offset[2 * r + 1] + offset_shift : 0.f;
    float u, v;
    if (patch_u) {      // ray r of the patch x patch pixel block whose origin the two draws place inside the image (network.py:919-925: randint over
                        // [0, W - patch] x [0, H - patch], W = 2 cx, H = 2 cy)
        u = (float)(r % patch) + floorf(patch_u[0] * (floorf(cx * 2.f) - (float)patch + 1.f));
        v = (float)(r / patch) + floorf(patch_u[1] * (floorf(cy * 2.f) - (float)patch + 1.f));
    } else {
        u = uv[2 * r];
        v = uv[2 * r + 1];
    }
    // lift (rend_util.py:112-125) at depth 1, then camera-to-world
    const float x1 = u + ox, y1 = v + oy;
    const float xl = (x1 - cx + cy * sk / fy - sk * y1 / fy) / fx, yl = (y1 - cy) / fy;
    float w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = pose[4 * i] * xl + pose[4 * i + 1] * yl + pose[4 * i + 2] + pose[4 * i + 3];
    const float o[3] = {pose[3], pose[7], pose[11]};
    float d[3];
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = w[i] / w[3] - o[i];
    const float dn = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] /= dn;
    // depth scale: z component of the normalised camera-frame ray, with 2x the offset in training (quirk Q1)
    const float x2 = u + 2.f * ox, y2 = v + 2.f * oy;
    const float xl2 = (x2 - cx + cy * sk / fy - sk * y2 / fy) / fx, yl2 = (y2 - cy) / fy;
    const float ds = 1.f / fmaxf(sqrtf(xl2 * xl2 + yl2 * yl2 + 1.f), 1e-12f);
    // exit from the cube [-bound, bound]^3 (ray_sampler.py:48-60), capped at far_cap
    float tn = -INFINITY, tf = INFINITY;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float a = (-bound - o[i]) / (d[i] + 1e-15f), b = (bound - o[i]) / (d[i] + 1e-15f);
        tn = fmaxf(tn, fminf(a, b));
        tf = fminf(tf, fmaxf(a, b));
    }
    if (tf < tn) tf = 1e9f;
    const float far = fminf(tf, far_cap);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) { ray_dirs[3 * r + i] = d[i]; cam_loc[3 * r + i] = o[i]; }
        depth_scale[r] = ds;
    }
    // stratified depths
    const float step = 1.0f / (float)(S - 1);
    for (int i = lane; i < S; i += kWave) {
        auto zlin = [&](int k) {
            const float t = (k < S / 2) ? step * (float)k : 1.0f - step * (float)(S - 1 - k);   // torch.linspace
            return near * (1.f - t) + far * t;
        };
        const float zi = zlin(i);
        float val = zi;
        if (t_rand) {
            const float lower = (i == 0) ? zi : 0.5f * (zi + zlin(i - 1));
            const float upper = (i == S - 1) ? zi : 0.5f * (zlin(i + 1) + zi);
            val = lower + (upper - lower) * t_rand[(size_t)r * S + i];
        }
        lds[i] = val;
        z0[(size_t)r * S + i] = val;
        if (x) {   // positions of the first sweep (k_ray_points' arithmetic, same roundings)
            const float inv_df = __fdiv_rn(1.0f, divide_factor);
            const size_t p = ((size_t)r * S + i) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float v = __fadd_rn(o[c], __fmul_rn(val, d[c]));
                x[p + c] = v;
                x01[p + c] = __fmul_rn(__fadd_rn(__fmul_rn(v, inv_df), 1.0f), 0.5f);
            }
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = lane; i + 1 < S; i += kWave) { const float dd = lds[i + 1] - lds[i]; acc += dd * dd; }
    acc = wave_sum(acc);
    if (lane == 0) {
        const float b = sqrtf((1.0f / (4.0f * logf(eps + 1.0f))) * acc);
        beta_init[r] = b;
        if (beta_work) beta_work[r] = b;      // the sampler's working copy (its update kernels overwrite it round by round)
    }
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

template <typename... Args>
void launch_draw(int R, int m_cap, int n_extra, hipStream_t st, Args... args) {
    // 128 threads per ray (one wave per ray leaves a SIMD with ONE wave walking 6-7 sections through libm exp / expm1: 15.9-17.6 us per launch at 64
    // threads, 11.9-12.4 at 128, 11.6-15.3 at 256 in the iteration, measured in round 3)
    const size_t lds = (4 * (size_t)m_cap + 3 * 4 + (size_t)n_extra) * sizeof(float);
    k_sampler_draw<128><<<dim3(R), dim3(128), lds, st>>>(args...);
}

}  // namespace

extern "C" {

static int sampler_update_launch(float *z, float *sdf, int32_t ld, int32_t m_old, const float *samples, const float *new_sdf, int32_t s_new,
                                 float *beta, const float *beta0, float eps, int32_t beta_iters, float *beta_max, int32_t R, const hsGate *gate,
                                 const int32_t *m_dev, const UpdDraw &dr, void *stream) {
    if (R <= 0) return HS_OK;
    if (!z || !sdf || !samples || !new_sdf || !beta || !beta0 || !beta_max) return HS_ERR_NULL;
    const int m = m_dev ? ld : m_old + s_new;   // device-side count: size the scratch for the row capacity
    if (m < 2 || m > ld || m > HS_SAMPLER_MAX_M) return HS_ERR_ARG;
    // 256 threads (four waves) per ray whatever the size of the merged set: 64 / 128 threads for the early rounds' 128 / 256 sections measured equal
    // (round 5: the kernel waits on its dependent chain, not on issue slots), 512 slower.  The line search holds a thread's <= 4 sections in registers
    // (error_bound_regs); larger merged sets fall back to the LDS form inside the kernel.
    const hsGate g = gate ? *gate : hsGate{nullptr, nullptr};
    const size_t lds = (6 * m + 3 * 8) * sizeof(float);
    k_sampler_update<256><<<dim3(R), dim3(256), lds, (hipStream_t)stream>>>(z, sdf, ld, m_old, samples, new_sdf, s_new, beta, beta0, eps, beta_iters, beta_max, R, g, m_dev, dr, true);
    return check_launch();
}

int hs_sampler_update(float *z, float *sdf, int32_t ld, int32_t m_old, const float *samples, const float *new_sdf, int32_t s_new,
                      float *beta, const float *beta0, float eps, int32_t beta_iters, float *beta_max, int32_t R, const hsGate *gate, const int32_t *m_dev,
                      void *stream) {
    return sampler_update_launch(z, sdf, ld, m_old, samples, new_sdf, s_new, beta, beta0, eps, beta_iters, beta_max, R, gate, m_dev, UpdDraw{}, stream);
}

int hs_sampler_update_draw(float *z, float *sdf, int32_t ld, int32_t m_old, const float *samples, const float *new_sdf, int32_t s_new,
                           float *beta, const float *beta0, float eps, int32_t beta_iters, float *beta_max, int32_t R, const hsGate *gate,
                           float add_tiny, int32_t n_out, float *out, const float *cam_loc, const float *ray_dirs, float divide_factor, float *x,
                           float *x01, void *stream) {
    if (!out || n_out <= 0) return HS_ERR_NULL;
    if (x && (!x01 || !cam_loc || !ray_dirs || divide_factor == 0.f)) return HS_ERR_NULL;
    UpdDraw dr{};
    dr.out = out; dr.n_out = n_out; dr.add_tiny = add_tiny;
    dr.ext.o = cam_loc; dr.ext.d = ray_dirs; dr.ext.x = x; dr.ext.x01 = x01; dr.ext.divide_factor = divide_factor;
    return sampler_update_launch(z, sdf, ld, m_old, samples, new_sdf, s_new, beta, beta0, eps, beta_iters, beta_max, R, gate, nullptr, dr, stream);
}

int hs_sampler_draw(const float *z, const float *sdf, int32_t ld, int32_t m, const float *beta, int32_t mode, float add_tiny, const float *u,
                    int32_t n_out, float *out, int32_t R, const hsGate *gate, const int32_t *m_dev, void *stream) {
    if (R <= 0 || n_out <= 0) return HS_OK;
    if (!z || !sdf || !beta || !out) return HS_ERR_NULL;
    if (m_dev) m = ld;
    if (m < 2 || m > ld || m > HS_SAMPLER_MAX_M || (mode != 0 && mode != 1)) return HS_ERR_ARG;
    launch_draw(R, m, 0, (hipStream_t)stream, z, sdf, ld, m, beta, mode, add_tiny, u, n_out, out, R, gate ? *gate : hsGate{nullptr, nullptr}, m_dev, DrawExt{}, TailExt{});
    return check_launch();
}

int hs_sampler_draw_step(const float *z, const float *sdf, int32_t ld, const float *beta, int32_t mode, float add_tiny, const float *u, int32_t n_out,
                         float *out, int32_t R, const hsSamplerCtl *ctl_in, hsSamplerCtl *ctl_out, const float *beta_max, const float *beta0,
                         int32_t s_new, int32_t max_rounds, const float *cam_loc, const float *ray_dirs, float divide_factor, float *x, float *x01,
                         void *stream) {
    return hs_sampler_draw_steps(z, sdf, ld, beta, mode, add_tiny, u, n_out, out, R, ctl_in, ctl_out, beta_max, beta0, s_new, max_rounds, 1, cam_loc,
                                 ray_dirs, divide_factor, x, x01, stream);
}

int hs_sampler_draw_steps(const float *z, const float *sdf, int32_t ld, const float *beta, int32_t mode, float add_tiny, const float *u, int32_t n_out,
                          float *out, int32_t R, const hsSamplerCtl *ctl_in, hsSamplerCtl *ctl_out, const float *beta_max, const float *beta0,
                          int32_t s_new, int32_t max_rounds, int32_t n_steps, const float *cam_loc, const float *ray_dirs, float divide_factor,
                          float *x, float *x01, void *stream) {
    if (n_steps < 1) return HS_ERR_ARG;
    if (R <= 0 || n_out <= 0) return HS_OK;
    if (!z || !sdf || !beta || !out || !ctl_in || !ctl_out || !beta_max || !beta0) return HS_ERR_NULL;
    if (ctl_in == ctl_out) return HS_ERR_ARG;
    if (x && (!x01 || !cam_loc || !ray_dirs || divide_factor == 0.f)) return HS_ERR_NULL;
    if (ld < 2 || ld > HS_SAMPLER_MAX_M || (mode != 0 && mode != 1)) return HS_ERR_ARG;
    const DrawExt ext{ctl_in, ctl_out, beta_max, beta0, s_new, max_rounds, n_steps, cam_loc, ray_dirs, x, x01, divide_factor};
    launch_draw(R, ld, 0, (hipStream_t)stream, z, sdf, ld, ld, beta, mode, add_tiny, u, n_out, out, R, hsGate{nullptr, nullptr}, (const int32_t *)nullptr, ext, TailExt{});
    return check_launch();
}

int hs_sampler_tail(const float *z, const float *sdf, int32_t ld, const float *beta, float add_tiny, const float *u, int32_t n_out, float *out, int32_t R,
                    const hsSamplerCtl *ctl_in, hsSamplerCtl *ctl_out, const float *beta_max, const float *beta0, int32_t s_new, int32_t max_rounds,
                    int32_t n_steps, const float *u_pick, const int64_t *pick_in, int32_t n_extra, float near, float far, const float *near_rays,
                    const float *far_rays, const int64_t *eik_idx, const float *eik_u, float *z_out, float *z_eik, void *stream) {
    if (n_steps < 1 || n_extra < 0) return HS_ERR_ARG;
    if (R <= 0 || n_out <= 0) return HS_OK;
    if (!z || !sdf || !beta || !out || !ctl_in || !ctl_out || !beta_max || !beta0 || !z_out || (z_eik && !eik_idx && !eik_u)) return HS_ERR_NULL;
    if (ctl_in == ctl_out) return HS_ERR_ARG;
    // the merged row has at least s_new entries; the tail reuses two m-float scratch arrays for its n = n_out + 2 + n_extra values
    if (ld < 2 || ld > HS_SAMPLER_MAX_M || n_out + 2 + n_extra > s_new || n_extra > s_new) return HS_ERR_ARG;
    const DrawExt ext{ctl_in, ctl_out, beta_max, beta0, s_new, max_rounds, n_steps, nullptr, nullptr, nullptr, nullptr, 1.f};
    const TailExt tail{u_pick, pick_in, n_extra, near, far, near_rays, far_rays, eik_idx, eik_u, z_out, z_eik};
    launch_draw(R, ld, n_extra, (hipStream_t)stream, z, sdf, ld, ld, beta, 1, add_tiny, u, n_out, out, R, hsGate{nullptr, nullptr}, (const int32_t *)nullptr, ext, tail);
    return check_launch();
}

int hs_sampler_step(hsSamplerCtl *ctl, const float *beta_max, const float *beta0, int32_t s_new, int32_t max_rounds, void *stream) {
    if (!ctl || !beta_max || !beta0) return HS_ERR_NULL;
    k_sampler_step<<<1, 1, 0, (hipStream_t)stream>>>(ctl, beta_max, beta0, s_new, max_rounds);
    return check_launch();
}

int hs_sampler_pick(const hsSamplerCtl *ctl, const float *u, int32_t n_extra, int64_t *pick, void *stream) {
    if (n_extra <= 0) return HS_OK;
    if (!ctl || !pick) return HS_ERR_NULL;
    k_sampler_pick<<<1, 256, 0, (hipStream_t)stream>>>(ctl, u, n_extra, pick);
    return check_launch();
}

int hs_sampler_final(const float *z_samples, int32_t n_s, const float *z, int32_t ld, const int64_t *pick, int32_t n_extra, float near, float far,
                     const int64_t *eik_idx, float *z_out, float *z_eik, int32_t R, const float *near_rays, const float *far_rays, const float *eik_u,
                     void *stream) {
    if (R <= 0) return HS_OK;
    if (!z_samples || !z || !z_out || (n_extra > 0 && !pick) || (z_eik && !eik_idx && !eik_u)) return HS_ERR_NULL;
    const int n = n_s + 2 + n_extra;
    if (n > 4096) return HS_ERR_ARG;
    k_sampler_final<<<dim3(R), dim3(kWave), 2 * n * sizeof(float), (hipStream_t)stream>>>(z_samples, n_s, z, ld, pick, n_extra, near, far, eik_idx,
                                                                                          z_out, z_eik, R, near_rays, far_rays, eik_u);
    return check_launch();
}

int hs_ray_setup(const float *uv, const float *ray_offset, const float *pose, const float *intrinsics, const float *t_rand, int32_t S, float near,
                 float far_cap, float bound, float eps, float *ray_dirs, float *cam_loc, float *depth_scale, float *z0, float *beta_init, int32_t R,
                 float divide_factor, float *x, float *x01, float offset_shift, float *rot_out, float *beta_work, const float *patch_u, int32_t patch,
                 void *stream) {
    if (R <= 0) return HS_OK;
    if (S < 2 || S > 4096) return HS_ERR_ARG;
    if ((!uv && !patch_u) || !pose || !intrinsics || !ray_dirs || !cam_loc || !depth_scale || !z0 || !beta_init) return HS_ERR_NULL;
    if (patch_u && (patch < 1 || patch * patch != R)) return HS_ERR_ARG;
    if (x && (!x01 || divide_factor == 0.f)) return HS_ERR_ARG;
    k_ray_setup<<<dim3(R), dim3(kWave), S * sizeof(float), (hipStream_t)stream>>>(uv, ray_offset, pose, intrinsics, t_rand, S, near, far_cap, bound, eps,
                                                                                 ray_dirs, cam_loc, depth_scale, z0, beta_init, R, divide_factor, x, x01, offset_shift, rot_out,
                                                                                 beta_work, patch_u, patch);
    return check_launch();
}

}  // extern "C"
