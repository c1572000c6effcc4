// holoscene_amd/csrc/iter_ops.hip -- the head and the tail of a training iteration, one launch each (gfx950).
//
// Before the first ray is set up an iteration needs: the weight-normalised matrices of its eight Linear layers (nn.utils.weight_norm,
// model/network.py:158-159 of the reference: W = g v / ||v||_row), the density's beta = |beta| + beta_min (model/density.py:28-30), its
// random draws (ray jitter network.py:785, stratified offsets ray_sampler.py:79, inverse-CDF draws :238, Eikonal points
// network.py:847-853), and the optimiser's step count / bias corrections / learning rates advanced by one (torch.optim.Adam +
// ExponentialLR, training/holoscene_train.py:156-169, 428).  Round 3 spent six launches on that (abs-shift, weight norm, two fills for
// PyTorch's Philox state + the generator kernel, the one-thread Adam tick) -- ~5 us each inside the replayed graph whatever they do.
// hs_iter_prologue does all of it in ONE launch: workgroups [0, W) normalise four rows each, workgroups [W, W + G) fill the pool of
// U[0, 1) draws (Philox-4x32-10 keyed by a device-resident (seed, counter) pair; the LAST generator workgroup to finish advances the
// counter, so every replay of a captured graph draws a new pool and no workgroup can see the new counter early), the last workgroup
// evaluates beta and ticks the optimiser state (nothing else in this launch reads that state).
// hs_iter_epilogue is the matching tail in front of the Adam sweep: the weight-norm backward of all layers + beta's backward (incl. the
// sum of the per-ray partial derivatives the compositing kernel leaves) in one launch, written wherever the caller points -- the flat
// gradient buffer's views, so that no copy launch follows.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"
#include "batch_draw.h"

namespace {

struct WnJobsI { hsWnJob j[HS_PACK_MAX_JOBS]; int32_t row_end[HS_PACK_MAX_JOBS]; int32_t n; };

__device__ __forceinline__ float wave_sum64i(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// one wave per output row (the arithmetic of appearance_mlp.hip: k_weight_norm, which stays the stand-alone entry point)
template <bool BWD>
__device__ __forceinline__ void wn_row(const WnJobsI &jobs, int row_g, int lane) {
    int j = 0;
    while (j < jobs.n && row_g >= jobs.row_end[j]) j++;
    if (j >= jobs.n) return;
    const hsWnJob jb = jobs.j[j];
    const int row = row_g - (j ? jobs.row_end[j - 1] : 0);
    const float *v = jb.v + (size_t)row * jb.cols;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < jb.cols; c += 64) {
        const float x = v[c];
        ss += x * x;
        if (BWD) dot += jb.gW[(size_t)row * jb.cols + c] * x;
    }
    const float norm = sqrtf(wave_sum64i(ss));
    const float g = jb.g[row];
    if (!BWD) {
        const float sc = g / norm;
        for (int c = lane; c < jb.cols; c += 64) jb.W[(size_t)row * jb.cols + c] = v[c] * sc;
    } else {
        dot = wave_sum64i(dot);
        const float sc = g / norm, back = dot / (norm * norm);
        for (int c = lane; c < jb.cols; c += 64) jb.gv[(size_t)row * jb.cols + c] = sc * (jb.gW[(size_t)row * jb.cols + c] - v[c] * back);
        if (lane == 0) jb.gg[row] = dot / norm;
    }
}

// Philox-4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> four 32-bit words
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x, hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }      // 24 bits: [0, 1)

constexpr int kRngPerThread = 16, kRngPerBlock = 256 * kRngPerThread;

constexpr int kZeroPerBlock = 256 * 16;      // floats a zeroing workgroup clears (four 16-byte stores per thread)
struct PrologueArgs {
    WnJobsI wn;
    int32_t wn_blocks, rng_blocks, zero_blocks;
    float *zero;                                // [n_zero] floats to clear (the small parameters' gradient storage + the accumulator pool), 16-byte aligned
    int64_t n_zero;
    float *pool;              // rng
    int64_t n_pool;
    uint64_t *rng_state;      // [0] seed, [1] counter, [2] workgroups done (0 between launches)
    const float *beta, *beta_min;
    float *beta_out;
    int32_t n_beta;
    hsAdamState *adam;        // tick (NULL: none)
    float beta1, beta2;
    double gamma;
    // the iteration's batch (batch_draw.h): draw_blocks workgroups in FRONT of the others -- its chain of dependent round trips is the longest
    // thing in this launch (22 us as a launch of its own); nothing else here reads what it writes
    int32_t draw_blocks, draw_uniform, draw_pixels, draw_total;
    int64_t *draw_out;
    hsDrawSched draw;
    DrawGatherJobs draw_jobs;
};

__device__ __forceinline__ void adam_tick_device(hsAdamState *st, float beta1, float beta2, double gamma) {       // optim.hip: k_adam_tick
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

__global__ __launch_bounds__(256) void k_iter_prologue(PrologueArgs a) {
    if ((int)blockIdx.x < a.draw_blocks) {
        draw_gather_sched_body((int)blockIdx.x, a.draw_blocks, a.draw, a.draw_uniform, a.draw_pixels, a.draw_total, a.draw_out, a.draw_jobs);
        return;
    }
    const int b = (int)blockIdx.x - a.draw_blocks;
    if (b < a.wn_blocks) {
        wn_row<false>(a.wn, b * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
        return;
    }
    if (b < a.wn_blocks + a.rng_blocks) {
        const uint64_t seed = a.rng_state[0], ctr = a.rng_state[1];
        const int64_t base = (int64_t)(b - a.wn_blocks) * kRngPerBlock;
        const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int q = 0; q < kRngPerThread / 4; q++) {
            const int64_t i = base + (int64_t)q * 1024 + threadIdx.x * 4;        // four consecutive draws per thread and pass: one 16-byte store
            if (i >= a.n_pool) break;
            const uint4 r = philox4x32_10(make_uint4((uint32_t)(i >> 2), (uint32_t)((uint64_t)i >> 34), (uint32_t)ctr, (uint32_t)(ctr >> 32)), key);
            const float f[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
            if (i + 3 < a.n_pool && ((uintptr_t)(a.pool + i) & 15) == 0) *reinterpret_cast<float4 *>(a.pool + i) = make_float4(f[0], f[1], f[2], f[3]);
            else
                for (int e = 0; e < 4 && i + e < a.n_pool; e++) a.pool[i + e] = f[e];
        }
        // every thread of this workgroup has read the counter; the last workgroup to get here advances it for the next launch
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long *>(a.rng_state + 2), 1ull);
            if (done == (unsigned long long)(a.rng_blocks - 1)) {
                a.rng_state[1] = ctr + 1;
                a.rng_state[2] = 0;
            }
        }
        return;
    }
    if (b < a.wn_blocks + a.rng_blocks + a.zero_blocks) {      // the iteration's memset rides along (it was a launch of its own: 5 us + a launch gap)
        const int64_t base = (int64_t)(b - a.wn_blocks - a.rng_blocks) * kZeroPerBlock;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t i = base + (int64_t)q * 1024 + threadIdx.x * 4;
            if (i + 3 < a.n_zero) *reinterpret_cast<float4 *>(a.zero + i) = make_float4(0.f, 0.f, 0.f, 0.f);
            else
                for (int e = 0; e < 4 && i + e < a.n_zero; e++) a.zero[i + e] = 0.f;
        }
        return;
    }
    // the last workgroup: beta and the optimiser tick
    for (int i = threadIdx.x; i < a.n_beta; i += 256) a.beta_out[i] = fabsf(a.beta[i]) + a.beta_min[0];
    if (a.adam && threadIdx.x == 0) adam_tick_device(a.adam, a.beta1, a.beta2, a.gamma);
}

constexpr int kMaxBetaParts = 4;
struct EpilogueArgs {
    WnJobsI wn;
    int32_t wn_blocks;
    const float *beta;
    const float *part[kMaxBetaParts];     // partial cotangents of |beta| + beta_min: arrays [len, n_beta], all summed here
    int32_t len[kMaxBetaParts];
    float *g_beta_out;
    int32_t n_beta, n_parts;
};

__global__ __launch_bounds__(256) void k_iter_epilogue(EpilogueArgs a) {
    const int b = blockIdx.x;
    if (b < a.wn_blocks) {
        wn_row<true>(a.wn, b * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
        return;
    }
    // the last workgroup: d |beta| = sgn(beta) (torch.sgn: 0 at 0) times the summed cotangent
    __shared__ float part[256];
    for (int i = 0; i < a.n_beta; i++) {
        float s = 0.f;
        for (int q = 0; q < a.n_parts; q++)
            for (int p = threadIdx.x; p < a.len[q]; p += 256) s += a.part[q][(size_t)p * a.n_beta + i];
        part[threadIdx.x] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float v = a.beta[i];
            a.g_beta_out[i] = part[0] * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
        }
        __syncthreads();
    }
}

int fill_wn(WnJobsI &wj, const hsWnJob *jobs, int32_t n_jobs, bool backward, int &total) {
    if (n_jobs < 0 || n_jobs > HS_PACK_MAX_JOBS) return HS_ERR_ARG;
    if (n_jobs > 0 && !jobs) return HS_ERR_NULL;
    total = 0;
    for (int i = 0; i < n_jobs; i++) {
        wj.j[i] = jobs[i];
        if (!jobs[i].v || !jobs[i].g || (backward ? (!jobs[i].gW || !jobs[i].gv || !jobs[i].gg) : !jobs[i].W)) return HS_ERR_NULL;
        if (jobs[i].rows < 1 || jobs[i].cols < 1) return HS_ERR_ARG;
        total += jobs[i].rows;
        wj.row_end[i] = total;
    }
    wj.n = n_jobs;
    return HS_OK;
}

}  // namespace

extern "C" {

int hs_iter_prologue_draw(const hsWnJob *jobs, int32_t n_jobs, float *rng_pool, int64_t n_rng, uint64_t *rng_state, const float *beta,
                          const float *beta_min, float *beta_out, int32_t n_beta, hsAdamState *adam, float beta1, float beta2, double gamma, float *zero,
                          int64_t n_zero, const hsDrawSched *draw, int32_t n_uniform, int32_t total_pixels, int32_t n_out, int64_t *draw_out,
                          const hsGatherJob *gather, int32_t n_gather, void *stream) {
    static_assert(kDrawThreads == 256, "the draw's workgroups ride in a 256-thread launch");
    PrologueArgs a;
    int rows = 0;
    const int rc = fill_wn(a.wn, jobs, n_jobs, false, rows);
    if (rc != HS_OK) return rc;
    if (n_rng < 0 || n_beta < 0 || n_zero < 0 || (n_zero > 0 && ((uintptr_t)zero & 15))) return HS_ERR_ARG;
    if (n_zero > 0 && !zero) return HS_ERR_NULL;
    a.zero = zero; a.n_zero = n_zero;
    a.zero_blocks = (int32_t)((n_zero + kZeroPerBlock - 1) / kZeroPerBlock);
    if ((n_rng > 0 && (!rng_pool || !rng_state)) || (n_beta > 0 && (!beta || !beta_min || !beta_out))) return HS_ERR_NULL;
    a.wn_blocks = (rows + 3) / 4;
    a.rng_blocks = (int32_t)((n_rng + kRngPerBlock - 1) / kRngPerBlock);
    a.pool = rng_pool; a.n_pool = n_rng; a.rng_state = rng_state;
    a.beta = beta; a.beta_min = beta_min; a.beta_out = beta_out; a.n_beta = n_beta;
    a.adam = adam; a.beta1 = beta1; a.beta2 = beta2; a.gamma = gamma;
    a.draw_blocks = 0;
    if (draw) {     /* hs_draw_gather_sched's arguments and checks */
        if (!draw->frames || !draw->sched || !draw->cursor || !draw_out) return HS_ERR_NULL;
        if (draw->n_sched < 1 || draw->n_frames < 1 || n_uniform < 0 || total_pixels < 1 || n_out < 0 || n_gather < 0 || n_gather > HS_GATHER_MAX_JOBS) return HS_ERR_ARG;
        if (n_gather > 0 && !gather) return HS_ERR_NULL;
        a.draw = *draw; a.draw_uniform = n_uniform; a.draw_pixels = total_pixels; a.draw_total = n_out; a.draw_out = draw_out;
        a.draw_jobs.n = n_gather;
        int64_t most = n_out;
        for (int i = 0; i < n_gather; i++) {
            const hsGatherJob &j = gather[i];
            if (j.n < 0 || j.row_bytes < 0 || (j.row_bytes & 3)) return HS_ERR_ARG;
            if (j.n > 0 && !j.dst) return HS_ERR_NULL;
            if (j.idx == draw_out && j.n != n_out) return HS_ERR_ARG;
            a.draw_jobs.j[i] = j;
            most = j.n > most ? j.n : most;
        }
        a.draw_blocks = (int32_t)((most + kDrawThreads - 1) / kDrawThreads);
    }
    const bool tail = n_beta > 0 || adam != nullptr;
    const int grid = a.draw_blocks + a.wn_blocks + a.rng_blocks + a.zero_blocks + (tail ? 1 : 0);
    if (grid == 0) return HS_OK;
    k_iter_prologue<<<grid, 256, 0, (hipStream_t)stream>>>(a);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

int hs_iter_prologue(const hsWnJob *jobs, int32_t n_jobs, float *rng_pool, int64_t n_rng, uint64_t *rng_state, const float *beta,
                     const float *beta_min, float *beta_out, int32_t n_beta, hsAdamState *adam, float beta1, float beta2, double gamma, float *zero,
                     int64_t n_zero, void *stream) {
    return hs_iter_prologue_draw(jobs, n_jobs, rng_pool, n_rng, rng_state, beta, beta_min, beta_out, n_beta, adam, beta1, beta2, gamma, zero, n_zero, nullptr, 0, 0,
                                 0, nullptr, nullptr, 0, stream);
}

int hs_iter_epilogue(const hsWnJob *jobs, int32_t n_jobs, const float *beta, const float *const *g_beta_parts, const int32_t *part_len,
                     int32_t n_parts, float *g_beta_out, int32_t n_beta, void *stream) {
    EpilogueArgs a;
    int rows = 0;
    const int rc = fill_wn(a.wn, jobs, n_jobs, true, rows);
    if (rc != HS_OK) return rc;
    if (n_beta < 0 || n_parts < 0 || n_parts > kMaxBetaParts) return HS_ERR_ARG;
    if (n_beta > 0 && (!beta || !g_beta_out)) return HS_ERR_NULL;
    if (n_parts > 0 && (!g_beta_parts || !part_len)) return HS_ERR_NULL;
    for (int q = 0; q < n_parts; q++) {
        if (part_len[q] < 0) return HS_ERR_ARG;
        if (part_len[q] > 0 && !g_beta_parts[q]) return HS_ERR_NULL;
        a.part[q] = g_beta_parts[q];
        a.len[q] = part_len[q];
    }
    a.wn_blocks = (rows + 3) / 4;
    a.beta = beta; a.g_beta_out = g_beta_out; a.n_beta = n_beta; a.n_parts = n_parts;
    const int grid = a.wn_blocks + (n_beta > 0 ? 1 : 0);
    if (grid == 0) return HS_OK;
    k_iter_epilogue<<<grid, 256, 0, (hipStream_t)stream>>>(a);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

}  // extern "C"
