// batch_draw.h -- device code of the per-iteration pixel draw (see batch_ops.hip for what it replaces and why a keyed permutation), shared by the
// stand-alone launches (batch_ops.hip) and the head-of-iteration launch that takes the scheduled draw along (iter_ops.hip: k_iter_prologue).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "holoscene_hip.h"

namespace {

constexpr int kDrawThreads = 256;
constexpr int kDrawLdsCls = 510;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {      // splitmix64 finaliser
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// P(i) for a keyed pseudo-random permutation P of [0, n), n >= 2, i < n
__device__ __forceinline__ uint32_t perm_index(uint32_t i, uint32_t n, uint64_t key) {
    int h = 1;
    while ((1u << (2 * h)) < n) h++;             // 4^h >= n: two h-bit halves
    const uint32_t mask = (1u << h) - 1u;
    uint32_t x = i;
    do {
        uint32_t l = x >> h, r = x & mask;
#pragma unroll
        for (int round = 0; round < 4; round++) {
            const uint32_t f = (uint32_t)(mix64(key + ((uint64_t)round << 56) + r) >> 32) & mask;
            const uint32_t t = l ^ f;
            l = r;
            r = t;
        }
        x = (l << h) | r;
    } while (x >= n);
    return x;
}

struct DrawArgs {
    const int32_t *class_ptr, *class_pix, *out_off;
    int32_t n_cls, per_class, n_bg, n_uniform, total_pixels;
    uint64_t seed, counter;
    int64_t *out;
};

// the pixel that output position t of the batch holds (t < out_off[n_cls + 1]).  Segment by bisection over out_off (a linear walk is
// n_cls dependent global loads: 17 us at 33 segments)
__device__ __forceinline__ int64_t drawn_pixel(const DrawArgs &a, int32_t t) {
    // the segment table through LDS when it fits (one load latency for the workgroup instead of log2(n_cls) dependent ones per thread)
    __shared__ int32_t s_off[kDrawLdsCls + 2];
    const bool staged = a.n_cls <= kDrawLdsCls;
    if (staged) {
        for (int i = threadIdx.x; i < a.n_cls + 2; i += kDrawThreads) s_off[i] = a.out_off[i];
        __syncthreads();
    }
    const int32_t *off = staged ? s_off : a.out_off;
    int lo = 0, hi = a.n_cls;           // the largest c in [0, n_cls] with out_off[c] <= t (empty segments in front of it are skipped)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= t) lo = mid; else hi = mid - 1;
    }
    const int c = lo;
    const bool uniform = c == a.n_cls;
    const int32_t first = uniform ? 0 : a.class_ptr[c];
    const int32_t n = uniform ? a.total_pixels : a.class_ptr[c + 1] - first;
    const int32_t quota = uniform ? a.n_uniform : (c == 0 ? a.n_bg : a.per_class);
    const int32_t i = t - off[c];
    int32_t pos = i;            // n <= quota: the whole class (ns_dataset.py:422-427); never taken by the uniform half of a real image
    if (n > quota) pos = (int32_t)perm_index((uint32_t)i, (uint32_t)n, mix64(a.seed ^ mix64(a.counter * 0x100000001b3ull + (uint64_t)c)));
    return uniform ? (int64_t)pos : (int64_t)a.class_pix[first + pos];
}

struct DrawGatherJobs { hsGatherJob j[HS_GATHER_MAX_JOBS]; int32_t n; };

// hs_draw_gather with batch number, frame and counter read from device memory (include/holoscene_hip.h: hsDrawSched): a node of the iteration's graph.
// No fences: workgroups exchange nothing but the two atomically accessed words -- each reads the batch number before it takes its ticket (its thread 0
// holds the value before the first barrier), and the one whose ticket is the last stores the next number after every other has read this one.
__device__ __forceinline__ void draw_gather_sched_body(int block, int nblocks, const hsDrawSched &s, int32_t n_uniform, int32_t total_pixels, int32_t total,
                                                       int64_t *out, const DrawGatherJobs &jobs) {
    __shared__ unsigned long long s_b;
    if (threadIdx.x == 0) s_b = __hip_atomic_load(reinterpret_cast<unsigned long long *>(s.cursor), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint64_t b = s_b;
    const int32_t f = s.sched[b % (uint64_t)s.n_sched];
    const hsFrameDesc *fd = s.frames + f;
    const DrawArgs a{fd->class_ptr, fd->class_pix, fd->out_off, fd->n_cls, fd->per_class, fd->n_bg, n_uniform, total_pixels, s.seed, s.counter_base + b, out};
    const int32_t t = block * kDrawThreads + threadIdx.x;
    const int64_t pix = drawn_pixel(a, t < total ? t : 0);
    if (t < total) out[t] = pix;
    uint32_t buf[HS_GATHER_MAX_JOBS][4];
#pragma unroll
    for (int q = 0; q < HS_GATHER_MAX_JOBS; q++) {
        if (q >= jobs.n) break;
        const hsGatherJob jb = jobs.j[q];
        const int words = jb.row_bytes >> 2;
        if (t >= jb.n || words > 4) continue;
        const int64_t r = jb.idx == out ? pix : (jb.idx ? jb.idx[t] : (int64_t)f);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(jb.src ? jb.src : fd->src[q]) + r * words;
#pragma unroll
        for (int w = 0; w < 4; w++) buf[q][w] = w < words ? src[w] : 0u;
    }
#pragma unroll
    for (int q = 0; q < HS_GATHER_MAX_JOBS; q++) {
        if (q >= jobs.n) break;
        const hsGatherJob jb = jobs.j[q];
        const int words = jb.row_bytes >> 2;
        if (t >= jb.n) continue;
        uint32_t *dst = reinterpret_cast<uint32_t *>(jb.dst) + (int64_t)t * words;
        if (words <= 4) {
#pragma unroll
            for (int w = 0; w < 4; w++)
                if (w < words) dst[w] = buf[q][w];
        } else {
            const int64_t r = jb.idx == out ? pix : (jb.idx ? jb.idx[t] : (int64_t)f);
            const uint32_t *src = reinterpret_cast<const uint32_t *>(jb.src ? jb.src : fd->src[q]) + r * words;
            for (int w = 0; w < words; w++) dst[w] = src[w];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int *ticket = reinterpret_cast<unsigned int *>(s.cursor + 1);
        if (atomicInc(ticket, (unsigned int)nblocks - 1u) == (unsigned int)nblocks - 1u)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(s.cursor), (unsigned long long)(b + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace
