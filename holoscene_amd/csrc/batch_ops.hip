// holoscene_amd/csrc/batch_ops.hip -- the per-iteration pixel draw of a training batch, on the device (gfx950).
//
// Reference: NSDataset.__getitem__ (datasets/ns_dataset.py:409-430), run in 8 DataLoader worker processes: half of the R rays split
// evenly over the instance classes present in the frame (class 0 takes the remainder; a class with fewer pixels than its quota gives all
// of them), each share = the first `want` entries of torch.randperm over the class's pixel list, i.e. a uniformly random subset; the
// other half = the first entries of a permutation of ALL pixels.  On the host that is ~4.5 ms per batch (one 262 144-element
// permutation plus one per class) -- twice a whole training iteration here.
//
// On the device "the first `want` entries of a random permutation" is taken literally: entry i of the share of class c is P_c(i), with
// P_c a KEYED PSEUDO-RANDOM PERMUTATION of [0, size_c) -- a four-round balanced Feistel network on the smallest even number of bits that
// holds size_c, round function splitmix64 keyed by (seed, counter, class), walked until it lands inside [0, size_c) (cycle walking: the
// restriction of a bijection of [0, 4^h) to the orbit entries below n is a bijection of [0, n); fewer than four steps on average).  One
// thread per output position, no shared state, no ordering step: distinctness is a property of the map, and a (seed, counter) pair names
// ONE batch whatever the scheduling.  (Round 3 drew by rejection into an LDS hash set and then sorted the accepted set by a hash to hide
// the insertion order: 45 barrier stages, 20.8 us for 1 024 pixels.  This form is one short wave-parallel pass, and because a thread now
// KNOWS its pixel the moment it has computed it, the row gather of the batch (hs_gather_rows) rides in the same launch: hs_draw_gather.)
// Same distribution family as the reference's rule (a uniformly random subset per class; the order inside a batch is immaterial to every
// consumer); the reference's own permutations remain injectable on the host path for the parity fixtures.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "holoscene_hip.h"
#include "batch_draw.h"

namespace {

__global__ __launch_bounds__(kDrawThreads) void k_draw_pixels(DrawArgs a, int32_t total) {
    const int32_t t = blockIdx.x * kDrawThreads + threadIdx.x;
    const int64_t pix = drawn_pixel(a, t < total ? t : 0);       // (every thread: the segment table is staged by the whole workgroup)
    if (t < total) a.out[t] = pix;
}

// the draw and the batch's row gather in one launch: dst_j[t, :] = src_j[pixel(t), :] for the jobs indexed by the drawn pixels (idx == out),
// dst_j[i, :] = src_j[idx_j[i], :] for the others (the frame's pose row: one row by a one-entry index).  Rows in 4-byte words.

__global__ __launch_bounds__(kDrawThreads) void k_draw_gather(DrawArgs a, int32_t total, DrawGatherJobs jobs) {
    const int32_t t = blockIdx.x * kDrawThreads + threadIdx.x;
    const int64_t pix = drawn_pixel(a, t < total ? t : 0);       // (every thread: the segment table is staged by the whole workgroup)
    if (t < total) a.out[t] = pix;
    // rows of up to four words (every per-pixel array of a batch): ALL loads first, then the stores -- source and destination may alias as far as
    // the compiler knows, so a load-store loop per job is one memory round trip after the other (17 us for six jobs)
    uint32_t buf[HS_GATHER_MAX_JOBS][4];
#pragma unroll
    for (int q = 0; q < HS_GATHER_MAX_JOBS; q++) {
        if (q >= jobs.n) break;
        const hsGatherJob jb = jobs.j[q];
        const int words = jb.row_bytes >> 2;
        if (t >= jb.n || words > 4) continue;
        const int64_t r = jb.idx == a.out ? pix : jb.idx[t];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(jb.src) + r * words;
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
        } else {        // a long row (the frame's pose: one row per batch)
            const int64_t r = jb.idx == a.out ? pix : jb.idx[t];
            const uint32_t *src = reinterpret_cast<const uint32_t *>(jb.src) + r * words;
            for (int w = 0; w < words; w++) dst[w] = src[w];
        }
    }
}

// the scheduled form as a launch of its own (batch_draw.h: draw_gather_sched_body; k_iter_prologue takes the same body along)
__global__ __launch_bounds__(kDrawThreads) void k_draw_gather_sched(hsDrawSched s, int32_t n_uniform, int32_t total_pixels, int32_t total, int64_t *out,
                                                                   DrawGatherJobs jobs) {
    draw_gather_sched_body((int)blockIdx.x, (int)gridDim.x, s, n_uniform, total_pixels, total, out, jobs);
}

int draw_args(DrawArgs &a, const int32_t *class_ptr, const int32_t *class_pix, const int32_t *out_off, int32_t n_cls, int32_t per_class, int32_t n_bg,
              int32_t n_uniform, int32_t total_pixels, uint64_t seed, uint64_t counter, int64_t *out) {
    if (n_cls < 1 || n_cls > 4096 || per_class < 0 || n_bg < 0 || n_uniform < 0 || total_pixels < 1) return HS_ERR_ARG;
    if (!class_ptr || !class_pix || !out_off || !out) return HS_ERR_NULL;
    a = DrawArgs{class_ptr, class_pix, out_off, n_cls, per_class, n_bg, n_uniform, total_pixels, seed, counter, out};
    return HS_OK;
}

}  // namespace

extern "C" {

int hs_draw_pixels(const int32_t *class_ptr, const int32_t *class_pix, const int32_t *out_off, int32_t n_cls, int32_t per_class, int32_t n_bg,
                   int32_t n_uniform, int32_t total_pixels, int32_t n_out, uint64_t seed, uint64_t counter, int64_t *out, void *stream) {
    DrawArgs a;
    const int rc = draw_args(a, class_ptr, class_pix, out_off, n_cls, per_class, n_bg, n_uniform, total_pixels, seed, counter, out);
    if (rc != HS_OK) return rc;
    if (n_out < 0) return HS_ERR_ARG;
    if (n_out == 0) return HS_OK;
    k_draw_pixels<<<(n_out + kDrawThreads - 1) / kDrawThreads, kDrawThreads, 0, (hipStream_t)stream>>>(a, n_out);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

int hs_draw_gather(const int32_t *class_ptr, const int32_t *class_pix, const int32_t *out_off, int32_t n_cls, int32_t per_class, int32_t n_bg,
                   int32_t n_uniform, int32_t total_pixels, int32_t n_out, uint64_t seed, uint64_t counter, int64_t *out, const hsGatherJob *jobs,
                   int32_t n_jobs, void *stream) {
    DrawArgs a;
    const int rc = draw_args(a, class_ptr, class_pix, out_off, n_cls, per_class, n_bg, n_uniform, total_pixels, seed, counter, out);
    if (rc != HS_OK) return rc;
    if (n_out < 0 || n_jobs < 0 || n_jobs > HS_GATHER_MAX_JOBS) return HS_ERR_ARG;
    if (n_jobs > 0 && !jobs) return HS_ERR_NULL;
    DrawGatherJobs gj;
    gj.n = n_jobs;
    int64_t most = n_out;
    for (int i = 0; i < n_jobs; i++) {
        const hsGatherJob &j = jobs[i];
        if (j.n < 0 || j.row_bytes < 0 || (j.row_bytes & 3)) return HS_ERR_ARG;
        if (j.n > 0 && (!j.src || !j.dst || !j.idx)) return HS_ERR_NULL;
        if (j.idx == out && j.n != n_out) return HS_ERR_ARG;        // a job on the drawn pixels gathers exactly the batch
        gj.j[i] = j;
        most = j.n > most ? j.n : most;
    }
    if (most == 0) return HS_OK;
    k_draw_gather<<<(unsigned)((most + kDrawThreads - 1) / kDrawThreads), kDrawThreads, 0, (hipStream_t)stream>>>(a, n_out, gj);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

int hs_draw_gather_sched(const hsDrawSched *sched, int32_t n_uniform, int32_t total_pixels, int32_t n_out, int64_t *out, const hsGatherJob *jobs,
                         int32_t n_jobs, void *stream) {
    if (!sched || !out) return HS_ERR_NULL;
    if (!sched->frames || !sched->sched || !sched->cursor) return HS_ERR_NULL;
    if (sched->n_sched < 1 || sched->n_frames < 1 || n_uniform < 0 || total_pixels < 1 || n_out < 0 || n_jobs < 0 || n_jobs > HS_GATHER_MAX_JOBS) return HS_ERR_ARG;
    if (n_jobs > 0 && !jobs) return HS_ERR_NULL;
    DrawGatherJobs gj;
    gj.n = n_jobs;
    int64_t most = n_out;
    for (int i = 0; i < n_jobs; i++) {
        const hsGatherJob &j = jobs[i];
        if (j.n < 0 || j.row_bytes < 0 || (j.row_bytes & 3)) return HS_ERR_ARG;
        if (j.n > 0 && !j.dst) return HS_ERR_NULL;           /* src NULL: per frame; idx NULL: the frame's own row */
        if (j.idx == out && j.n != n_out) return HS_ERR_ARG;
        gj.j[i] = j;
        most = j.n > most ? j.n : most;
    }
    if (most == 0) return HS_OK;
    k_draw_gather_sched<<<(unsigned)((most + kDrawThreads - 1) / kDrawThreads), kDrawThreads, 0, (hipStream_t)stream>>>(*sched, n_uniform, total_pixels, n_out, out, gj);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

}  // extern "C"
