// assemble.h -- hs_assemble's job table and workgroup body (see small_ops.hip for what it replaces), shared by its own launch (small_ops.hip: k_assemble)
// and by the table-scatter launches that take a backward stage's slice sums along in front of their own workgroups (hash_encode.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "holoscene_hip.h"

namespace {

struct AsmJobs { hsAsmJob j[HS_ASM_MAX_JOBS]; int32_t first[HS_ASM_MAX_JOBS + 1]; int32_t n; };      // (j[] in LAUNCH order: the long reductions first)

__device__ __forceinline__ float bf16_at(const void *p, int64_t i) { return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t *>(p)[i] << 16); }
__device__ __forceinline__ float term_at(const hsAsmTerm &tm, int64_t i) { return tm.src_bf16 ? bf16_at(tm.src, i) : tm.src[i]; }

// Two forms per job, chosen from its longest reduction:
//   * GROUPED (red < 256 everywhere: the split-M partial stacks of the weight-gradient kernels, 6-130 slices): a workgroup = 32 units x 8
//     slice groups; group y adds blocks y, y + 8, ... of its unit, the eight partial sums meet in LDS (appearance_mlp.hip: k_sum_slices'
//     scheme -- one thread walking 128 slices is 128 dependent-latency loads).  A unit is a QUAD of four consecutive destination columns
//     read with one 8- or 16-byte load per block when every term allows it (no column map, everything a multiple of four), else one element.
//   * WIDE (a term with red >= 256: column sums over per-workgroup partials): a whole WAVE per element -- lane l adds blocks l, l + 64, ...,
//     eight loads in flight, the lanes meet by shuffles (a plain loop is one dependent-latency load at a time: 22 us for 3 136 blocks).
__device__ __forceinline__ void assemble_body(const AsmJobs &jobs, int block) {
    __shared__ float4 part[8][32];
    // one-dimensional grid, job q owns workgroups [first[q], first[q + 1]): a (largest job) x (jobs) grid is mostly workgroups with nothing to
    // do, and dispatching 11 000 of them costs more than the sums (13.8 us for the colour branch's eleven jobs, 3 us for any one alone)
    int q = 0;
    while (q + 1 < jobs.n && block >= jobs.first[q + 1]) q++;
    const hsAsmJob &jb = jobs.j[q];
    const int bx = block - jobs.first[q], gx = jobs.first[q + 1] - jobs.first[q];
    const int64_t total = (int64_t)jb.rows * jb.cols;
    bool wide = false, quads = (jb.cols & 3) == 0;
    const bool dst_vec = (jb.dst_ld & 3) == 0 && (((uintptr_t)jb.dst) & 15) == 0;        // (a column window may start anywhere: scalar stores then)
    for (int t = 0; t < jb.n_terms; t++) {
        const hsAsmTerm &tm = jb.term[t];
        wide = wide || tm.red >= 256;
        quads = quads && !tm.col_map && (tm.col0 & 3) == 0 && (tm.ld & 3) == 0 && (tm.red_stride & 3) == 0 && (((uintptr_t)tm.src) & 15) == 0;
    }
    if (wide) {
        const int lane = threadIdx.x & 63;
        for (int64_t i = (int64_t)bx * 4 + (threadIdx.x >> 6); i < total; i += (int64_t)gx * 4) {
            const int r = (int)(i / jb.cols), c = (int)(i - (int64_t)r * jb.cols);
            float v = 0.f;
            for (int t = 0; t < jb.n_terms; t++) {
                const hsAsmTerm &tm = jb.term[t];
                const int64_t at = (int64_t)r * tm.ld + (tm.col_map ? tm.col_map[c] : tm.col0 + c);
                float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                int k = lane;
                for (; k + 7 * 64 < tm.red; k += 8 * 64) {
#pragma unroll
                    for (int u = 0; u < 8; u++) s8[u] += term_at(tm, at + (int64_t)(k + 64 * u) * tm.red_stride);
                }
                for (; k < tm.red; k += 64) s8[0] += term_at(tm, at + (int64_t)k * tm.red_stride);
                v += ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) jb.dst[(int64_t)r * jb.dst_ld + c] = v;
        }
        return;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t units = quads ? total >> 2 : total;
    for (int64_t u0 = (int64_t)bx * 32; u0 < units; u0 += (int64_t)gx * 32) {     // (workgroup-uniform bound: the barriers below are safe)
        const int64_t u = u0 + tx;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        int r = 0, c = 0;
        if (u < units) {
            const uint32_t e = (uint32_t)(quads ? u << 2 : u);        // (rows * cols < 2^31: 32-bit division -- the 64-bit one is ~100 instructions)
            r = (int)(e / (uint32_t)jb.cols);
            c = (int)(e - (uint32_t)r * (uint32_t)jb.cols);
            for (int t = 0; t < jb.n_terms; t++) {
                const hsAsmTerm &tm = jb.term[t];
                const int64_t at = (int64_t)r * tm.ld + (tm.col_map ? tm.col_map[c] : tm.col0 + c) + (int64_t)ty * tm.red_stride;
                const int64_t step = 8 * tm.red_stride;
                const int n_k = tm.red > ty ? (tm.red - ty + 7) >> 3 : 0;      // blocks ty, ty + 8, ...
                if (quads && tm.src_bf16) {
                    const uint16_t *p = reinterpret_cast<const uint16_t *>(tm.src) + at;
#pragma unroll 4
                    for (int k = 0; k < n_k; k++, p += step) {
                        const uint2 v = *reinterpret_cast<const uint2 *>(p);
                        a.x += __uint_as_float(v.x << 16); a.y += __uint_as_float(v.x & 0xffff0000u);
                        a.z += __uint_as_float(v.y << 16); a.w += __uint_as_float(v.y & 0xffff0000u);
                    }
                } else if (quads) {
                    const float *p = tm.src + at;
#pragma unroll 4
                    for (int k = 0; k < n_k; k++, p += step) {
                        const float4 v = *reinterpret_cast<const float4 *>(p);
                        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                    }
                } else if (tm.src_bf16) {
                    const uint16_t *p = reinterpret_cast<const uint16_t *>(tm.src) + at;
#pragma unroll 4
                    for (int k = 0; k < n_k; k++, p += step) a.x += __uint_as_float((uint32_t)*p << 16);
                } else {
                    const float *p = tm.src + at;
#pragma unroll 4
                    for (int k = 0; k < n_k; k++, p += step) a.x += *p;
                }
            }
        }
        part[ty][tx] = a;
        __syncthreads();
        if (ty == 0 && u < units) {
            float4 s4 = part[0][tx];
#pragma unroll
            for (int y = 1; y < 8; y++) { const float4 v = part[y][tx]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
            float *dst = jb.dst + (int64_t)r * jb.dst_ld + c;
            if (quads && dst_vec) *reinterpret_cast<float4 *>(dst) = s4;
            else if (quads) { dst[0] = s4.x; dst[1] = s4.y; dst[2] = s4.z; dst[3] = s4.w; }
            else *dst = s4.x;
        }
        __syncthreads();
    }
}

// the launch-order job table of `jobs` (the long reductions first) -> HS_OK and the number of workgroups in aj.first[aj.n]
inline int fill_asm_jobs(const hsAsmJob *jobs, int32_t n_jobs, AsmJobs &aj) {
    if (n_jobs < 0 || n_jobs > HS_ASM_MAX_JOBS) return HS_ERR_ARG;
    aj.n = n_jobs;
    aj.first[0] = 0;
    if (n_jobs == 0) return HS_OK;
    if (!jobs) return HS_ERR_NULL;
    int placed = 0;
    // two passes: the wide jobs (one workgroup walking hundreds of blocks: the longest single chain of the launch) get the first workgroups,
    // not the last ones behind two rounds of the others (+2.6 us measured)
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < n_jobs; i++) {
            const hsAsmJob &j = jobs[i];
            if (pass == 0) {
                if (j.rows < 1 || j.cols < 1 || j.n_terms < 1 || j.n_terms > HS_ASM_MAX_TERMS || j.dst_ld < j.cols) return HS_ERR_ARG;
                if (!j.dst) return HS_ERR_NULL;
            }
            bool wide = false, quads = (j.cols & 3) == 0;
            for (int t = 0; t < j.n_terms; t++) {
                const hsAsmTerm &tm = j.term[t];
                if (!tm.src) return HS_ERR_NULL;
                if (tm.red < 1) return HS_ERR_ARG;
                wide = wide || tm.red >= 256;
                quads = quads && !tm.col_map && (tm.col0 & 3) == 0 && (tm.ld & 3) == 0 && (tm.red_stride & 3) == 0 && (((uintptr_t)tm.src) & 15) == 0;
            }
            if (wide != (pass == 0)) continue;
            aj.j[placed] = j;
            const int64_t n = (int64_t)j.rows * j.cols;
            // a wave per element (wide) / 32 units per workgroup and pass, a unit = four elements where the kernel can take quads; at most 640
            // workgroups per job
            int64_t blocks = wide ? (n + 3) / 4 : ((quads ? n / 4 : n) + 31) / 32;
            blocks = blocks < 1 ? 1 : (blocks > 640 ? 640 : blocks);
            aj.first[placed + 1] = aj.first[placed] + (int32_t)blocks;
            placed++;
        }
    return HS_OK;
}

}  // namespace
