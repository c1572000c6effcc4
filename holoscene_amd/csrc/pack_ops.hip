// holoscene_amd/csrc/pack_ops.hip -- the three multi-job utility launches every path shares (moved out of appearance_mlp.hip in round 6, unchanged):
//   hs_pack_bf16    fp32 master matrices -> bf16 operand images (sub-blocks, zero padding, transposes), one launch for all of them
//   hs_sum_slices   sums of split-M partial stacks (weight-gradient slices), one launch for all of them
//   hs_weight_norm  nn.utils.weight_norm's W = g v / |v| forward and its backward for every layer of a model in one launch
#include "launch_util.h"
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"
#include "mfma_mlp.h"

namespace {

// ---------------------------------------------------------------------------------------------------------- weight packing
// fp32 master matrices -> the bf16 operand images the fused kernels read (sub-blocks, zero padding, transposes): one launch
// instead of ~20 slice / cast / transpose / pad kernels per iteration.
struct PackJobs { hsPackJob j[HS_PACK_MAX_JOBS]; };

__global__ __launch_bounds__(256) void k_pack_bf16(PackJobs jobs) {
    const hsPackJob jb = jobs.j[blockIdx.y];
    const int total = jb.dst_rows * jb.dst_cols;
    uint16_t *dst = reinterpret_cast<uint16_t *>(jb.dst);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / jb.dst_cols, c = i - r * jb.dst_cols;
        float v = 0.f;
        if (r < jb.rows && c < jb.cols)
            v = jb.scale * (jb.transpose ? jb.src[(size_t)(jb.row0 + c) * jb.ld + jb.col0 + r] : jb.src[(size_t)(jb.row0 + r) * jb.ld + jb.col0 + c]);
        dst[i] = (uint16_t)(pack_bf16(v, 0.f) & 0xffffu);
    }
}

// Sum over the S slices of the split-M weight-gradient GEMMs (bf16 [S, n] -> fp32 [n]) for up to HS_PACK_MAX_JOBS matrices in
// one launch (was one ATen reduce launch of ~11 us per matrix, 11 per iteration; the data is 17 MB each).
struct SumJobs { hsSumJob j[HS_PACK_MAX_JOBS]; };

// 256 threads = 32 element quads x 8 slice groups: group y adds slices y, y + 8, ... of its four elements, the eight partial sums meet in
// LDS.  (One thread per quad walking all 128 slices left a 256 x 256 result with 64 workgroups of 128 dependent loads each: 19 us.)
__global__ __launch_bounds__(256) void k_sum_slices(SumJobs jobs) {
    __shared__ float4 part[8][32];
    const hsSumJob jb = jobs.j[blockIdx.y];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint16_t *src = reinterpret_cast<const uint16_t *>(jb.src);
    const float *sf = reinterpret_cast<const float *>(jb.src);
    for (int64_t i0 = (int64_t)blockIdx.x * 128; i0 < jb.n; i0 += (int64_t)gridDim.x * 128) {
        const int64_t i = i0 + tx * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i + 3 < jb.n) {
            if (jb.src_f32) {
#pragma unroll 4
                for (int s_ = ty; s_ < jb.slices; s_ += 8) {
                    const float4 v = *reinterpret_cast<const float4 *>(sf + (size_t)s_ * jb.n + i);
                    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                }
            } else {
#pragma unroll 4
                for (int s_ = ty; s_ < jb.slices; s_ += 8) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(src + (size_t)s_ * jb.n + i);
                    a.x += __uint_as_float(v.x << 16); a.y += __uint_as_float(v.x & 0xffff0000u);
                    a.z += __uint_as_float(v.y << 16); a.w += __uint_as_float(v.y & 0xffff0000u);
                }
            }
        }
        part[ty][tx] = a;
        __syncthreads();
        if (ty == 0 && i < jb.n) {
            float4 r = part[0][tx];
#pragma unroll
            for (int y = 1; y < 8; y++) { const float4 v = part[y][tx]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
            *reinterpret_cast<float4 *>(jb.dst + i) = r;       // n % 4 == 0 (checked by hs_sum_slices)
        }
        __syncthreads();
    }
}

// Weight normalisation W = g * v / ||v||_row (nn.utils.weight_norm, dim 0; model/network.py:158-159) and its backward for
// several layers in one launch each: one wave per output row.
struct WnJobs { hsWnJob j[HS_PACK_MAX_JOBS]; int32_t row_end[HS_PACK_MAX_JOBS]; int32_t n; };

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <bool BWD>
__global__ __launch_bounds__(256) void k_weight_norm(WnJobs jobs) {
    const int row_g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
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
    const float norm = sqrtf(wave_sum64(ss));
    const float g = jb.g[row];
    if (!BWD) {
        const float sc = g / norm;
        for (int c = lane; c < jb.cols; c += 64) jb.W[(size_t)row * jb.cols + c] = v[c] * sc;
    } else {
        dot = wave_sum64(dot);
        const float sc = g / norm, back = dot / (norm * norm);
        for (int c = lane; c < jb.cols; c += 64) jb.gv[(size_t)row * jb.cols + c] = sc * (jb.gW[(size_t)row * jb.cols + c] - v[c] * back);
        if (lane == 0) jb.gg[row] = dot / norm;
    }
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

}  // namespace

extern "C" {

int hs_pack_bf16(const hsPackJob *jobs, int32_t n_jobs, void *stream) {
    if (n_jobs < 0 || n_jobs > HS_PACK_MAX_JOBS) return HS_ERR_ARG;
    if (n_jobs == 0) return HS_OK;
    if (!jobs) return HS_ERR_NULL;
    PackJobs pj;
    int max_total = 1;
    for (int i = 0; i < n_jobs; i++) {
        pj.j[i] = jobs[i];
        if (!jobs[i].src || !jobs[i].dst) return HS_ERR_NULL;
        if (jobs[i].rows > jobs[i].dst_rows || jobs[i].cols > jobs[i].dst_cols || jobs[i].dst_rows <= 0 || jobs[i].dst_cols <= 0) return HS_ERR_ARG;
        const int t = jobs[i].dst_rows * jobs[i].dst_cols;
        max_total = t > max_total ? t : max_total;
    }
    const int gx = (max_total + 255) / 256 < 64 ? (max_total + 255) / 256 : 64;
    k_pack_bf16<<<dim3(gx, n_jobs), 256, 0, (hipStream_t)stream>>>(pj);
    return check_launch();
}

int hs_sum_slices(const hsSumJob *jobs, int32_t n_jobs, void *stream) {
    if (n_jobs < 0 || n_jobs > HS_PACK_MAX_JOBS) return HS_ERR_ARG;
    if (n_jobs == 0) return HS_OK;
    if (!jobs) return HS_ERR_NULL;
    SumJobs sj;
    int64_t max_n = 4;
    for (int i = 0; i < n_jobs; i++) {
        sj.j[i] = jobs[i];
        if (!jobs[i].src || !jobs[i].dst) return HS_ERR_NULL;
        if (jobs[i].slices < 1 || jobs[i].n < 1 || (jobs[i].n & 3)) return HS_ERR_ARG;   // rows of 4 elements: 8-byte loads, 16-byte stores
        max_n = jobs[i].n > max_n ? jobs[i].n : max_n;
    }
    const int64_t want = (max_n + 127) / 128;
    k_sum_slices<<<dim3((unsigned)(want < 1024 ? want : 1024), n_jobs), 256, 0, (hipStream_t)stream>>>(sj);
    return check_launch();
}

int hs_weight_norm(const hsWnJob *jobs, int32_t n_jobs, int32_t backward, void *stream) {
    if (n_jobs < 0 || n_jobs > HS_PACK_MAX_JOBS) return HS_ERR_ARG;
    if (n_jobs == 0) return HS_OK;
    if (!jobs) return HS_ERR_NULL;
    WnJobs wj;
    int total = 0;
    for (int i = 0; i < n_jobs; i++) {
        wj.j[i] = jobs[i];
        if (!jobs[i].v || !jobs[i].g || (backward ? (!jobs[i].gW || !jobs[i].gv || !jobs[i].gg) : !jobs[i].W)) return HS_ERR_NULL;
        if (jobs[i].rows < 1 || jobs[i].cols < 1) return HS_ERR_ARG;
        total += jobs[i].rows;
        wj.row_end[i] = total;
    }
    wj.n = n_jobs;
    if (backward) k_weight_norm<true><<<(total + 3) / 4, 256, 0, (hipStream_t)stream>>>(wj);
    else k_weight_norm<false><<<(total + 3) / 4, 256, 0, (hipStream_t)stream>>>(wj);
    return check_launch();
}

}  // extern "C"
