// holoscene_amd/csrc/wgrad.hip -- weight gradients of the fused MLP kernels: C[n][m] = sum_rows A[row][n] * B[row][m] for row-major bf16
// activations / cotangents A [M, NA], B [M, MB] with M ~ 1e5..4e5 rows (gfx950).
//
// Reference: autograd's `grad_output.t() @ input` of every nn.Linear on the path (model/network.py:203-206, 586-612).  These products
// are streaming reductions over the rows: 0.1-0.4 GB of operands for a 256 x 256 result, bound by HBM.  The library's batched split-M
// GEMM (torch.bmm over 128 row slices) gives a 256 x 256 result ONE macro-tile per slice = 128 workgroups = half of the chip, and one
// launch per product.  Here one launch takes a table of products: workgroup -> (product, row slice), so 2+ products fill the 256 CUs
// together, every workgroup streams its rows through LDS once (64-row chunks, one or two in flight ahead of the one being multiplied)
// and keeps the whole result of its slice in registers (256 x 256 fp32 = 128 VGPRs per wave, 8 waves).  The operands run along the ROWS
// of row-major tiles, which is what gfx950's transposing LDS read delivers (`ds_read_b64_tr_b16`, sdf_mlp.hip: tr_frag).
// Partial results leave as bf16 slices [S, NA, MB] (the same rounding the library's bf16 bmm output had); hs_sum_slices adds them in fp32.
// Status (round 6): the benchmarked path forms its weight gradients in wgrad_pairs.hip; hs_wgrad_rows serves the workgroup-tile kernel family
// (appearance_mlp.hip, the d_out > 64 / non-stock fallbacks' row-major operands) and the tests that cross-check the two.
#include "launch_util.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "holoscene_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kThreadsG = 512, kWavesG = 8;
constexpr int RC = 64;              // rows per chunk
constexpr int PADG = 8;             // LDS row padding (elements): rows 4 banks apart, the 4-row transposing reads conflict-free

struct WgradJobs { hsWgradJob j[HS_WGRAD_MAX_JOBS]; };

__device__ __forceinline__ uint32_t lds_addr(const uint16_t *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint16_t *)p; }

// rows r0..r0+7 of the lane's column as one MFMA operand: two transposing reads 4 rows apart (sdf_mlp.hip: tr_frag)
__device__ __forceinline__ bf16x8 tr_frag8(uint32_t addr, uint32_t step) {
    uint2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\t"
                 "ds_read_b64_tr_b16 %1, %3\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "v"(addr + step)
                 : "memory");
    const uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
    return *reinterpret_cast<const bf16x8 *>(w);
}

__device__ __forceinline__ uint16_t f2bf16(float f) {      // round to nearest even (finite inputs)
    const uint32_t u = __float_as_uint(f);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// one 64-row chunk of an operand, 16 bytes per thread and pass; rows beyond the slice are zero
template <int W>
struct ChunkG { uint4 v[RC * (W / 8) / kThreadsG > 0 ? RC * (W / 8) / kThreadsG : 1]; };

template <int W>
__device__ __forceinline__ ChunkG<W> load_rows(const uint16_t *__restrict__ src, int64_t row0, int64_t row_end) {
    ChunkG<W> c;
    constexpr int SEG = W / 8, N = RC * SEG;
#pragma unroll
    for (int i = 0; i < (N + kThreadsG - 1) / kThreadsG; i++) {
        const int idx = threadIdx.x + i * kThreadsG, row = idx / SEG, seg = idx - row * SEG;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (idx < N && row0 + row < row_end) v = *reinterpret_cast<const uint4 *>(src + (size_t)(row0 + row) * W + seg * 8);
        c.v[i] = v;
    }
    return c;
}

template <int W>
__device__ __forceinline__ void store_rows(uint16_t *lds, const ChunkG<W> &c) {
    constexpr int SEG = W / 8, N = RC * SEG;
#pragma unroll
    for (int i = 0; i < (N + kThreadsG - 1) / kThreadsG; i++) {
        const int idx = threadIdx.x + i * kThreadsG, row = idx / SEG, seg = idx - row * SEG;
        if (idx < N) *reinterpret_cast<uint4 *>(lds + (size_t)row * (W + PADG) + seg * 8) = c.v[i];
    }
}

// NA x MB result; waves as WN x WM, each owning TN x TM tiles of 32 x 32
template <int NA, int MB>
__device__ __forceinline__ void wgrad_slice(const hsWgradJob &job, int slice, int S, uint16_t *lds) {
    constexpr int tiles_n = NA / 32, tiles_m = MB / 32;
    constexpr int WN = tiles_n < 4 ? tiles_n : 4, WM = kWavesG / WN;
    constexpr int TN = tiles_n / WN, TM = tiles_m / WM;
    static_assert(TN >= 1 && TM >= 1 && WN * TN == tiles_n && WM * TM == tiles_m, "unsupported result shape");
    constexpr int PA = NA + PADG, PB = MB + PADG;
    uint16_t *Abuf[2] = {lds, lds + RC * PA + RC * PB};
    uint16_t *Bbuf[2] = {lds + RC * PA, lds + 2 * RC * PA + RC * PB};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const uint16_t *A = reinterpret_cast<const uint16_t *>(job.A), *B = reinterpret_cast<const uint16_t *>(job.B);
    const int64_t rows = job.M / S, row_begin = (int64_t)slice * rows, row_end = row_begin + rows;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
    // transposing reads: in each 16-lane group lane L points at tile[r0 + (L >> 2)][c0 + 4 (L & 3)] and lane i receives rows r0..r0+3 of
    // column c0 + i; group g = lane >> 4 covers columns 16 (g & 1) and rows 8 (g >> 1) of the 32-column x 16-row operand block
    const int L16 = lane & 15, cg = (lane >> 4) & 1, rg = lane >> 5;
    const uint32_t a_off = (uint32_t)(((rg * 8 + (L16 >> 2)) * PA + wn * TN * 32 + 16 * cg + 4 * (L16 & 3)) * 2);
    const uint32_t b_off = (uint32_t)(((rg * 8 + (L16 >> 2)) * PB + wm * TM * 32 + 16 * cg + 4 * (L16 & 3)) * 2);
    const int64_t nchunks = (rows + RC - 1) / RC;
    auto multiply = [&](int cur) {
        const uint32_t abase = lds_addr(Abuf[cur]) + a_off, bbase = lds_addr(Bbuf[cur]) + b_off;
#pragma unroll
        for (int ks = 0; ks < RC / 16; ks++) {
            bf16x8 af[TN], bfr[TM];
#pragma unroll
            for (int a = 0; a < TN; a++) af[a] = tr_frag8(abase + (uint32_t)((ks * 16 * PA + a * 32) * 2), 4 * PA * 2);
#pragma unroll
            for (int b = 0; b < TM; b++) bfr[b] = tr_frag8(bbase + (uint32_t)((ks * 16 * PB + b * 32) * 2), 4 * PB * 2);
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    };
    // Where the accumulators leave room (results narrower than 256 x 256) chunks travel TWO rounds ahead of their multiplication (two
    // register sets, unrolled in pairs): with one 64 KB chunk in flight per workgroup the chip holds 16 MB of requests, just the
    // bandwidth-latency product of HBM under load (256 x 128: 31 -> 27 us, 32 x 256: 19 -> 16 us; 256 x 256 would spill 83 registers)
    constexpr bool DEEP = TN * TM <= 4;
    ChunkG<NA> ca0 = load_rows<NA>(A, row_begin, row_end), ca1;
    ChunkG<MB> cb0 = load_rows<MB>(B, row_begin, row_end), cb1;
    if constexpr (DEEP) {
        if (nchunks > 1) { ca1 = load_rows<NA>(A, row_begin + RC, row_end); cb1 = load_rows<MB>(B, row_begin + RC, row_end); }
    }
    store_rows<NA>(Abuf[0], ca0);
    store_rows<MB>(Bbuf[0], cb0);
    __syncthreads();
    if constexpr (DEEP) {
        for (int64_t c = 0; c < nchunks; c += 2) {
            if (c + 2 < nchunks) { ca0 = load_rows<NA>(A, row_begin + (c + 2) * RC, row_end); cb0 = load_rows<MB>(B, row_begin + (c + 2) * RC, row_end); }
            multiply(0);
            if (c + 1 < nchunks) { store_rows<NA>(Abuf[1], ca1); store_rows<MB>(Bbuf[1], cb1); }
            __syncthreads();
            if (c + 1 >= nchunks) break;
            if (c + 3 < nchunks) { ca1 = load_rows<NA>(A, row_begin + (c + 3) * RC, row_end); cb1 = load_rows<MB>(B, row_begin + (c + 3) * RC, row_end); }
            multiply(1);
            if (c + 2 < nchunks) { store_rows<NA>(Abuf[0], ca0); store_rows<MB>(Bbuf[0], cb0); }
            __syncthreads();
        }
    } else {
        for (int64_t c = 0; c < nchunks; c++) {
            const int cur = (int)(c & 1);
            if (c + 1 < nchunks) { ca0 = load_rows<NA>(A, row_begin + (c + 1) * RC, row_end); cb0 = load_rows<MB>(B, row_begin + (c + 1) * RC, row_end); }
            multiply(cur);
            if (c + 1 < nchunks) { store_rows<NA>(Abuf[cur ^ 1], ca0); store_rows<MB>(Bbuf[cur ^ 1], cb0); }
            __syncthreads();
        }
    }
    // D[n][m]: lane (m = lane & 31, h), register r <-> n = 8 (r >> 2) + 4 h + (r & 3) of the tile
    uint16_t *dst = reinterpret_cast<uint16_t *>(job.part) + (size_t)slice * NA * MB;
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int m = (wm * TM + b) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = (wn * TN + a) * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                dst[(size_t)n * MB + m] = f2bf16(acc[a][b][r]);
            }
        }
}

__global__ __launch_bounds__(kThreadsG) void k_wgrad_rows(WgradJobs jobs, int S) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int j = blockIdx.x / S, slice = blockIdx.x - j * S;
    const hsWgradJob &job = jobs.j[j];
    if (job.NA == 256 && job.MB == 256) wgrad_slice<256, 256>(job, slice, S, lds);
    else if (job.NA == 256 && job.MB == 128) wgrad_slice<256, 128>(job, slice, S, lds);
    else if (job.NA == 32 && job.MB == 256) wgrad_slice<32, 256>(job, slice, S, lds);
}

}  // namespace

extern "C" {

int hs_wgrad_rows(const hsWgradJob *jobs, int32_t n_jobs, int32_t slices, void *stream) {
    if (n_jobs < 0 || n_jobs > HS_WGRAD_MAX_JOBS || slices < 1) return HS_ERR_ARG;
    if (n_jobs == 0) return HS_OK;
    if (!jobs) return HS_ERR_NULL;
    WgradJobs wj;
    for (int i = 0; i < n_jobs; i++) {
        const hsWgradJob &j = jobs[i];
        const bool shape_ok = (j.NA == 256 && (j.MB == 256 || j.MB == 128)) || (j.NA == 32 && j.MB == 256);
        if (!shape_ok || j.M < slices || (j.M % slices) != 0) return HS_ERR_ARG;
        if (!j.A || !j.B || !j.part) return HS_ERR_NULL;
        wj.j[i] = j;
    }
    const size_t lds = 2 * (size_t)RC * ((256 + PADG) + (256 + PADG)) * sizeof(uint16_t);      // two chunks of the widest operand pair
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_wgrad_rows, (int)lds);
    k_wgrad_rows<<<n_jobs * slices, kThreadsG, lds, (hipStream_t)stream>>>(wj, slices);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

}  // extern "C"
