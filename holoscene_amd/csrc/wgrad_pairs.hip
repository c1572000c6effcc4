// holoscene_amd/csrc/wgrad_pairs.hip -- weight gradients of the reverse-over-reverse trunk (trunk_rr.hip): C = sum over operand PAIRS of A^T B.
//
// Every weight matrix of the trunk receives two outer-product sums per sample in that formulation (dW1 = a1~^T h0 + v1^T u0~, likewise
// dW0 and dW2).  Same streaming split-M reduction as wgrad.hip -- workgroup = (product, row slice), rows through LDS in 64-row chunks, the
// whole result of the slice in registers, operands along the ROWS by gfx950's transposing LDS read -- with two differences:
//   * a job names up to two (A, B) pairs that are accumulated into the SAME registers (one partial per slice instead of two);
//   * an operand is either row-major [M, W] (W = 32 or 80: output cotangent, one-hot arg-min, network inputs) or TILE-PACKED
//     [M / 32][16][64] x 16 B, the layout the wave-tile kernels keep their 256-wide activations in (trunk_rr.hip) -- a 16-byte piece is
//     lane (row, half)'s two runs of four neurons of one k-step, dropped into the row-major LDS tile as two 8-byte writes.
// Result shapes: 256 x 256, 256 x 128 (B = [M, 80]: columns 80..127 of the LDS tile stay zero), 32 x 256.  Partials leave as bf16
// slices [S, NA, MB] like wgrad.hip's; hs_sum_slices adds them in fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "holoscene_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kThreadsP = 512, kWavesP = 8;
constexpr int RCP = 64;             // rows per chunk (two 32-row tiles of a tile-packed operand)
constexpr int PADP = 24;           // row pitch of the LDS tiles = width + 24 elements = 140 words = 12 mod 64 banks: the 4-row x 16-word transposed fragment reads collide two-way instead of four-way (+ 8: SQ_LDS_BANK_CONFLICT was 7 x SQ_ACTIVE_INST_LDS), the 8-byte tile-packed stores two-way (125 -> 120 us per launch; an XOR swizzle without padding, conflict-free on paper for both, measured 133 us: its address arithmetic costs more than the conflicts)

struct PairJobs { hsWgradPairJob j[HS_WGRAD_MAX_JOBS]; int32_t first[HS_WGRAD_MAX_JOBS + 1]; int32_t n; };

__device__ __forceinline__ uint32_t lds_addr_p(const uint16_t *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint16_t *)p; }

__device__ __forceinline__ bf16x8 tr_frag8p(uint32_t addr, uint32_t step) {
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

__device__ __forceinline__ uint16_t f2bf16p(float f) {
    const uint32_t u = __float_as_uint(f);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// one 64-row chunk of an operand of logical width W (LDS tile width WT >= W): 16 bytes per thread and pass
template <int W> struct ChunkP { uint4 v[(RCP * (W / 8) + kThreadsP - 1) / kThreadsP]; };

// TP: W / 16 k-steps per tile (256: the trunk's and the colour branch's activations, 128: the colour branch's assembled inputs).  Piece index
// within the chunk = (tile 0..1, k-step, lane): contiguous in memory, exactly like a row-major chunk
template <int W, bool TP>
__device__ __forceinline__ ChunkP<W> load_chunk_p(const uint16_t *__restrict__ src, int64_t row0, int64_t row_end) {      // rows >= row_end read as zero
    ChunkP<W> c;
    constexpr int SEG = W / 8, N = RCP * SEG;
#pragma unroll
    for (int i = 0; i < (N + kThreadsP - 1) / kThreadsP; i++) {
        const int idx = threadIdx.x + i * kThreadsP;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (TP) {
            const int row = 32 * (idx / (4 * W)) + (idx & 31);   // piece (tile, s, lane): 4 W pieces per tile, lane & 31 = row within the tile
            if (idx < N && row0 + row < row_end) v = *reinterpret_cast<const uint4 *>(src + (size_t)row0 * W + (size_t)idx * 8);
        } else {
            const int row = idx / SEG, seg = idx - row * SEG;
            if (idx < N && row0 + row < row_end) v = *reinterpret_cast<const uint4 *>(src + (size_t)(row0 + row) * W + seg * 8);
        }
        c.v[i] = v;
    }
    return c;
}

template <int W, int WT, bool TP>
__device__ __forceinline__ void store_chunk_p(uint16_t *lds, const ChunkP<W> &c) {
    constexpr int SEG = W / 8, N = RCP * SEG, P = WT + PADP;
#pragma unroll
    for (int i = 0; i < (N + kThreadsP - 1) / kThreadsP; i++) {
        const int idx = threadIdx.x + i * kThreadsP;
        if (idx >= N) continue;
        if constexpr (TP) {
            const int tile = idx / (4 * W), s = (idx >> 6) % (W / 16), lane = idx & 63, row = 32 * tile + (lane & 31), h = lane >> 5;
            uint16_t *dst = lds + (size_t)row * P + 16 * s + 4 * h;        // neurons 16 s + 4 h + 0..3 | 16 s + 8 + 4 h + 0..3
            *reinterpret_cast<uint2 *>(dst) = make_uint2(c.v[i].x, c.v[i].y);
            *reinterpret_cast<uint2 *>(dst + 8) = make_uint2(c.v[i].z, c.v[i].w);
        } else {
            const int row = idx / SEG, seg = idx - row * SEG;
            *reinterpret_cast<uint4 *>(lds + (size_t)row * P + seg * 8) = c.v[i];
        }
    }
}

// NA x MB result; A operands of width WA (= NA), B operands of width WB <= MB.  ATP / BTP: operand layout (both pairs of a job alike)
template <int NA, int MB, int WB, bool ATP, bool BTP>
__device__ __forceinline__ void pair_slice(const hsWgradPairJob &job, int slice, int S, uint16_t *lds) {
    constexpr int tiles_n = NA / 32, tiles_m = MB / 32;
    constexpr int WN = tiles_n < 4 ? tiles_n : 4, WM = kWavesP / WN;
    constexpr int TN = tiles_n / WN, TM = tiles_m / WM;
    static_assert(TN >= 1 && TM >= 1 && WN * TN == tiles_n && WM * TM == tiles_m, "unsupported result shape");
    constexpr int PA = NA + PADP, PB = MB + PADP;
    uint16_t *Abuf[2] = {lds, lds + RCP * PA + RCP * PB};
    uint16_t *Bbuf[2] = {lds + RCP * PA, lds + 2 * RCP * PA + RCP * PB};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % WN, wm = wave / WN;
    // slice = ceil(tiles / S) whole 32-row tiles; the last slices may be short or empty (their partial is then all zero): the caller cuts
    // every job of a launch in proportion to its BYTES, not to a divisor of its tile count
    const int64_t rows = ((job.M / 32 + S - 1) / S) * 32, row_begin = (int64_t)slice * rows;
    const int64_t row_end = row_begin + rows < job.M ? row_begin + rows : job.M;
    // row-major operands hold job.rows valid rows (<= M = whole 32-row tiles); tile-packed ones are zero-filled beyond them by their producers
    const int64_t end_a = ATP ? row_end : (row_end < job.rows ? row_end : job.rows), end_b = BTP ? row_end : (row_end < job.rows ? row_end : job.rows);
    if constexpr (WB < MB) {        // the padding columns of both B tiles: zero once; column WB becomes the ONES column below, the rest is never written again
        const int wz = job.B0 ? MB - WB : MB;           // (a job without B data: the whole tile)
        const int c0 = job.B0 ? WB : 0;
        for (int i = threadIdx.x; i < 2 * RCP * wz; i += kThreadsP) {
            const int buf = i / (RCP * wz), r = (i / wz) % RCP, cidx = i % wz;
            Bbuf[buf][(size_t)r * PB + c0 + cidx] = 0;
        }
    }
    // job.ones: during the FIRST pair column WB of the B tile holds 1 -> column WB of the result = column sums of that pair's A operand
    // (a bias gradient for free: the rows are streaming through anyway)
    auto ones_column = [&](uint16_t *Bt, bool first_pair) {
        if constexpr (WB < MB)
            if (job.ones && threadIdx.x < RCP) Bt[(size_t)threadIdx.x * PB + WB] = first_pair ? (uint16_t)0x3f80 : (uint16_t)0;
    };
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
    // job.colsum: column sums of the FIRST pair's A operand (a bias gradient) beside the product -- one more MFMA per A fragment against
    // an all-ones B fragment, on the waves of column group 0 (the rows are streaming through anyway; the kernel is bound by their bytes)
    const bool want_cs = job.colsum != nullptr && wm == 0;
    f32x16 accs[TN];
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int i = 0; i < 16; i++) accs[a][i] = 0.f;
    const int L16 = lane & 15, cg = (lane >> 4) & 1, rg = lane >> 5;
    const uint32_t a_off = (uint32_t)(((rg * 8 + (L16 >> 2)) * PA + wn * TN * 32 + 16 * cg + 4 * (L16 & 3)) * 2);
    const uint32_t b_off = (uint32_t)(((rg * 8 + (L16 >> 2)) * PB + wm * TM * 32 + 16 * cg + 4 * (L16 & 3)) * 2);
    auto multiply = [&](int cur, bool first_pair) {
        const uint32_t abase = lds_addr_p(Abuf[cur]) + a_off, bbase = lds_addr_p(Bbuf[cur]) + b_off;
#pragma unroll
        for (int ks = 0; ks < RCP / 16; ks++) {
            bf16x8 af[TN], bfr[TM];
#pragma unroll
            for (int a = 0; a < TN; a++) af[a] = tr_frag8p(abase + (uint32_t)((ks * 16 * PA + a * 32) * 2), 4 * PA * 2);
#pragma unroll
            for (int b = 0; b < TM; b++) bfr[b] = tr_frag8p(bbase + (uint32_t)((ks * 16 * PB + b * 32) * 2), 4 * PB * 2);
#pragma unroll
            for (int a = 0; a < TN; a++)
#pragma unroll
                for (int b = 0; b < TM; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
            if (want_cs && first_pair) {
                const uint32_t one2 = 0x3f803f80u;      // bf16 1.0 twice
                const uint32_t w4[4] = {one2, one2, one2, one2};
                const bf16x8 ones = *reinterpret_cast<const bf16x8 *>(w4);
#pragma unroll
                for (int a = 0; a < TN; a++) accs[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], ones, accs[a], 0, 0, 0);
            }
        }
    };
    // chunk sequence: all chunks of pair 0, then all chunks of pair 1 (when present) -- one pipeline, one chunk in flight ahead
    const int64_t nchunks = row_end > row_begin ? (row_end - row_begin + RCP - 1) / RCP : 0;
    const int npairs = job.A1 ? 2 : 1;
    const int64_t total = nchunks * npairs;
    auto operands = [&](int64_t c, const uint16_t *&A, const uint16_t *&B, int64_t &r0) {
        const bool second = c >= nchunks;
        A = reinterpret_cast<const uint16_t *>(second ? job.A1 : job.A0);
        B = reinterpret_cast<const uint16_t *>(second ? job.B1 : job.B0);
        r0 = row_begin + (second ? c - nchunks : c) * RCP;
    };
    const uint16_t *A, *B;
    int64_t r0;
    operands(0, A, B, r0);
    ChunkP<NA> ca = load_chunk_p<NA, ATP>(A, r0, end_a);
    ChunkP<WB> cb;
    if (B) cb = load_chunk_p<WB, BTP>(B, r0, end_b);
    __syncthreads();        // (the padding zeros above)
    store_chunk_p<NA, NA, ATP>(Abuf[0], ca);
    if (B) store_chunk_p<WB, MB, BTP>(Bbuf[0], cb);
    ones_column(Bbuf[0], true);
    __syncthreads();
    for (int64_t c = 0; c < total; c++) {
        const int cur = (int)(c & 1);
        if (c + 1 < total) {
            operands(c + 1, A, B, r0);
            ca = load_chunk_p<NA, ATP>(A, r0, end_a);
            if (B) cb = load_chunk_p<WB, BTP>(B, r0, end_b);
        }
        multiply(cur, c < nchunks);
        if (c + 1 < total) {
            store_chunk_p<NA, NA, ATP>(Abuf[cur ^ 1], ca);
            if (B) store_chunk_p<WB, MB, BTP>(Bbuf[cur ^ 1], cb);
            ones_column(Bbuf[cur ^ 1], c + 1 < nchunks);
        }
        __syncthreads();
    }
    if (want_cs && (lane & 31) == 0) {      // every column of the ones product holds the sum: lanes 0 and 32 write their 16 rows each
        float *cs = job.colsum + (size_t)slice * NA;
#pragma unroll
        for (int a = 0; a < TN; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) cs[(wn * TN + a) * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)] = accs[a][r];
    }
    uint16_t *dst = reinterpret_cast<uint16_t *>(job.part) + (size_t)slice * NA * MB;
#pragma unroll
    for (int a = 0; a < TN; a++)
#pragma unroll
        for (int b = 0; b < TM; b++) {
            const int m = (wm * TM + b) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int nn = (wn * TN + a) * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                dst[(size_t)nn * MB + m] = f2bf16p(acc[a][b][r]);
            }
        }
}

__global__ __launch_bounds__(kThreadsP) void k_wgrad_pairs(PairJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first[j + 1]) j++;
    const hsWgradPairJob &job = jobs.j[j];
    const int slice = blockIdx.x - jobs.first[j], S = job.slices;
    if (job.kind == HS_WGP_256x256) pair_slice<256, 256, 256, true, true>(job, slice, S, lds);            // a~^T h  + v^T u~   (all tile-packed)
    else if (job.kind == HS_WGP_256x80) pair_slice<256, 128, 80, true, false>(job, slice, S, lds);        // a0~^T xt + v0^T ux~ (B row-major [M, 80])
    else if (job.kind == HS_WGP_32x256) pair_slice<32, 256, 256, false, true>(job, slice, S, lds);        // y~^T h1 + onehot^T u1~ (A row-major [M, 32])
    else if (job.kind == HS_WGP_256x256_RM) pair_slice<256, 256, 256, false, false>(job, slice, S, lds);  // row-major operands (the value+Jacobian rows
    else if (job.kind == HS_WGP_256x80_RM) pair_slice<256, 128, 80, false, false>(job, slice, S, lds);    //  of the Eikonal points: gA1^T H0, gA0^T Xp)
    else if (job.kind == HS_WGP_256x128_RM) pair_slice<256, 128, 128, false, false>(job, slice, S, lds);  // the appearance branch's products (row-major
    else if (job.kind == HS_WGP_32x256_RM) pair_slice<32, 256, 256, false, false>(job, slice, S, lds);    //  activations): gA^T xin, gy^T r1
    else if (job.kind == HS_WGP_256x128_TP) pair_slice<256, 128, 128, true, true>(job, slice, S, lds);    // colour branch, wave-tile form: cotangent^T [features | encodings]
}

}  // namespace

extern "C" {

int hs_wgrad_pairs(const hsWgradPairJob *jobs, int32_t n_jobs, void *stream) {
    if (n_jobs < 0 || n_jobs > HS_WGRAD_MAX_JOBS) return HS_ERR_ARG;
    if (n_jobs == 0) return HS_OK;
    if (!jobs) return HS_ERR_NULL;
    PairJobs pj;
    pj.n = n_jobs;
    pj.first[0] = 0;
    for (int i = 0; i < n_jobs; i++) {
        const hsWgradPairJob &j = jobs[i];
        if (j.kind < HS_WGP_256x256 || j.kind > HS_WGP_256x128_TP) return HS_ERR_ARG;
        if (j.colsum && j.kind == HS_WGP_32x256) return HS_ERR_ARG;      // column sums: the 256-row results only
        // a slice is a whole number of 32-row tiles: the tile-packed operands are addressed by tile
        if (j.slices < 1 || j.M < 32 || (j.M % 32) != 0 || j.rows < 0 || j.rows > j.M) return HS_ERR_ARG;
        const bool no_b = !j.B0 && j.ones && (j.kind == HS_WGP_256x80 || j.kind == HS_WGP_256x80_RM) && !j.A1;     /* column sums only */
        if (!j.A0 || (!j.B0 && !no_b) || !j.part || (!j.A1) != (!j.B1)) return HS_ERR_NULL;
        pj.j[i] = j;
        pj.first[i + 1] = pj.first[i] + j.slices;
    }
    const size_t lds = 2 * (size_t)RCP * ((256 + PADP) + (256 + PADP)) * sizeof(uint16_t);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)k_wgrad_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    k_wgrad_pairs<<<pj.first[n_jobs], kThreadsP, lds, (hipStream_t)stream>>>(pj);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

}  // extern "C"
