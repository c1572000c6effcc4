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
#include "launch_util.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "holoscene_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kThreadsP = 512, kWavesP = 8;
constexpr int RCP = 64;             // rows per chunk (two 32-row tiles of a tile-packed operand)
constexpr int PADP = 24;           // row pitch of the LDS tiles = width + 24 elements = 140 words = 12 mod 64 banks: the 4-row x 16-word transposed fragment reads collide two-way instead of four-way (+ 8: SQ_LDS_BANK_CONFLICT was 7 x SQ_ACTIVE_INST_LDS), the 8-byte tile-packed stores two-way (125 -> 120 us per launch; an XOR swizzle without padding, conflict-free on paper for both, measured 133 us: its address arithmetic costs more than the conflicts)

struct PairJobs { hsWgradPairJob j[HS_WGRAD_MAX_JOBS]; int32_t first[HS_WGRAD_MAX_JOBS + 1]; int32_t n; int32_t dma; };

__device__ __forceinline__ uint32_t lds_addr_p(const uint16_t *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint16_t *)p; }

// gfx950's transposing LDS read through the compiler's builtin (not inline asm): the wait for the result is placed by the compiler at the
// first use, so the twelve reads of a k-step are in flight together and the next k-step's are issued under this one's MFMAs
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_frag8p(uint32_t addr, uint32_t step) {
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4 *)(uintptr_t)addr);
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4 *)(uintptr_t)(addr + step));
    const i16x8 w = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, w);
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

// the slice's partial (bf16 [NA, MB]) and, when asked for, the column sums of the first pair's A operand (fp32 [NA])
template <int NA, int MB, int TN, int TM>
__device__ __forceinline__ void write_result(const hsWgradPairJob &job, int slice, const f32x16 (&acc)[TN][TM], const f32x16 (&accs)[TN], bool want_cs, int wn,
                                             int wm, int lane) {
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
    write_result<NA, MB, TN, TM>(job, slice, acc, accs, want_cs, wn, wm, lane);
}

// ---------------------------------------------------------------------------------------------------------------- LDS-DMA form
// The same reduction with the rows brought in by LDS-DMA (global_load_lds) instead of through registers: a stage = ONE 32-row tile of both
// operands, as many stages as 144 KB of LDS hold (4 for two 256-wide tile-packed operands, up to 7 for the narrow kinds), all but one in
// flight -- ~100 KB per CU on their way at any time instead of one 64 KB chunk for part of the cycle
// (the register form measured 3.8 TB/s: one chunk period ~ one loaded-HBM round trip).  No staging registers, no ds_write pass.
//   * a tile-packed operand lands AS IT IS, one 1 KB k-step block per DMA instruction, and the transposing fragment reads address it in
//     place.  Two choices make those reads conflict-free: within a block the 16-byte granules are permuted (free: each lane of the DMA
//     names its own source) so that rows r..r+3 of both lane halves are 128 contiguous bytes, and the blocks sit at a pitch of
//     1 KB + 128 B so that the two k-steps of a 32-column tile fall on the two bank halves;
//   * a row-major operand ([M, 80] / [M, 32]) lands linearly (pitch 160 / 64 B); rows beyond job.rows are read as copies of the last
//     valid row: their partner in the pair is tile-packed, whose rows there are zero (hs_wgrad_pairs' contract for these kinds).
// The vector-memory counter retires in order and the loop issues nothing else, so a wave waits for "all but the younger stages' requests".
// (round 5: THREE stages' worth of LDS for the widest kind instead of the four that fit -- and 4 / 6 instead of 5 / 8 stages for the narrower kinds:
// 113 -> 106 us per launch under rocprofv3 on three boxes, the bench's median -7 us in five alternating pairs.  Fewer requests in flight per
// compute unit (2 x 36 KB instead of 3 x 36 KB) serve the same bytes sooner: the queue in front of the memory is what a stage waits in.
// -DHS_WGP_STAGES=4 restores round 4's depth; all three stages of the narrow kinds alone, or five, measured slower than either.)
#ifndef HS_WGP_STAGES
#define HS_WGP_STAGES 3
#endif
constexpr int kLdsD = HS_WGP_STAGES * 2 * 16 * (1024 + 128), kBlkD = 1024 + 128;      // 144 KB of stages; their number follows from the kind's stage size
template <int W, bool TP> struct OperandD {
    static constexpr int pieces = TP ? W / 16 : (64 * W + 1023) / 1024;       // DMA instructions (1 KB each) per stage
    static constexpr int bytes = TP ? pieces * kBlkD : pieces * 1024;         // LDS bytes per stage
};

template <int W, bool TP>
__device__ __forceinline__ void request_piece(const char *__restrict__ src, int p, int64_t tile, int64_t rows_valid, char *lds_op, int lane) {
    const char *from;
    char *to;
    if constexpr (TP) {
        const int srcl = ((lane >> 2) & 1) * 32 + (lane >> 3) * 4 + (lane & 3);       // granule g of the block <- piece lane (h = g[2], r = 4 g[5:3] + g[1:0])
        from = src + ((size_t)tile * (W / 16) + p) * 1024 + srcl * 16;
        to = lds_op + p * kBlkD;
    } else {
        const int b = p * 1024 + lane * 16, row = b / (2 * W);
        int64_t rg = tile * 32 + row;
        rg = rg < rows_valid ? rg : rows_valid - 1;
        from = src + (size_t)rg * (2 * W) + (b - row * 2 * W);
        to = lds_op + p * 1024;
    }
    // The request is inline asm ON PURPOSE.  Through the builtin the compiler knows that LDS is being written behind its back and, having no
    // alias information for LDS, makes the next LDS read of ANY address wait for EVERY request in flight (s_waitcnt vmcnt(0) in front of the
    // first fragment read of a stage): the stages would land one after the other with nothing overlapped.  This way the landing is tracked
    // by wait_vm() alone; the compiler's own vector-memory waits only become more conservative (the counter retires in issue order).
    const uint32_t to_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)to);
    // `nt`: the operands are at their LAST read (1 GB per iteration that would otherwise sweep the hash tables and the weight images out of the
    // caches: bench median -15 us in four alternating pairs; -DHS_WGP_NT_LOADS=0 for A/B)
#ifndef HS_WGP_NT_LOADS
#define HS_WGP_NT_LOADS 1
#endif
#if HS_WGP_NT_LOADS
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(to_lds), "v"(from) : "memory");
#else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(to_lds), "v"(from) : "memory");
#endif
}

#define HS_WAIT_VM_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm(int n) {      // s_waitcnt takes an immediate
    switch (n) {
        HS_WAIT_VM_CASE(0) HS_WAIT_VM_CASE(1) HS_WAIT_VM_CASE(2) HS_WAIT_VM_CASE(3) HS_WAIT_VM_CASE(4) HS_WAIT_VM_CASE(5) HS_WAIT_VM_CASE(6) HS_WAIT_VM_CASE(7)
        HS_WAIT_VM_CASE(8) HS_WAIT_VM_CASE(9) HS_WAIT_VM_CASE(10) HS_WAIT_VM_CASE(11) HS_WAIT_VM_CASE(12) HS_WAIT_VM_CASE(13) HS_WAIT_VM_CASE(14)
        HS_WAIT_VM_CASE(15) HS_WAIT_VM_CASE(16) HS_WAIT_VM_CASE(17) HS_WAIT_VM_CASE(18) HS_WAIT_VM_CASE(19) HS_WAIT_VM_CASE(20) HS_WAIT_VM_CASE(21)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // (waiting for more than asked is always correct)
    }
}

template <int NA, int MB, int WB, bool ATP, bool BTP>
__device__ __forceinline__ void pair_slice_dma(const hsWgradPairJob &job, int slice, int S, char *lds) {
    constexpr int tiles_n = NA / 32, tiles_m = MB / 32;
    constexpr int WN = tiles_n < 4 ? tiles_n : 4, WM = kWavesP / WN;
    constexpr int TN = tiles_n / WN, TM = tiles_m / WM;
    static_assert(TN >= 1 && TM >= 1 && WN * TN == tiles_n && WM * TM == tiles_m, "unsupported result shape");
    using OA = OperandD<NA, ATP>;
    using OB = OperandD<WB, BTP>;
    constexpr int SA = OA::bytes, SB = (OB::bytes + 1023) / 1024 * 1024 + (BTP ? 0 : 1024), STG = SA + SB;     // (row-major B: fragment reads of the padding columns stay inside)
    constexpr int P = OA::pieces + OB::pieces;
    // as many stages as fit, at most 8: the narrow kinds (18-27 KB per stage) need more of them in flight for the same bytes on their way
    constexpr int NS = kLdsD / STG < 8 ? kLdsD / STG : 8;
    static_assert(NS >= 3 && (NS - 2) * ((P + kWavesP - 1) / kWavesP) <= 21, "stage count outside wait_vm's range");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave % WN, wm = wave / WN;
    const int mine = (P - wave + kWavesP - 1) / kWavesP;        // DMA instructions of this wave per stage
    // which tiles a slice sums: by default tiles slice, slice + S, slice + 2 S, ... -- the workgroups of a job then read ONE moving window of
    // S consecutive tiles per operand (DRAM pages and the memory-side cache see a few sequential streams instead of S scattered ones);
    // reserved bit 1: ceil(tiles / S) consecutive tiles like the register form (the tests compare the two forms bit for bit that way)
    const int64_t T = job.M / 32;
    const bool strided = !(job.reserved & 2);
    const int64_t tiles_per = (T + S - 1) / S, tile_begin = strided ? slice : (int64_t)slice * tiles_per, tile_step = strided ? S : 1;
    const int64_t ntile = strided ? (slice < T ? (T - slice + S - 1) / S : 0)
                                  : (tile_begin < T ? (tile_begin + tiles_per < T ? tiles_per : T - tile_begin) : 0);
    const int npairs = job.A1 ? 2 : 1;
    const int64_t total = ntile * npairs;
    auto request = [&](int64_t c) {
        const bool second = c >= ntile;
        const char *A = reinterpret_cast<const char *>(second ? job.A1 : job.A0), *B = reinterpret_cast<const char *>(second ? job.B1 : job.B0);
        const int64_t tile = tile_begin + (second ? c - ntile : c) * tile_step;
        char *st = lds + (int)(c % NS) * STG;
#pragma unroll
        for (int i = 0; i < (P + kWavesP - 1) / kWavesP; i++) {
            const int q = wave + kWavesP * i;
            if (q < OA::pieces) request_piece<NA, ATP>(A, q, tile, job.rows, st, lane);
            else if (q < P) request_piece<WB, BTP>(B, q - OA::pieces, tile, job.rows, st + SA, lane);
        }
    };
    f32x16 acc[TN][TM], accs[TN];
#pragma unroll
    for (int a = 0; a < TN; a++) {
#pragma unroll
        for (int i = 0; i < 16; i++) accs[a][i] = 0.f;
#pragma unroll
        for (int b = 0; b < TM; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
    }
    const bool want_cs = job.colsum != nullptr && wm == 0;
    const int L16 = lane & 15, cg = (lane >> 4) & 1, rg = lane >> 5, run = L16 & 3, r4 = L16 >> 2;
    // lane part of a fragment address + this wave's first tile; per (tile, k-step of 16 rows) an immediate on top
    constexpr int a_tile = ATP ? 2 * kBlkD : 64, a_ks = ATP ? 512 : 16 * 2 * NA, a_step = ATP ? 128 : 8 * NA;
    constexpr int b_tile = BTP ? 2 * kBlkD : 64, b_ks = BTP ? 512 : 16 * 2 * WB, b_step = BTP ? 128 : 8 * WB;
    const uint32_t a_lane = (ATP ? cg * kBlkD + rg * 256 + (run & 1) * 64 + r4 * 16 + (run >> 1) * 8 : (8 * rg + r4) * 2 * NA + (16 * cg + 4 * run) * 2) + wn * TN * a_tile;
    const uint32_t b_lane = (BTP ? cg * kBlkD + rg * 256 + (run & 1) * 64 + r4 * 16 + (run >> 1) * 8 : (8 * rg + r4) * 2 * WB + (16 * cg + 4 * run) * 2) + wm * TM * b_tile;
    auto multiply = [&](const char *st, bool first_pair) {
        const uint32_t abase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)st + a_lane, bbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)(st + SA) + b_lane;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 af[TN], bfr[TM];
#pragma unroll
            for (int a = 0; a < TN; a++) af[a] = tr_frag8p(abase + (uint32_t)(a * a_tile + ks * a_ks), a_step);
#pragma unroll
            for (int b = 0; b < TM; b++) {
                bfr[b] = tr_frag8p(bbase + (uint32_t)(b * b_tile + ks * b_ks), b_step);
                if constexpr (WB < MB) {        // columns >= WB of the result tile: zero operand
                    const bool valid = (wm * TM + b) * 32 + 16 * cg < WB;
                    const uint32_t z[4] = {0u, 0u, 0u, 0u};
                    if (!valid) bfr[b] = *reinterpret_cast<const bf16x8 *>(z);
                }
            }
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
    for (int64_t c = 0; c < NS - 1 && c < total; c++) request(c);
    for (int64_t c = 0; c < total; c++) {
        const int64_t younger = total - 1 - c < NS - 2 ? total - 1 - c : NS - 2;      // stages requested after c and still allowed in flight
        wait_vm((int)younger * mine);
        // (the bare barrier instruction: __syncthreads() carries a workgroup fence, for which the compiler drains the vector-memory counter --
        // every stage in flight -- before the barrier.  Nothing here needs that fence: LDS is written by the DMA only, whose landing the
        // counted wait above covers, and the fragment reads of stage c - 1 were consumed by its MFMAs.)
        __builtin_amdgcn_s_barrier();         // stage c has landed for every wave; every wave is done with stage c - 1, whose buffer the next request reuses
        if (c + NS - 1 < total) request(c + NS - 1);
        multiply(lds + (int)(c % NS) * STG, c < ntile);
    }
    write_result<NA, MB, TN, TM>(job, slice, acc, accs, want_cs, wn, wm, lane);
}

__global__ __launch_bounds__(kThreadsP) void k_wgrad_pairs(PairJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.first[j + 1]) j++;
    const hsWgradPairJob &job = jobs.j[j];
    const int slice = blockIdx.x - jobs.first[j], S = job.slices;
    if (jobs.dma && !job.ones && !(job.reserved & 1) && job.rows > 0) {      // (rows == 0: a row-major operand has no row to clamp to)       // the all-tile-packed and the mixed kinds: rows by LDS-DMA (row-major-only kinds and the ones column: register form)
        char *ldsc = reinterpret_cast<char *>(lds);
        if (job.kind == HS_WGP_256x256) { pair_slice_dma<256, 256, 256, true, true>(job, slice, S, ldsc); return; }
        if (job.kind == HS_WGP_256x80) { pair_slice_dma<256, 128, 80, true, false>(job, slice, S, ldsc); return; }
        if (job.kind == HS_WGP_32x256) { pair_slice_dma<32, 256, 256, false, true>(job, slice, S, ldsc); return; }
        if (job.kind == HS_WGP_256x128_TP) { pair_slice_dma<256, 128, 128, true, true>(job, slice, S, ldsc); return; }
    }
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
    constexpr bool dma = true;
    pj.dma = dma ? 1 : 0;
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
    const size_t lds_reg = 2 * (size_t)RCP * ((256 + PADP) + (256 + PADP)) * sizeof(uint16_t), lds_dma = (size_t)kLdsD;
    const size_t lds = lds_reg > lds_dma ? lds_reg : lds_dma;
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_wgrad_pairs, (int)lds);
    k_wgrad_pairs<<<pj.first[n_jobs], kThreadsP, lds, (hipStream_t)stream>>>(pj);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

}  // extern "C"
