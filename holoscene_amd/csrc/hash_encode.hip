// holoscene_amd/csrc/hash_encode.hip -- multiresolution hash-grid encoder for gfx950 (MI355X).
//
// Hand-written replacement for the reference's five CUDA kernels
// (hashencoder/src/hashencoder.cu:104-595); results are bit-identical to
// oracle/hash_oracle.c for the forward / dy_dx / input-backward / grad_grad
// paths (same operation order, -ffp-contract=off) and equal up to float
// atomic-add ordering for the two scatter paths.
//
// MI355X notes
//  * one lane = one (point, level); a wave covers 64 consecutive points of ONE
//    level, so coordinate loads and level-major stores are fully coalesced and
//    every level parameter lives in SGPRs;
//  * the 8 corner gathers of a lane are issued back to back (8 outstanding
//    global_load_dwordx2) before the first use;
//  * schedule 1 pins each level to one XCD (blocks are dealt to XCDs round
//    robin), so a 4 MiB hashed level stays in that XCD's private 4 MiB L2
//    instead of all 16 levels (48.8 MB) competing for every L2;
//  * dy_dx may be written level-major ([L,B,D*C]) so that each wave writes one
//    contiguous 64*D*C*4-byte run;
//  * scatter uses hardware fp32 L2 atomics (global_atomic_add_f32); NULL
//    outputs skip whole phases (no scatter when only d/dx is needed).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "holoscene_hip.h"
#include "batch_draw.h"
#include "assemble.h"
#include "adam_math.h"

namespace {

// experiments (tools/ab_lib.sh): dynamic LDS nobody uses = fewer resident workgroups (occupancy / bytes in flight per compute unit)
#ifndef HS_BIN_LDS_EXTRA
#define HS_BIN_LDS_EXTRA 0
#endif
#ifndef HS_FWD_LDS_PAD
#define HS_FWD_LDS_PAD 0
#endif
// experiments (tools/ab_lib.sh): the gathers' outputs as non-temporal stores
#ifndef HS_NT_GATHER_OUT
#define HS_NT_GATHER_OUT 0
#endif
#ifndef HS_NT_GATHER_DYDX
#define HS_NT_GATHER_DYDX 0
#endif
constexpr int kThreads = 256;
#ifndef HS_HASH_FWD_THREADS
#define HS_HASH_FWD_THREADS 256
#endif
constexpr int kFwdThreads = HS_HASH_FWD_THREADS;      // the gather kernels' workgroup

struct LevelScales {
    float v[HS_MAX_LEVELS];
};

struct LevelInfo {
    float scale;
    uint32_t res;
    uint32_t table;
    uint32_t offset;
    bool hashed;
};

__device__ __forceinline__ float smoothstep(float t) { return t * t * (3.0f - 2.0f * t); }
__device__ __forceinline__ float smoothstep_d(float t) { return 6 * t * (1.0f - t); }

// blockIdx.x -> (level, chunk).  Schedule 1: block b lands on XCD b%8; XCD x owns
// levels {x, 15-x, 16+x, 31-x, ...} so cheap coarse and expensive fine levels pair up.
__device__ __forceinline__ void decode_block(uint32_t L, uint32_t n_chunks, int schedule, uint32_t &level, uint32_t &chunk, uint32_t first = 0u) {
    const uint32_t bid = blockIdx.x - first;        // (first: workgroups in front of this kernel's own, a multiple of 8 -- the XCD of a block is blockIdx % 8)
    if (schedule == 1) {
        const uint32_t xcd = bid & 7u, j = bid >> 3;
        const uint32_t slot = j / n_chunks;
        chunk = j - slot * n_chunks;
        level = (slot & 1u) ? (slot * 8u + 7u - xcd) : (slot * 8u + xcd);
    } else {
        level = bid / n_chunks;
        chunk = bid - level * n_chunks;
    }
    (void)L;
}

template <int D>
__device__ __forceinline__ LevelInfo level_info(const int32_t *__restrict__ offsets, uint32_t level, const LevelScales &sc) {
    LevelInfo li;
    li.offset = (uint32_t)offsets[level];
    li.table = (uint32_t)offsets[level + 1] - li.offset;
    li.scale = sc.v[level];
    li.res = (uint32_t)ceilf(li.scale) + 1u;
    uint32_t stride = 1;
#pragma unroll
    for (int d = 0; d < D; d++)
        if (stride <= li.table) stride *= li.res;
    li.hashed = stride > li.table;
    return li;
}

template <int D>
__device__ __forceinline__ uint32_t cell_index(const LevelInfo &li, const uint32_t g[D]) {
    uint32_t idx;
    if (li.hashed) {
        idx = g[0];
        if (D > 1) idx ^= g[1] * 2654435761u;
        if (D > 2) idx ^= g[2] * 805459861u;
    } else {
        uint32_t stride = 1;
        idx = 0;
#pragma unroll
        for (int d = 0; d < D; d++) {
            idx += g[d] * stride;
            stride *= li.res;
        }
    }
    if ((li.table & (li.table - 1u)) == 0u) return idx & (li.table - 1u);
    return idx >= li.table ? idx % li.table : idx;
}

// returns false for points outside [0,1]^D
template <int D>
__device__ __forceinline__ bool locate(const float *__restrict__ x, const LevelInfo &li, uint32_t g[D], float w[D], float dw[D]) {
    float p[D];
    bool inside = true;
#pragma unroll
    for (int d = 0; d < D; d++) {
        p[d] = x[d];
        inside = inside && !(p[d] < 0.f || p[d] > 1.f);
    }
    if (!inside) return false;
#pragma unroll
    for (int d = 0; d < D; d++) {
        float pos = p[d] * li.scale;
        const float fl = floorf(pos);
        g[d] = (uint32_t)fl;
        pos -= (float)g[d];
        dw[d] = smoothstep_d(pos);
        w[d] = smoothstep(pos);
    }
    return true;
}

template <int C>
struct Vec {
    float v[C];
};

template <int C>
__device__ __forceinline__ Vec<C> load_entry(const float *__restrict__ p) {
    Vec<C> r;
    if constexpr (C == 1) {
        r.v[0] = p[0];
    } else if constexpr (C == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        r.v[0] = t.x; r.v[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < C; i += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(p + i);
            r.v[i] = t.x; r.v[i + 1] = t.y; r.v[i + 2] = t.z; r.v[i + 3] = t.w;
        }
    }
    return r;
}

// Batched-over-grids launches (hsHashLayout::grid_id): point b reads / updates the table of grid grid_id[b], which starts
// grid_id[b] * grid_stride entries after `embeddings`; every grid has the same level geometry (`offsets`).
__device__ __forceinline__ uint32_t grid_of(const hsHashLayout &lay, uint32_t b, bool inb = true) {
    return (lay.grid_id != nullptr && inb) ? (uint32_t)lay.grid_id[b] : 0u;
}
__device__ __forceinline__ size_t grid_entry0(const hsHashLayout &lay, uint32_t gid, const LevelInfo &li) {
    return (size_t)gid * (size_t)lay.grid_stride + (size_t)li.offset;
}

// ------------------------------------------------------------------------------------ forward
template <int D, int C, bool DYDX>
__global__ __launch_bounds__(kFwdThreads) void k_hash_fwd(const float *__restrict__ x, const float *__restrict__ emb,
                                                        const int32_t *__restrict__ offsets, float *__restrict__ out,
                                                        float *__restrict__ dydx, uint32_t B, uint32_t L, LevelScales sc,
                                                        hsHashLayout lay, uint32_t n_chunks) {
    uint32_t level, chunk;
    if (lay.gate.a != nullptr && !(*lay.gate.a > *lay.gate.b)) return;   // hsGate: this sampler round was not needed
    decode_block(L, n_chunks, lay.schedule, level, chunk);
    const uint32_t b = chunk * kFwdThreads + threadIdx.x;
    if (b >= B) return;
    const LevelInfo li = level_info<D>(offsets, level, sc);
    const float *__restrict__ grid = emb + grid_entry0(lay, grid_of(lay, b), li) * C;
    float *o = out + (int64_t)level * lay.level_stride + (int64_t)b * lay.point_stride;
    float *j = DYDX ? dydx + (int64_t)level * lay.dydx_level_stride + (int64_t)b * lay.dydx_point_stride : nullptr;

    uint32_t g[D];
    float w[D], dw[D];
    // an EMPTY level (offsets[l + 1] == offsets[l]) encodes to zeros: how a grid of fewer than 16 levels is presented to the fused 16-level kernels
    if (li.table == 0u || !locate<D>(x + (size_t)b * D, li, g, w, dw)) {
#pragma unroll
        for (int c = 0; c < C; c++) o[c] = 0.f;
        if (DYDX) {
#pragma unroll
            for (int i = 0; i < D * C; i++) j[i] = 0.f;
        }
        return;
    }
    // issue all 2^D gathers first
    Vec<C> e[1 << D];
#pragma unroll
    for (int corner = 0; corner < (1 << D); corner++) {
        uint32_t gl[D];
#pragma unroll
        for (int d = 0; d < D; d++) gl[d] = g[d] + ((corner >> d) & 1);
        e[corner] = load_entry<C>(grid + (size_t)cell_index<D>(li, gl) * C);
    }
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = 0.f;
#pragma unroll
    for (int corner = 0; corner < (1 << D); corner++) {
        float wt = 1.f;
#pragma unroll
        for (int d = 0; d < D; d++) wt *= ((corner >> d) & 1) ? w[d] : 1 - w[d];
#pragma unroll
        for (int c = 0; c < C; c++) acc[c] += wt * e[corner].v[c];
    }
    if constexpr (C == 2) {
        *reinterpret_cast<float2 *>(o) = make_float2(acc[0], acc[1]);
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) o[c] = acc[c];
    }
    if (DYDX) {
#pragma unroll
        for (int gd = 0; gd < D; gd++) {
            float ga[C];
#pragma unroll
            for (int c = 0; c < C; c++) ga[c] = 0.f;
#pragma unroll
            for (int k = 0; k < (1 << (D - 1)); k++) {
                float wt = li.scale;
                int bits = 0;
#pragma unroll
                for (int nd = 0; nd < D - 1; nd++) {
                    const int d = (nd >= gd) ? nd + 1 : nd;
                    if ((k >> nd) & 1) { wt *= w[d]; bits |= 1 << d; }
                    else wt *= 1 - w[d];
                }
#pragma unroll
                for (int c = 0; c < C; c++) ga[c] += wt * (e[bits | (1 << gd)].v[c] - e[bits].v[c]) * dw[gd];
            }
#pragma unroll
            for (int c = 0; c < C; c++) j[gd * C + c] = ga[c];
        }
    }
}

// ------------------------------------------------------------------------------------ forward, two lanes per point
// The value-only sweep (D = 3) with the two x-corners of a cell on NEIGHBOURING LANES.  The gather's cost is the L1's tag rate -- one
// 128-byte line per clock and CU: 16.8 M corner reads of a 131 072-point sweep over 256 CUs are 33 us of lookups, the kernel took
// 40-49 -- and with one point per lane every corner is its own instruction, so the x-neighbour's line (the same one for 15 of 16 cells,
// hashed levels included: x and x + 1 differ in their low bits only, the hash XORs them into the low index bits) is looked up again.
// With lanes (2p, 2p + 1) = corners (x, x + 1) of point p one gather instruction covers 32 points x 2 corners and touches ~half the
// lines.  The sum keeps the reference's order (corner 0, 1, ..., 7: the x bit alternates): lane 2p adds its own product, then its
// neighbour's through a DPP operand, so the result stays bit-exact with the one-lane kernel.
// DYDX: also d out / d x01 (the one-lane kernel's expressions and summation order: for the x derivative the even lane takes its
// neighbour's four entries through DPP, for y / z the two lanes' terms alternate in the sum as in the value).
template <int C, bool DYDX>
__global__ __launch_bounds__(kFwdThreads) void k_hash_fwd_pair(const float *__restrict__ x, const float *__restrict__ emb,
                                                             const int32_t *__restrict__ offsets, float *__restrict__ out, float *__restrict__ dydx,
                                                             uint32_t B, uint32_t L, LevelScales sc, hsHashLayout lay, uint32_t n_chunks) {
    constexpr int D = 3;
    uint32_t level, chunk;
    if (lay.gate.a != nullptr && !(*lay.gate.a > *lay.gate.b)) return;
    decode_block(L, n_chunks, lay.schedule, level, chunk);
    const uint32_t t = chunk * kFwdThreads + threadIdx.x, b = t >> 1, xb = t & 1u;
    if (b >= B) return;                      // both lanes of a pair leave together
    const LevelInfo li = level_info<D>(offsets, level, sc);
    const float *__restrict__ grid = emb + grid_entry0(lay, grid_of(lay, b), li) * C;
    float *o = out + (int64_t)level * lay.level_stride + (int64_t)b * lay.point_stride;
    float *jo = DYDX ? dydx + (int64_t)level * lay.dydx_level_stride + (int64_t)b * lay.dydx_point_stride : nullptr;
    uint32_t g[D];
    float w[D], dw[D];
    if (li.table == 0u || !locate<D>(x + (size_t)b * D, li, g, w, dw)) {      // (an empty level: zeros, as in k_hash_fwd)
        if (xb == 0) {
            if (!DYDX && lay.out_bf16) {
                o[0] = 0.f;             // (one zero word)
            } else {
#pragma unroll
                for (int c = 0; c < C; c++) o[c] = 0.f;
            }
            if (DYDX) {
#pragma unroll
                for (int i = 0; i < D * C; i++) jo[i] = 0.f;
            }
        }
        return;
    }
    Vec<C> e[4];
#pragma unroll
    for (int yz = 0; yz < 4; yz++) {
        const uint32_t gl[D] = {g[0] + xb, g[1] + (yz & 1), g[2] + (yz >> 1)};
        e[yz] = load_entry<C>(grid + (size_t)cell_index<D>(li, gl) * C);
    }
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = 0.f;
#pragma unroll
    for (int yz = 0; yz < 4; yz++) {
        float wt = 1.f;
        wt *= xb ? w[0] : 1 - w[0];
        wt *= (yz & 1) ? w[1] : 1 - w[1];
        wt *= (yz >> 1) ? w[2] : 1 - w[2];
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float p = wt * e[yz].v[c];
            acc[c] += p;                                                                                     // corner (0, yz) on the even lane
            acc[c] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0xB1, 0xf, 0xf, true));   // corner (1, yz): quad_perm [1,0,3,2]
        }
    }
    if (xb == 0) {
        if constexpr (C == 2) {
            if (!DYDX && lay.out_bf16) {       // one word per (point, level): both channels as bf16 (hsHashLayout::out_bf16)
                typedef float f2_t __attribute__((ext_vector_type(2)));
                typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
                const f2_t v = {acc[0], acc[1]};
                // (NT: the words go to ANOTHER XCD's sweep workgroup through memory in any case; kept out of this XCD's L2 they leave its two levels' lines there)
                if (HS_NT_GATHER_OUT) __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2_t)), reinterpret_cast<uint32_t *>(o));
                else *reinterpret_cast<uint32_t *>(o) = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2_t));
            } else if (HS_NT_GATHER_DYDX && DYDX) {
                typedef float f2nt_t __attribute__((ext_vector_type(2)));
                const f2nt_t vv = {acc[0], acc[1]};
                __builtin_nontemporal_store(vv, reinterpret_cast<f2nt_t *>(o));
            } else {
                *reinterpret_cast<float2 *>(o) = make_float2(acc[0], acc[1]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) o[c] = acc[c];
        }
    }
    if constexpr (DYDX) {
        auto swap = [](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true)); };
        float ga[D][C];
        // d/dx: term k = (y, z) = yz: scale (y ? w1 : 1 - w1) (z ? w2 : 1 - w2) * (e[x = 1] - e[x = 0]) * dw0, all four on the even lane
#pragma unroll
        for (int c = 0; c < C; c++) ga[0][c] = 0.f;
#pragma unroll
        for (int yz = 0; yz < 4; yz++) {
            float wt = li.scale;
            wt *= (yz & 1) ? w[1] : 1 - w[1];
            wt *= (yz >> 1) ? w[2] : 1 - w[2];
#pragma unroll
            for (int c = 0; c < C; c++) ga[0][c] += wt * (swap(e[yz].v[c]) - e[yz].v[c]) * dw[0];
        }
        // d/dy, d/dz: term k = (x, o) with o the other of (y, z): the lane holding x computes it, the even lane adds them in k order
#pragma unroll
        for (int gd = 1; gd < D; gd++) {
#pragma unroll
            for (int c = 0; c < C; c++) ga[gd][c] = 0.f;
#pragma unroll
            for (int ob = 0; ob < 2; ob++) {            // bit of the other dimension (z for gd = 1, y for gd = 2)
                float wt = li.scale;
                wt *= xb ? w[0] : 1 - w[0];
                wt *= ob ? w[3 - gd] : 1 - w[3 - gd];
                const int lo = gd == 1 ? (ob << 1) : ob, hi = gd == 1 ? (1 | (ob << 1)) : (ob | 2);     // yz without / with bit gd
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float t = wt * (e[hi].v[c] - e[lo].v[c]) * dw[gd];
                    ga[gd][c] += t;             // k = (x = 0, ob)
                    ga[gd][c] += swap(t);       // k = (x = 1, ob)
                }
            }
        }
        if (xb == 0) {
#pragma unroll
            for (int gd = 0; gd < D; gd++)
#pragma unroll
                for (int c = 0; c < C; c++) {
                    if (HS_NT_GATHER_DYDX) __builtin_nontemporal_store(ga[gd][c], jo + gd * C + c);
                    else jo[gd * C + c] = ga[gd][c];
                }
        }
    }
}

// ------------------------------------------------------------------------------------ wave-merged scatter
// Neighbouring lanes are neighbouring samples of one ray, so on coarse levels (and wherever the sampler
// concentrates samples near a surface) many lanes of a wave hit the SAME cell and their atomics would
// serialise on one L2 address (measured: level 0 scatter 10x slower on ray-ordered than on random points).
// Lanes whose base cell equals their predecessor's form a run; a segmented inclusive scan (wave shuffles)
// sums the 2^D*C corner contributions over each run and only the run's last lane issues atomics.
// All 64 lanes must call this (inactive lanes pass valid=false).
// Returns whether this lane ends a run (and now holds the run's sums in `cache`).
template <int D, int C>
__device__ __forceinline__ bool wave_merge(const uint32_t g[D], float cache[(1 << D) * C], bool valid, uint32_t key = 0u) {
    const int lane = threadIdx.x & 63;
    bool same_prev = valid && lane > 0;
    same_prev = same_prev && ((uint32_t)__shfl_up(key, 1) == key);     // (batched launches: the same cell of ANOTHER grid is another cell)
#pragma unroll
    for (int d = 0; d < D; d++) {
        const uint32_t pg = __shfl_up(valid ? g[d] : 0xffffffffu, 1);
        same_prev = same_prev && (pg == g[d]);
    }
    const unsigned long long heads = __ballot(!same_prev);
    bool tail = valid;
    if (heads != ~0ull) {  // wave-uniform: at least one run longer than one lane
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            // lane may absorb lane-off iff no run starts in (lane-off, lane]
            const unsigned long long span = (~0ull >> (63 - lane)) & (~0ull << ((lane - off + 1) & 63));
            const bool ok = lane >= off && (heads & span) == 0ull;
            // no lane reaches back `off` lanes inside its run = no run is longer than `off`: the remaining (longer) steps have nothing to add.
            // Runs are mostly 2-4 lanes long, so this ends after two or three of the six steps (each moves (1 << D) * C values across lanes)
            if (__ballot(ok) == 0ull) break;
#pragma unroll
            for (int i = 0; i < (1 << D) * C; i++) {
                const float t = __shfl_up(cache[i], off);
                if (ok) cache[i] += t;
            }
        }
        tail = valid && (lane == 63 || ((heads >> (lane + 1)) & 1ull));
    }
    return tail;
}

template <int D, int C>
__device__ __forceinline__ void scatter_cell(float *__restrict__ gg, const LevelInfo &li, const uint32_t g[D], float cache[(1 << D) * C],
                                             bool valid, uint32_t key = 0u) {
    if (!wave_merge<D, C>(g, cache, valid, key)) return;
#pragma unroll
    for (int corner = 0; corner < (1 << D); corner++) {
        uint32_t gl[D];
#pragma unroll
        for (int d = 0; d < D; d++) gl[d] = g[d] + ((corner >> d) & 1);
        float *e = gg + (size_t)cell_index<D>(li, gl) * C;
#pragma unroll
        for (int c = 0; c < C; c++) {
            // x + (+-0) = x: an exactly-zero contribution needs no atomic.  Volume-rendering weights vanish exactly
            // behind the first surface and far in front of it, so ~1/4 of all samples arrive here with an all-zero
            // cotangent (measured, tools/zero_rows.py) and the scatter is bound by the atomic issue rate.
            const float v = cache[corner * C + c];
            if (v != 0.f) unsafeAtomicAdd(e + c, v);
        }
    }
}

// ------------------------------------------------------------------------------------ binned scatter (hashed levels)
// Random global float atomics retire at one lane-operation per clock per XCD (~21 G/s for the chip) whatever their addresses
// (tools/exp/atomic_xcd.hip); a hashed level gives the wave-merge above nothing to merge, so a dense backward pass spends
// ~1 ms per table issuing 27 M of them.  Scattered plain STORES run 4x faster and LDS atomics 9x, hence two phases:
//   1. (in the scatter kernels) every non-zero corner contribution becomes a 4+4C-byte record {cell within bin, values},
//      appended to the bin (1/128 of the level's table) its cell falls in: ranks inside the workgroup by LDS atomics, one
//      global atomic per (workgroup, touched bin) to reserve space;
//   2. k_hash_bin_reduce: one workgroup per (level, bin) accumulates its records in LDS (32 KB = 4 096 cells x 2 floats at
//      T = 2^19) and adds the result to the table with plain coalesced read-modify-writes -- it owns those cells.
// Dense (coarse) levels went through the wave-merged atomic path alone at first; with dense cotangents (early training: every sample
// contributes) their atomics were the tail of the launch (418 us of hs_hash_bwd_jac at beta = 0.1, 251 us since they are binned too:
// bins = slabs of ceil(res^3 / kBins) cells).  Records that do not fit the bin's capacity fall back to atomics.
constexpr uint32_t kBins = HS_SCATTER_BINS;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kReduceLds = 32768u * 128u / kBins;   // bytes of LDS per reduce workgroup: 4 096 cells x 2 floats at 128 bins; five workgroups per CU hide each other's latency
#ifndef HS_BIN_THREADS
#define HS_BIN_THREADS 512
#endif
constexpr int kBinThreads = HS_BIN_THREADS;                // threads of a reduce / step workgroup; kBinV quads per thread cover the bin in one pass
constexpr int kBinV = (int)(kReduceLds / 16u) / kBinThreads;
static_assert(kBinV >= 1 && kBinV * kBinThreads * 16 == (int)kReduceLds, "a reduce workgroup covers its bin in one pass of whole quads");

template <int C>
struct BinRecord { uint32_t cell; float v[C]; };

// cells per bin: table / kBins for the power-of-two hashed tables; for a dense level (res^3 cells) the quotient rounded up to a
// number of cells whose C floats fill whole float4s (C = 2: even, as before; C = 1: a multiple of four -- k_hash_bin_reduce zeroes,
// accumulates and writes back whole quads) -- the last bins are then partly or wholly empty.
template <int C>
__device__ __forceinline__ uint32_t bin_width(const LevelInfo &li) {
    constexpr uint32_t q = (C % 4 == 0) ? 1u : (C % 2 == 0) ? 2u : 4u;      // cells per float4 group
    const uint32_t w = (li.table + kBins - 1u) / kBins;
    return (w + q - 1u) / q * q;
}

__device__ int g_bin_dense = 1;      // HOLOSCENE_BIN_DENSE=0 (A/B): dense levels keep the wave-merged atomic path (set once by the launchers)
template <int C>
__device__ __forceinline__ bool binned_level(const LevelInfo &li, const void *ws) {
    if (ws == nullptr || (size_t)bin_width<C>(li) * C * sizeof(float) > kReduceLds) return false;
    if (li.hashed) return (li.table & (li.table - 1u)) == 0u && li.table >= 64u * kBins;
    return g_bin_dense != 0 && li.table >= 32u * kBins;
}

// all threads of the workgroup call this (level is workgroup-uniform)
template <int D, int C>
__device__ __forceinline__ void bin_cell(const hsHashLayout &lay, float *__restrict__ gg, const LevelInfo &li, uint32_t level, const uint32_t g[D],
                                         float cache[(1 << D) * C], bool valid) {
    __shared__ uint32_t hist[kBins], base[kBins];
    // samples of a ray crowd around the surface: even on the finest level consecutive lanes often share a cell, so runs are
    // merged in the wave first and only their last lane emits records (632 k -> far fewer per level in the benchmark state)
    valid = wave_merge<D, C>(g, cache, valid);
    uint32_t *counts = reinterpret_cast<uint32_t *>(lay.scatter_ws);
    BinRecord<C> *records = reinterpret_cast<BinRecord<C> *>(reinterpret_cast<char *>(lay.scatter_ws) + HS_MAX_LEVELS * kBins * sizeof(uint32_t));
    const uint32_t per_bin = bin_width<C>(li);
    // cell -> bin without an integer division per corner (~25 instructions each, twice per corner): the hashed levels' bins are a power
    // of two wide (a shift); a dense level's quotient comes from the float reciprocal, exact after one correction step (cells < 2^24)
    const bool p2 = (per_bin & (per_bin - 1u)) == 0u;
    const uint32_t sh = 31u - (uint32_t)__clz((int)per_bin);
    const float inv_bin = 1.0f / (float)per_bin;
    auto bin_of = [&](uint32_t c) -> uint32_t {
        if (p2) return c >> sh;                  // (workgroup-uniform branch)
        uint32_t q = (uint32_t)((float)c * inv_bin);
        if (q * per_bin > c) q--;
        else if ((q + 1u) * per_bin <= c) q++;
        return q;
    };
    if (threadIdx.x < kBins) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t cell[1 << D], rank[1 << D], bins[1 << D];
    bool nz[1 << D];
#pragma unroll
    for (int corner = 0; corner < (1 << D); corner++) {
        uint32_t gl[D];
#pragma unroll
        for (int d = 0; d < D; d++) gl[d] = g[d] + ((corner >> d) & 1);
        bool any = false;
#pragma unroll
        for (int c = 0; c < C; c++) any |= cache[corner * C + c] != 0.f;
        nz[corner] = valid && any;
        cell[corner] = nz[corner] ? cell_index<D>(li, gl) : 0u;
        bins[corner] = bin_of(cell[corner]);
        rank[corner] = nz[corner] ? atomicAdd(&hist[bins[corner]], 1u) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < kBins && hist[threadIdx.x] != 0u) base[threadIdx.x] = atomicAdd(&counts[level * kBins + threadIdx.x], hist[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int corner = 0; corner < (1 << D); corner++) {
        if (!nz[corner]) continue;
        const uint32_t bin = bins[corner], pos = base[bin] + rank[corner];
        if (pos < lay.scatter_cap) {
            BinRecord<C> r;
            r.cell = cell[corner] - bin * per_bin;
#pragma unroll
            for (int c = 0; c < C; c++) r.v[c] = cache[corner * C + c];
            records[((size_t)level * kBins + bin) * lay.scatter_cap + pos] = r;
        } else {   // bin full: straight to the table
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float v = cache[corner * C + c];
                if (v != 0.f) unsafeAtomicAdd(gg + (size_t)cell[corner] * C + c, v);
            }
        }
    }
}

// (a kernel, not hipMemsetAsync: a memset node inside a captured graph cost ~0.5 ms per replay on this stack)
__global__ void k_zero_u32(uint32_t *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

template <int D, int C>
__global__ __launch_bounds__(kBinThreads) void k_hash_bin_reduce(float *__restrict__ gemb, const int32_t *__restrict__ offsets, uint32_t L, LevelScales sc,
                                                          hsHashLayout lay) {
    extern __shared__ float acc[];
    const uint32_t level = blockIdx.y, bin = blockIdx.x;
    const LevelInfo li = level_info<D>(offsets, level, sc);
    if (!binned_level<C>(li, lay.scatter_ws)) return;
    const uint32_t *counts = reinterpret_cast<const uint32_t *>(lay.scatter_ws);
    const BinRecord<C> *records = reinterpret_cast<const BinRecord<C> *>(reinterpret_cast<const char *>(lay.scatter_ws) +
                                                                          HS_MAX_LEVELS * kBins * sizeof(uint32_t)) +
                                  ((size_t)level * kBins + bin) * lay.scatter_cap;
    const uint32_t n = min(counts[level * kBins + bin], lay.scatter_cap);
    if (n == 0u) return;
    const uint32_t per_bin = bin_width<C>(li), first = bin * per_bin;
    if (first >= li.table) {                                     // (dense levels: bins past the end of the table hold nothing)
        __syncthreads();
        if (threadIdx.x == 0) const_cast<uint32_t *>(counts)[level * kBins + bin] = 0u;
        return;
    }
    const uint32_t nfl = min(per_bin, li.table - first) * C;     // floats of the table this bin covers
    const uint32_t nvec = per_bin * C / 4;                       // bin_width: a whole number of float4
    float4 *acc4 = reinterpret_cast<float4 *>(acc);
    // the table's current content of this thread's quads does not depend on the records: requested now, it arrives under the record
    // phase instead of costing a memory round trip (four serialised ones, when the loads were issued per touched quad) at the end of a
    // workgroup whose whole life is a chain of such round trips (2 048 workgroups on 1 280 slots: the launch lasts ~2 workgroup lives)
    float *dstf = gemb + ((size_t)li.offset + (size_t)first) * C;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dstf) & 15) == 0 && nfl == per_bin * C;
    constexpr int kV = kBinV;                                    // kReduceLds / 16 B quads = kV x kBinThreads: one pass
    float4 t[kV];
    if (vec_ok) {
#pragma unroll
        for (int k = 0; k < kV; k++) {
            const uint32_t j = threadIdx.x + k * blockDim.x;
            if (j < nvec) t[k] = reinterpret_cast<const float4 *>(dstf)[j];
        }
    }
    for (uint32_t i = threadIdx.x; i < nvec; i += blockDim.x) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // everybody has read this bin's count: leave it at zero for the next scatter through this work space (hsHashLayout::ws_clean: the
    // clearing launch in front of every scatter was 5 us of a ~85 us stage, twice per iteration)
    if (threadIdx.x == 0) const_cast<uint32_t *>(counts)[level * kBins + bin] = 0u;
    // records: independent loads, four in flight per thread before the first LDS atomic (eight cost an occupancy step: 74 registers,
    // 104 vs 101 us per scatter)
    uint32_t i = threadIdx.x;
    for (; i + 3 * blockDim.x < n; i += 4 * blockDim.x) {
        BinRecord<C> r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = records[i + k * blockDim.x];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int c = 0; c < C; c++) atomicAdd(&acc[r[k].cell * C + c], r[k].v[c]);
    }
    for (; i < n; i += blockDim.x) {
        const BinRecord<C> r = records[i];
#pragma unroll
        for (int c = 0; c < C; c++) atomicAdd(&acc[r.cell * C + c], r.v[c]);
    }
    __syncthreads();
    // this workgroup owns the bin's cells: plain read-modify-writes.  A dense level may start at an odd entry and end inside the bin:
    // 4-byte accesses there (at most ~3 200 floats per bin)
    if (!vec_ok) {
        for (uint32_t j = threadIdx.x; j < nfl; j += blockDim.x) {
            const float a = acc[j];
            if (a != 0.f) dstf[j] += a;
        }
        return;
    }
    float4 *dst = reinterpret_cast<float4 *>(dstf);
#pragma unroll
    for (int k = 0; k < kV; k++) {
        const uint32_t j = threadIdx.x + k * blockDim.x;
        if (j >= nvec) continue;
        const float4 a = acc4[j];
        if (a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f) {
            t[k].x += a.x; t[k].y += a.y; t[k].z += a.z; t[k].w += a.w;
            dst[j] = t[k];
        }
    }
}

// ------------------------------------------------------------------------------------ reduce-and-step (hsTableStep)
// The owner of a bin holds the FINAL gradient of its cells in LDS when this scatter is the table's only gradient producer of the
// iteration: it takes the Adam step there.  Against k_hash_bin_reduce + the table's share of k_adam_flat + the zero-fill of the gradient
// table this removes, per table and iteration, one 48.8 MB write (zero-fill), one read-modify-write of the touched quads (reduce) and
// one 48.8 MB read (the optimiser's g stream): what is left is the optimiser's own minimum, 3 reads + 3 writes per parameter.
// Every (level, bin) workgroup steps ALL its cells (dense Adam: an entry without a gradient decays its moments and still moves);
// levels that are not binned, and overflowed bins, add what the scatter put into `gemb` with atomics and return those floats to zero.
template <int D, int C>
__global__ __launch_bounds__(kBinThreads) void k_hash_bin_step(float *__restrict__ gemb, const int32_t *__restrict__ offsets, uint32_t L, LevelScales sc,
                                                        hsHashLayout lay, hsTableStep ts) {
    extern __shared__ float acc[];
    const uint32_t level = blockIdx.y, bin = blockIdx.x;
    const LevelInfo li = level_info<D>(offsets, level, sc);
    const hsAdamState *st = ts.state;
    const float step_size = st->step_size[ts.group], bc2_sqrt = st->bc2_sqrt;
    const bool binned = binned_level<C>(li, lay.scatter_ws);
    uint32_t *counts = reinterpret_cast<uint32_t *>(lay.scatter_ws);
    const uint32_t per_bin = bin_width<C>(li), first = bin * per_bin;
    const uint32_t total = binned ? counts[level * kBins + bin] : 0u;
    const uint32_t n = binned ? min(total, lay.scatter_cap) : 0u;
    const bool spilled = !binned || total > lay.scatter_cap || ts.prior != 0;      // `gemb` may hold contributions to this bin's cells
    if (first >= li.table) {                                     // (dense levels: bins past the end of the table hold nothing)
        __syncthreads();
        if (binned && threadIdx.x == 0) counts[level * kBins + bin] = 0u;
        return;
    }
    const uint32_t nfl = min(per_bin, li.table - first) * C;     // floats of the table this bin covers
    const size_t e0 = ((size_t)li.offset + (size_t)first) * C;
    float *G = gemb + e0, *P = ts.p + e0, *M = ts.m + e0, *V = ts.v + e0;
    if (!binned) {                                               // (a slab of a level that went through atomics alone: no LDS stage)
        for (uint32_t j = threadIdx.x; j < nfl; j += blockDim.x) {
            const float g = G[j];
            if (g != 0.f) G[j] = 0.f;
            float pp = P[j], mm = M[j], vv = V[j];
            adam1(pp, g, mm, vv, step_size, bc2_sqrt, ts.beta1, ts.beta2, ts.eps, ts.grad_scale);
            P[j] = pp; M[j] = mm; V[j] = vv;
        }
        return;
    }
    const BinRecord<C> *records = reinterpret_cast<const BinRecord<C> *>(reinterpret_cast<const char *>(lay.scatter_ws) +
                                                                          HS_MAX_LEVELS * kBins * sizeof(uint32_t)) +
                                  ((size_t)level * kBins + bin) * lay.scatter_cap;
    const uint32_t nvec = per_bin * C / 4;                       // bin_width: a whole number of float4
    float4 *acc4 = reinterpret_cast<float4 *>(acc);
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(M) |
                          reinterpret_cast<uintptr_t>(V)) & 15) == 0 && nfl == per_bin * C;
    constexpr int kV = kBinV;                                    // kReduceLds / 16 B quads = kV x kBinThreads: one pass
    // parameter and moments of this thread's quads do not depend on the records: requested now, they arrive under the record phase
    f32x4 tp[kV], tm[kV], tv[kV];
    if (vec_ok) {
#pragma unroll
        for (int k = 0; k < kV; k++) {
            const uint32_t j = threadIdx.x + k * blockDim.x;
            if (j < nvec) {
                tp[k] = reinterpret_cast<const f32x4 *>(P)[j];
                tm[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(M) + j);
                tv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(V) + j);
            }
        }
    }
    for (uint32_t i = threadIdx.x; i < nvec; i += blockDim.x) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (threadIdx.x == 0) counts[level * kBins + bin] = 0u;      // (everybody has read it) clean for the next scatter: hsHashLayout::ws_clean
    uint32_t i = threadIdx.x;
    for (; i + 3 * blockDim.x < n; i += 4 * blockDim.x) {
        BinRecord<C> r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = records[i + k * blockDim.x];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int c = 0; c < C; c++) atomicAdd(&acc[r[k].cell * C + c], r[k].v[c]);
    }
    for (; i < n; i += blockDim.x) {
        const BinRecord<C> r = records[i];
#pragma unroll
        for (int c = 0; c < C; c++) atomicAdd(&acc[r.cell * C + c], r.v[c]);
    }
    __syncthreads();
    if (!vec_ok) {       // a dense level may start at an odd entry and end inside the bin: 4-byte accesses there
        for (uint32_t j = threadIdx.x; j < nfl; j += blockDim.x) {
            float g = acc[j];
            if (spilled) {
                const float g0 = G[j];
                if (g0 != 0.f) { g += g0; G[j] = 0.f; }
            }
            float pp = P[j], mm = M[j], vv = V[j];
            adam1(pp, g, mm, vv, step_size, bc2_sqrt, ts.beta1, ts.beta2, ts.eps, ts.grad_scale);
            P[j] = pp; M[j] = mm; V[j] = vv;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kV; k++) {
        const uint32_t j = threadIdx.x + k * blockDim.x;
        if (j >= nvec) continue;
        float4 a = acc4[j];
        if (spilled) {
            const float4 g0 = reinterpret_cast<const float4 *>(G)[j];
            if (g0.x != 0.f || g0.y != 0.f || g0.z != 0.f || g0.w != 0.f) {
                a.x += g0.x; a.y += g0.y; a.z += g0.z; a.w += g0.w;
                reinterpret_cast<float4 *>(G)[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float ga[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float pp = tp[k][e], mm = tm[k][e], vv = tv[k][e];
            adam1(pp, ga[e], mm, vv, step_size, bc2_sqrt, ts.beta1, ts.beta2, ts.eps, ts.grad_scale);
            tp[k][e] = pp; tm[k][e] = mm; tv[k][e] = vv;
        }
        reinterpret_cast<f32x4 *>(P)[j] = tp[k];
        __builtin_nontemporal_store(tm[k], reinterpret_cast<f32x4 *>(M) + j);
        __builtin_nontemporal_store(tv[k], reinterpret_cast<f32x4 *>(V) + j);
    }
}

bool step_ok(const hsHashLayout &lay, const float *grad_embeddings) {
    const hsTableStep *ts = lay.step;
    return !ts || (grad_embeddings && ts->p && ts->m && ts->v && ts->state && ts->group >= 0 && ts->group < HS_ADAM_MAX_GROUPS && !lay.grid_id);
}

// the reduction that follows a scatter: plain (add to the gradient table) or with the optimiser step
template <int D, int C>
void launch_bin_reduce(float *grad_embeddings, const int32_t *offsets, uint32_t L, const LevelScales &sc, const hsHashLayout &lay, hipStream_t st) {
    if (lay.step) {
        hsHashLayout dev = lay;
        dev.step = nullptr;              // (a host pointer: the kernel receives the structure by value)
        k_hash_bin_step<D, C><<<dim3(kBins, L), dim3(kBinThreads), kReduceLds + HS_BIN_LDS_EXTRA, st>>>(grad_embeddings, offsets, L, sc, dev, *lay.step);
    } else if (lay.scatter_ws) {
        k_hash_bin_reduce<D, C><<<dim3(kBins, L), dim3(kBinThreads), kReduceLds, st>>>(grad_embeddings, offsets, L, sc, lay);
    }
}

// ------------------------------------------------------------------------------------ first backward: scatter
template <int D, int C>
__device__ __forceinline__ void hash_bwd_scatter_body(const float *__restrict__ grad, const float *__restrict__ x,
                                                      const int32_t *__restrict__ offsets, float *__restrict__ gemb,
                                                      uint32_t B, uint32_t L, const LevelScales &sc, const hsHashLayout &lay, uint32_t n_chunks, uint32_t first) {
    uint32_t level, chunk;
    decode_block(L, n_chunks, lay.schedule, level, chunk, first);
    const uint32_t b = chunk * kThreads + threadIdx.x;
    const LevelInfo li = level_info<D>(offsets, level, sc);
    if (li.table == 0u) return;       // an empty level has no table to scatter into (block-uniform)
    uint32_t g[D];
    float w[D], dw[D];
    const bool valid = b < B && locate<D>(x + (size_t)b * D, li, g, w, dw);
    float cache[(1 << D) * C];
    if (valid) {
        const float *go = grad + (int64_t)level * lay.level_stride + (int64_t)b * lay.point_stride;
        float gv[C];
#pragma unroll
        for (int c = 0; c < C; c++) gv[c] = go[c];
#pragma unroll
        for (int corner = 0; corner < (1 << D); corner++) {
            float wt = 1.f;
#pragma unroll
            for (int d = 0; d < D; d++) wt *= ((corner >> d) & 1) ? w[d] : 1 - w[d];
#pragma unroll
            for (int c = 0; c < C; c++) cache[corner * C + c] = wt * gv[c];
        }
    } else {
#pragma unroll
        for (int i = 0; i < (1 << D) * C; i++) cache[i] = 0.f;
#pragma unroll
        for (int d = 0; d < D; d++) g[d] = 0xffffffffu;
    }
    const uint32_t gid = grid_of(lay, b, b < B);
    if (binned_level<C>(li, lay.scatter_ws)) bin_cell<D, C>(lay, gemb + (size_t)li.offset * C, li, level, g, cache, valid);      // (never with grid_id: the launchers refuse)
    else scatter_cell<D, C>(gemb + grid_entry0(lay, gid, li) * C, li, g, cache, valid, gid);
}

template <int D, int C>
__global__ __launch_bounds__(kThreads) void k_hash_bwd_scatter(const float *__restrict__ grad, const float *__restrict__ x,
                                                                const int32_t *__restrict__ offsets, float *__restrict__ gemb,
                                                                uint32_t B, uint32_t L, LevelScales sc, hsHashLayout lay, uint32_t n_chunks) {
    hash_bwd_scatter_body<D, C>(grad, x, offsets, gemb, B, L, sc, lay, n_chunks, 0u);
}

// The same launch with the NEXT iteration's batch draw riding in front (batch_draw.h; include/holoscene_hip.h: hs_hash_bwd_draw): the draw is a
// 22-us chain of dependent round trips on a handful of workgroups and nothing in a training iteration reads its static batch block after the loss --
// as the first link of the next iteration it is on the critical path, under this scatter (39 us on 6 000 workgroups) it is not.  first = the draw's
// workgroups rounded up to 8, so that the scatter's blocks keep their XCDs.
struct DrawRider { hsDrawSched draw; int32_t blocks, n_uniform, total_pixels, n_out; int64_t *out; DrawGatherJobs jobs; };
// ... and a backward stage's slice sums (assemble.h: hs_assemble's jobs): their only reader is the end-of-pass epilogue, their inputs are complete before the
// scatter starts, and as a launch of their own they were 9-11 us between two long kernels
struct Riders { DrawRider draw; AsmJobs sums; int32_t sum_blocks; };
static_assert(kThreads == kDrawThreads, "the riders' workgroups are the scatter's size");
static_assert(sizeof(Riders) < 3200, "kernel arguments: 4 KB in all");
__device__ __forceinline__ void run_rider(const Riders &r, int block) {     // block < first
    if (block < r.draw.blocks) draw_gather_sched_body(block, r.draw.blocks, r.draw.draw, r.draw.n_uniform, r.draw.total_pixels, r.draw.n_out, r.draw.out, r.draw.jobs);
    else if (block - r.draw.blocks < r.sum_blocks) assemble_body(r.sums, block - r.draw.blocks);
}
template <int D, int C>
__global__ __launch_bounds__(kThreads) void k_hash_bwd_scatter_draw(const float *__restrict__ grad, const float *__restrict__ x,
                                                                     const int32_t *__restrict__ offsets, float *__restrict__ gemb,
                                                                     uint32_t B, uint32_t L, LevelScales sc, hsHashLayout lay, uint32_t n_chunks, Riders r, uint32_t first) {
    if (blockIdx.x < first) {
        run_rider(r, (int)blockIdx.x);
        return;
    }
    hash_bwd_scatter_body<D, C>(grad, x, offsets, gemb, B, L, sc, lay, n_chunks, first);
}

// ------------------------------------------------------------------------------------ first backward: d/dx
// grad_x[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]   (hashencoder.cu:347-372; same summation order)
template <int D, int C>
__global__ __launch_bounds__(kThreads) void k_hash_bwd_input(const float *__restrict__ grad, const float *__restrict__ dydx,
                                                              float *__restrict__ gx, uint32_t B, uint32_t L, hsHashLayout lay) {
    const uint32_t b = blockIdx.x * kThreads + threadIdx.x;
    if (b >= B) return;
    float r[D];
#pragma unroll
    for (int d = 0; d < D; d++) r[d] = 0.f;
    for (uint32_t l = 0; l < L; l++) {
        const float *go = grad + (int64_t)l * lay.level_stride + (int64_t)b * lay.point_stride;
        const float *j = dydx + (int64_t)l * lay.dydx_level_stride + (int64_t)b * lay.dydx_point_stride;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float gc = go[c];
#pragma unroll
            for (int d = 0; d < D; d++) r[d] += gc * j[d * C + c];
        }
    }
#pragma unroll
    for (int d = 0; d < D; d++) gx[(size_t)b * D + d] = r[d];
}

// ------------------------------------------------------------------------------------ second backward
template <int D, int C>
__global__ __launch_bounds__(kThreads) void k_hash_bwd2(const float *__restrict__ grad, const float *__restrict__ x,
                                                         const int32_t *__restrict__ offsets, const float *__restrict__ dydx,
                                                         const float *__restrict__ ggx, float *__restrict__ grad_grad,
                                                         float *__restrict__ g2emb, uint32_t B, uint32_t L, LevelScales sc,
                                                         hsHashLayout lay, uint32_t n_chunks) {
    uint32_t level, chunk;
    decode_block(L, n_chunks, lay.schedule, level, chunk);
    const uint32_t b = chunk * kThreads + threadIdx.x;
    const bool inb = b < B;
    float gx[D];
#pragma unroll
    for (int d = 0; d < D; d++) gx[d] = inb ? ggx[(size_t)b * D + d] : 0.f;
    const int64_t foff = (int64_t)level * lay.level_stride + (int64_t)b * lay.point_stride;
    if (grad_grad && inb) {  // hashencoder.cu:376-428
        const float *j = dydx + (int64_t)level * lay.dydx_level_stride + (int64_t)b * lay.dydx_point_stride;
#pragma unroll
        for (int c = 0; c < C; c++) {
            float r = 0.f;
#pragma unroll
            for (int d = 0; d < D; d++) r += gx[d] * j[d * C + c];
            grad_grad[foff + c] = r;
        }
    }
    if (!g2emb) return;  // uniform
    const LevelInfo li = level_info<D>(offsets, level, sc);
    if (li.table == 0u) return;       // (block-uniform)
    uint32_t g[D];
    float w[D], dw[D];
    const bool valid = inb && locate<D>(x + (size_t)b * D, li, g, w, dw);
    float cache[(1 << D) * C];  // hashencoder.cu:507-549
#pragma unroll
    for (int i = 0; i < (1 << D) * C; i++) cache[i] = 0.f;
    if (valid) {
        float gv[C];
#pragma unroll
        for (int c = 0; c < C; c++) gv[c] = grad[foff + c];
#pragma unroll
        for (int gd = 0; gd < D; gd++) {
#pragma unroll
            for (int k = 0; k < (1 << (D - 1)); k++) {
                float wt = li.scale;
                int bits = 0;
#pragma unroll
                for (int nd = 0; nd < D - 1; nd++) {
                    const int d = (nd >= gd) ? nd + 1 : nd;
                    if ((k >> nd) & 1) { wt *= w[d]; bits |= 1 << d; }
                    else wt *= 1 - w[d];
                }
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float v = wt * gv[c] * gx[gd] * dw[gd];
                    cache[(bits | (1 << gd)) * C + c] += v;
                    cache[bits * C + c] -= v;
                }
            }
        }
    } else {
#pragma unroll
        for (int d = 0; d < D; d++) g[d] = 0xffffffffu;
    }
    const uint32_t gid = grid_of(lay, b, inb);
    scatter_cell<D, C>(g2emb + grid_entry0(lay, gid, li) * C, li, g, cache, valid, gid);
}

// ------------------------------------------------------------------------------------ value+Jacobian backward
// grad_emb += d<feat, g_feat>/dE + d<dy_dx, g_dydx>/dE in ONE scatter pass.  The second term is the
// reference's second-backward embedding kernel (hashencoder.cu:432-595) with its rank-one cotangent
// grad[l,b,c]*ggx[b,d] generalised to an arbitrary g_dydx[l,b,d,c]; like the reference it ignores
// d/dx of dy_dx (hashgrid.py:101).
template <int D, int C>
__device__ __forceinline__ void hash_bwd_jac_body(const float *__restrict__ g_feat, const float *__restrict__ g_dydx,
                                                  const float *__restrict__ x, const int32_t *__restrict__ offsets,
                                                  float *__restrict__ gemb, uint32_t B, uint32_t L, const LevelScales &sc,
                                                  const hsHashLayout &lay, uint32_t n_chunks, uint32_t first) {
    uint32_t level, chunk;
    decode_block(L, n_chunks, lay.schedule, level, chunk, first);
    const uint32_t b = chunk * kThreads + threadIdx.x;
    const LevelInfo li = level_info<D>(offsets, level, sc);
    if (li.table == 0u) return;       // an empty level has no table to scatter into (block-uniform)
    uint32_t g[D];
    float w[D], dw[D];
    const bool valid = b < B && locate<D>(x + (size_t)b * D, li, g, w, dw);
    float cache[(1 << D) * C];
#pragma unroll
    for (int i = 0; i < (1 << D) * C; i++) cache[i] = 0.f;
    if (!valid) {
#pragma unroll
        for (int d = 0; d < D; d++) g[d] = 0xffffffffu;
    }
    if (valid && g_feat) {
        const float *go = g_feat + (int64_t)level * lay.level_stride + (int64_t)b * lay.point_stride;
        float gv[C];
#pragma unroll
        for (int c = 0; c < C; c++) gv[c] = go[c];
#pragma unroll
        for (int corner = 0; corner < (1 << D); corner++) {
            float wt = 1.f;
#pragma unroll
            for (int d = 0; d < D; d++) wt *= ((corner >> d) & 1) ? w[d] : 1 - w[d];
#pragma unroll
            for (int c = 0; c < C; c++) cache[corner * C + c] = wt * gv[c];
        }
    }
    const bool rank1 = lay.r1_ux != nullptr && b < lay.r1_n;
    if (valid && (rank1 || g_dydx)) {
        float G[D * C];
        if (rank1) {        // (scale * ux[level, c]) * g[d]: the product hs_trunk_rr_bwd_grad would have stored, same order of operations
            const float *ux = lay.r1_ux + ((size_t)b * L + level) * C, *gg = lay.r1_g + (size_t)b * D;
#pragma unroll
            for (int d = 0; d < D; d++)
#pragma unroll
                for (int c = 0; c < C; c++) G[d * C + c] = lay.r1_scale * ux[c] * gg[d];
        } else {
            const float *gj = g_dydx + (int64_t)level * lay.dydx_level_stride + (int64_t)b * lay.dydx_point_stride;
#pragma unroll
            for (int i = 0; i < D * C; i++) G[i] = gj[i];
        }
#pragma unroll
        for (int gd = 0; gd < D; gd++) {
#pragma unroll
            for (int k = 0; k < (1 << (D - 1)); k++) {
                float wt = li.scale;
                int bits = 0;
#pragma unroll
                for (int nd = 0; nd < D - 1; nd++) {
                    const int d = (nd >= gd) ? nd + 1 : nd;
                    if ((k >> nd) & 1) { wt *= w[d]; bits |= 1 << d; }
                    else wt *= 1 - w[d];
                }
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float v = wt * G[gd * C + c] * dw[gd];
                    cache[(bits | (1 << gd)) * C + c] += v;
                    cache[bits * C + c] -= v;
                }
            }
        }
    }
    const uint32_t gid = grid_of(lay, b, b < B);
    if (binned_level<C>(li, lay.scatter_ws)) bin_cell<D, C>(lay, gemb + (size_t)li.offset * C, li, level, g, cache, valid);      // (never with grid_id: the launchers refuse)
    else scatter_cell<D, C>(gemb + grid_entry0(lay, gid, li) * C, li, g, cache, valid, gid);
}

template <int D, int C>
__global__ __launch_bounds__(kThreads) void k_hash_bwd_jac(const float *__restrict__ g_feat, const float *__restrict__ g_dydx,
                                                            const float *__restrict__ x, const int32_t *__restrict__ offsets,
                                                            float *__restrict__ gemb, uint32_t B, uint32_t L, LevelScales sc,
                                                            hsHashLayout lay, uint32_t n_chunks) {
    hash_bwd_jac_body<D, C>(g_feat, g_dydx, x, offsets, gemb, B, L, sc, lay, n_chunks, 0u);
}

// with riders in front (see k_hash_bwd_scatter_draw): the trunk's slice sums
template <int D, int C>
__global__ __launch_bounds__(kThreads) void k_hash_bwd_jac_riders(const float *__restrict__ g_feat, const float *__restrict__ g_dydx,
                                                                   const float *__restrict__ x, const int32_t *__restrict__ offsets,
                                                                   float *__restrict__ gemb, uint32_t B, uint32_t L, LevelScales sc,
                                                                   hsHashLayout lay, uint32_t n_chunks, Riders r, uint32_t first) {
    if (blockIdx.x < first) {
        run_rider(r, (int)blockIdx.x);
        return;
    }
    hash_bwd_jac_body<D, C>(g_feat, g_dydx, x, offsets, gemb, B, L, sc, lay, n_chunks, first);
}

// ------------------------------------------------------------------------------------ host side
LevelScales make_scales(uint32_t L, float S, uint32_t H) {
    LevelScales sc;
    for (uint32_t l = 0; l < HS_MAX_LEVELS; l++) sc.v[l] = l < L ? exp2f((float)l * S) * (float)H - 1.0f : 0.f;  // hashencoder.cu:152
    return sc;
}

hsHashLayout reference_layout(uint32_t B, uint32_t D, uint32_t C, uint32_t L) {
    hsHashLayout lay = {};
    lay.level_stride = (int64_t)B * C;  // [L,B,C]
    lay.point_stride = C;
    lay.dydx_level_stride = (int64_t)D * C;  // [B,L,D,C]
    lay.dydx_point_stride = (int64_t)L * D * C;
    lay.schedule = 0;
    lay.gate.a = nullptr;
    lay.gate.b = nullptr;
    lay.scatter_ws = nullptr;
    lay.scatter_cap = 0;
    lay.grid_id = nullptr;
    lay.grid_stride = 0;
    lay.ws_clean = 0;
    lay.out_bf16 = 0;
    lay.r1_ux = lay.r1_g = nullptr;
    lay.r1_n = 0;
    lay.r1_scale = 0.f;
    lay.step = nullptr;
    return lay;
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

// (the dense levels go through the bins too: g_bin_dense stays 1 -- the wave-merged atomic path for them measured slower in round 4)
void apply_bin_dense_switch() {}

bool dims_ok(uint32_t D, uint32_t C, uint32_t L) { return (D == 2 || D == 3) && (C == 1 || C == 2 || C == 4 || C == 8) && L >= 1 && L <= HS_MAX_LEVELS; }

template <int V>
using Int = std::integral_constant<int, V>;

// Calls f(Int<D>{}, Int<C>{}) for the runtime (D, C) pair; dims_ok() has already vetted them.
template <class F>
void dispatch_dc(uint32_t D, uint32_t C, F &&f) {
    auto with_c = [&](auto d) {
        switch (C) {
            case 1: f(d, Int<1>{}); break;
            case 2: f(d, Int<2>{}); break;
            case 4: f(d, Int<4>{}); break;
            default: f(d, Int<8>{}); break;
        }
    };
    if (D == 3) with_c(Int<3>{});
    else with_c(Int<2>{});
}

// D = 3 gathers run two lanes per point (k_hash_fwd_pair); k_hash_fwd -- one lane per point -- serves D = 2
constexpr bool pair_forward() { return true; }

}  // namespace

extern "C" {

int hs_hash_fwd(const float *inputs, const float *embeddings, const int32_t *offsets, float *outputs, uint32_t B, uint32_t D,
                uint32_t C, uint32_t L, float S, uint32_t H, float *dy_dx, const hsHashLayout *layout, void *stream) {
    if (!dims_ok(D, C, L)) return HS_ERR_ARG;
    if (B == 0) return HS_OK;  // empty batches carry NULL data pointers
    if (!inputs || !embeddings || !offsets || !outputs || !layout) return HS_ERR_NULL;
    hsHashLayout lay = *layout;
    if (lay.grid_id && (lay.scatter_ws || lay.grid_stride <= 0)) return HS_ERR_ARG;   // the record bins are per (level, bin) of ONE table
    if (lay.schedule == 1 && (L % 8u) != 0u) lay.schedule = 0;
    if (lay.out_bf16 && (C != 2 || D != 3 || dy_dx || !pair_forward())) return HS_ERR_ARG;     // the packed-word output: the pair kernel's value form only
    const uint32_t n_chunks = (B + kFwdThreads - 1) / kFwdThreads;
    const LevelScales sc = make_scales(L, S, H);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(n_chunks * L), block(kFwdThreads);
    dispatch_dc(D, C, [&](auto d, auto c) {
        constexpr int D_ = decltype(d)::value, C_ = decltype(c)::value;
        if (D_ == 3 && pair_forward()) {
            const uint32_t n_chunks2 = (2 * B + kFwdThreads - 1) / kFwdThreads;      // two lanes per point
            if (dy_dx)
                k_hash_fwd_pair<C_, true><<<dim3(n_chunks2 * L), block, HS_FWD_LDS_PAD, st>>>(inputs, embeddings, offsets, outputs, dy_dx, B, L, sc, lay, n_chunks2);
            else
                k_hash_fwd_pair<C_, false><<<dim3(n_chunks2 * L), block, HS_FWD_LDS_PAD, st>>>(inputs, embeddings, offsets, outputs, nullptr, B, L, sc, lay, n_chunks2);
        } else if (dy_dx)
            k_hash_fwd<D_, C_, true><<<grid, block, 0, st>>>(inputs, embeddings, offsets, outputs, dy_dx, B, L, sc, lay, n_chunks);
        else
            k_hash_fwd<D_, C_, false><<<grid, block, 0, st>>>(inputs, embeddings, offsets, outputs, dy_dx, B, L, sc, lay, n_chunks);
    });
    return check_launch();
}

int hs_hash_bwd(const float *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, uint32_t B, uint32_t D,
                uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx, float *grad_inputs, const hsHashLayout *layout,
                void *stream) {
    return hs_hash_bwd_draw(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, layout, nullptr, 0, 0, 0, nullptr, nullptr, 0, nullptr, 0,
                            stream);
}

int hs_hash_bwd_draw(const float *grad, const float *inputs, const int32_t *offsets, float *grad_embeddings, uint32_t B, uint32_t D,
                     uint32_t C, uint32_t L, float S, uint32_t H, const float *dy_dx, float *grad_inputs, const hsHashLayout *layout,
                     const hsDrawSched *draw, int32_t n_uniform, int32_t total_pixels, int32_t n_out, int64_t *draw_out, const hsGatherJob *gather,
                     int32_t n_gather, const hsAsmJob *sums, int32_t n_sums, void *stream) {
    if (!dims_ok(D, C, L)) return HS_ERR_ARG;
    Riders riders;
    DrawRider &rider = riders.draw;
    rider.blocks = 0;
    riders.sum_blocks = 0;
    if (n_sums > 0) {   /* hs_assemble's jobs; they need a scatter launch to ride in */
        if (B == 0 || !grad_embeddings) return HS_ERR_ARG;
        const int rc = fill_asm_jobs(sums, n_sums, riders.sums);
        if (rc != HS_OK) return rc;
        riders.sum_blocks = riders.sums.first[n_sums];
    }
    if (draw) {     /* hs_draw_gather_sched's arguments and checks; the draw needs a scatter launch to ride in */
        if (B == 0 || !grad_embeddings) return HS_ERR_ARG;
        if (!draw->frames || !draw->sched || !draw->cursor || !draw_out) return HS_ERR_NULL;
        if (draw->n_sched < 1 || draw->n_frames < 1 || n_uniform < 0 || total_pixels < 1 || n_out < 0 || n_gather < 0 || n_gather > HS_GATHER_MAX_JOBS) return HS_ERR_ARG;
        if (n_gather > 0 && !gather) return HS_ERR_NULL;
        rider.draw = *draw; rider.n_uniform = n_uniform; rider.total_pixels = total_pixels; rider.n_out = n_out; rider.out = draw_out;
        rider.jobs.n = n_gather;
        int64_t most = n_out;
        for (int i = 0; i < n_gather; i++) {
            const hsGatherJob &j = gather[i];
            if (j.n < 0 || j.row_bytes < 0 || (j.row_bytes & 3)) return HS_ERR_ARG;
            if (j.n > 0 && !j.dst) return HS_ERR_NULL;
            if (j.idx == draw_out && j.n != n_out) return HS_ERR_ARG;
            rider.jobs.j[i] = j;
            most = j.n > most ? j.n : most;
        }
        rider.blocks = (int32_t)((most + kDrawThreads - 1) / kDrawThreads);
    }
    if (layout && layout->step && (B == 0 || !grad_embeddings)) {      // nothing to scatter: the table still takes its step
        if (!offsets || !step_ok(*layout, grad_embeddings)) return HS_ERR_NULL;
        hsHashLayout none = *layout;
        none.scatter_ws = nullptr;
        dispatch_dc(D, C, [&](auto d, auto c) {
            launch_bin_reduce<decltype(d)::value, decltype(c)::value>(grad_embeddings, offsets, L, make_scales(L, S, H), none, (hipStream_t)stream);
        });
        if (B == 0) return check_launch();
    }
    if (B == 0) return HS_OK;
    if (!grad || !inputs || !offsets || !layout) return HS_ERR_NULL;
    if (grad_inputs && !dy_dx) return HS_ERR_NULL;
    if (!step_ok(*layout, grad_embeddings)) return HS_ERR_NULL;
    hsHashLayout lay = *layout;
    if (lay.grid_id && (lay.scatter_ws || lay.grid_stride <= 0)) return HS_ERR_ARG;   // the record bins are per (level, bin) of ONE table
    if (lay.schedule == 1 && (L % 8u) != 0u) lay.schedule = 0;
    const uint32_t n_chunks = (B + kThreads - 1) / kThreads;
    hipStream_t st = (hipStream_t)stream;
    if (grad_embeddings) {
        const LevelScales sc = make_scales(L, S, H);
        apply_bin_dense_switch();
    if (lay.scatter_ws && !lay.ws_clean) k_zero_u32<<<(HS_MAX_LEVELS * kBins + 255) / 256, 256, 0, st>>>((uint32_t *)lay.scatter_ws, HS_MAX_LEVELS * kBins);
        dispatch_dc(D, C, [&](auto d, auto c) {
            if (rider.blocks + riders.sum_blocks > 0) {
                const uint32_t first = ((uint32_t)(rider.blocks + riders.sum_blocks) + 7u) & ~7u;
                k_hash_bwd_scatter_draw<decltype(d)::value, decltype(c)::value><<<dim3(first + n_chunks * L), dim3(kThreads), 0, st>>>(
                    grad, inputs, offsets, grad_embeddings, B, L, sc, lay, n_chunks, riders, first);
            } else {
                k_hash_bwd_scatter<decltype(d)::value, decltype(c)::value><<<dim3(n_chunks * L), dim3(kThreads), 0, st>>>(
                    grad, inputs, offsets, grad_embeddings, B, L, sc, lay, n_chunks);
            }
            launch_bin_reduce<decltype(d)::value, decltype(c)::value>(grad_embeddings, offsets, L, sc, lay, st);
        });
    }
    if (grad_inputs)
        dispatch_dc(D, C, [&](auto d, auto c) {
            k_hash_bwd_input<decltype(d)::value, decltype(c)::value><<<dim3(n_chunks), dim3(kThreads), 0, st>>>(grad, dy_dx, grad_inputs, B, L, lay);
        });
    return check_launch();
}

int hs_hash_bwd2(const float *grad, const float *inputs, const int32_t *offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                 float S, uint32_t H, const float *dy_dx, const float *grad_grad_inputs, float *grad_grad, float *grad2_embeddings,
                 const hsHashLayout *layout, void *stream) {
    if (!dims_ok(D, C, L) || C < 2) return HS_ERR_ARG;  // C == 1 unsupported, as in the reference (hashencoder.cu:678-684)
    if (B == 0 || (!grad_grad && !grad2_embeddings)) return HS_OK;
    if (!grad || !inputs || !offsets || !dy_dx || !grad_grad_inputs || !layout) return HS_ERR_NULL;
    hsHashLayout lay = *layout;
    if (lay.grid_id && (lay.scatter_ws || lay.grid_stride <= 0)) return HS_ERR_ARG;   // the record bins are per (level, bin) of ONE table
    if (lay.schedule == 1 && (L % 8u) != 0u) lay.schedule = 0;
    const uint32_t n_chunks = (B + kThreads - 1) / kThreads;
    const LevelScales sc = make_scales(L, S, H);
    hipStream_t st = (hipStream_t)stream;
    dispatch_dc(D, C, [&](auto d, auto c) {
        k_hash_bwd2<decltype(d)::value, decltype(c)::value><<<dim3(n_chunks * L), dim3(kThreads), 0, st>>>(
            grad, inputs, offsets, dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings, B, L, sc, lay, n_chunks);
    });
    return check_launch();
}

int hs_hash_bwd_jac(const float *g_feat, const float *g_dydx, const float *inputs, const int32_t *offsets, float *grad_embeddings,
                    uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const hsHashLayout *layout, void *stream) {
    return hs_hash_bwd_jac_sums(g_feat, g_dydx, inputs, offsets, grad_embeddings, B, D, C, L, S, H, layout, nullptr, 0, stream);
}

int hs_hash_bwd_jac_sums(const float *g_feat, const float *g_dydx, const float *inputs, const int32_t *offsets, float *grad_embeddings,
                         uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const hsHashLayout *layout, const hsAsmJob *sums,
                         int32_t n_sums, void *stream) {
    if (!dims_ok(D, C, L)) return HS_ERR_ARG;
    if (!layout) return HS_ERR_NULL;
    Riders riders;
    riders.draw.blocks = 0;
    riders.sum_blocks = 0;
    if (n_sums > 0) {   /* hs_assemble's jobs; they need a scatter launch to ride in */
        if (B == 0 || (!g_feat && !g_dydx && !layout->r1_ux)) return HS_ERR_ARG;
        const int rc = fill_asm_jobs(sums, n_sums, riders.sums);
        if (rc != HS_OK) return rc;
        riders.sum_blocks = riders.sums.first[n_sums];
    }
    if (B == 0 || (!g_feat && !g_dydx && !layout->r1_ux)) {
        if (!layout->step) return HS_OK;
        if (!offsets || !step_ok(*layout, grad_embeddings)) return HS_ERR_NULL;     // nothing to scatter: the table still takes its step
        hsHashLayout none = *layout;
        none.scatter_ws = nullptr;
        dispatch_dc(D, C, [&](auto d, auto c) {
            launch_bin_reduce<decltype(d)::value, decltype(c)::value>(grad_embeddings, offsets, L, make_scales(L, S, H), none, (hipStream_t)stream);
        });
        return check_launch();
    }
    if (!inputs || !offsets || !grad_embeddings) return HS_ERR_NULL;
    if (!step_ok(*layout, grad_embeddings)) return HS_ERR_NULL;
    hsHashLayout lay = *layout;
    if (lay.r1_ux && (!lay.r1_g || lay.r1_n > B)) return HS_ERR_ARG;
    if (lay.grid_id && (lay.scatter_ws || lay.grid_stride <= 0)) return HS_ERR_ARG;   // the record bins are per (level, bin) of ONE table
    if (lay.schedule == 1 && (L % 8u) != 0u) lay.schedule = 0;
    const uint32_t n_chunks = (B + kThreads - 1) / kThreads;
    const LevelScales sc = make_scales(L, S, H);
    hipStream_t st = (hipStream_t)stream;
    apply_bin_dense_switch();
    if (lay.scatter_ws && !lay.ws_clean) k_zero_u32<<<(HS_MAX_LEVELS * kBins + 255) / 256, 256, 0, st>>>((uint32_t *)lay.scatter_ws, HS_MAX_LEVELS * kBins);
    dispatch_dc(D, C, [&](auto d, auto c) {
        if (riders.sum_blocks > 0) {
            const uint32_t first = ((uint32_t)riders.sum_blocks + 7u) & ~7u;
            k_hash_bwd_jac_riders<decltype(d)::value, decltype(c)::value><<<dim3(first + n_chunks * L), dim3(kThreads), 0, st>>>(
                g_feat, g_dydx, inputs, offsets, grad_embeddings, B, L, sc, lay, n_chunks, riders, first);
        } else {
            k_hash_bwd_jac<decltype(d)::value, decltype(c)::value><<<dim3(n_chunks * L), dim3(kThreads), 0, st>>>(
                g_feat, g_dydx, inputs, offsets, grad_embeddings, B, L, sc, lay, n_chunks);
        }
        launch_bin_reduce<decltype(d)::value, decltype(c)::value>(grad_embeddings, offsets, L, sc, lay, st);
    });
    return check_launch();
}

int64_t hs_hash_scatter_ws_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t *cap_out) {
    if (!dims_ok(D, C, L)) return HS_ERR_ARG;
    const uint64_t per_level = (uint64_t)B << D;                           // corner contributions of one level
    const uint32_t cap = (uint32_t)(per_level / kBins * 3 / 2 + 1024);     // hashed cells are uniform over the bins: 50 % head room
    if (cap_out) *cap_out = cap;
    return (int64_t)(HS_MAX_LEVELS * kBins * sizeof(uint32_t)) + (int64_t)L * kBins * (int64_t)cap * (int64_t)(4 + 4 * C);
}

// ---- reference-compatible entry points (hashencoder/src/bindings.cpp:5-9)
int hs_hash_encode_forward(const float *inputs, const float *embeddings, const int32_t *offsets, float *outputs, uint32_t B, uint32_t D,
                           uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, float *dy_dx, void *stream) {
    if (B != 0 && calc_grad_inputs && !dy_dx) return HS_ERR_NULL;
    const hsHashLayout lay = reference_layout(B, D, C, L);
    return hs_hash_fwd(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs ? dy_dx : nullptr, &lay, stream);
}

int hs_hash_encode_backward(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets, float *grad_embeddings,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, const float *dy_dx,
                            float *grad_inputs, void *stream) {
    (void)embeddings;
    if (B != 0 && (!grad_embeddings || (calc_grad_inputs && (!dy_dx || !grad_inputs)))) return HS_ERR_NULL;
    const hsHashLayout lay = reference_layout(B, D, C, L);
    return hs_hash_bwd(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, calc_grad_inputs ? grad_inputs : nullptr, &lay, stream);
}

int hs_hash_encode_second_backward(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets, uint32_t B,
                                   uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs, const float *dy_dx,
                                   const float *grad_grad_inputs, float *grad_grad, float *grad2_embeddings, void *stream) {
    (void)embeddings;
    (void)calc_grad_inputs;
    if (B != 0 && (!grad_grad || !grad2_embeddings)) return HS_ERR_NULL;
    const hsHashLayout lay = reference_layout(B, D, C, L);
    return hs_hash_bwd2(grad, inputs, offsets, B, D, C, L, S, H, dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings, &lay, stream);
}

}  // extern "C"
