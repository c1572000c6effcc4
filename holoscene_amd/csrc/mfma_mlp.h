// holoscene_amd/csrc/mfma_mlp.h -- shared device machinery of the fused matrix-core MLP kernels (sdf_mlp.hip, appearance_mlp.hip).
//
// One workgroup = 512 threads = 8 waves owns a tile of BM = 128 rows and walks it through 256-wide layers without leaving
// the CU.  The product is formed transposed, D[neuron][row] = W . H^T (v_mfma_f32_32x32x16_bf16, fp32 accumulate), so both
// operands are 16-byte row reads from row-major LDS images -- weights [neuron][k] streamed L2 -> registers -> LDS in
// double-buffered chunks of 32 k, activations [row][k] resident in a tile that every layer rewrites in place -- and each
// lane ends up with 4 consecutive neurons of ONE row per accumulator quad.  wave = (neuron quarter nq, row half ph):
// 64 neurons x 64 rows, 2 x 2 accumulator tiles.  Row pitches carry +8 bf16 so every ds_read_b128 is conflict-free.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifndef HS_MLP_BM
#define HS_MLP_BM 128
#endif
constexpr int BM = HS_MLP_BM;     // rows per tile: 128 (8 waves, one workgroup per CU) or 64 (4 waves, two workgroups per CU)
constexpr int kThreads = 4 * BM;
constexpr int kRowWaves = BM / 32;  // waves that take part in a narrow (<= 32 outputs) last layer: 32 rows each
constexpr int kGridCap = 256 * (128 / BM);
constexpr int HID = 256;          // hidden width
constexpr int HP = HID + 8;       // activation row pitch (bf16)
#ifndef HS_MLP_KC
#define HS_MLP_KC 32
#endif
constexpr int KC = HS_MLP_KC;     // (largest) weight chunk depth: sizes the two chunk buffers.  64 halves the number of barrier rounds per
                                  // 256-deep layer; kernels that need the LDS for something else stay at 32
constexpr int WP = KC + 8;        // row pitch (bf16) of a KC-deep chunk; buffers are HID * WP apart whatever depth a layer streams with
constexpr int K0 = 96;            // padded input width (71 -> 96)

__device__ __forceinline__ uint32_t f2bf(float f) {  // round-to-nearest-even; inputs are finite here
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // one v_cvt_pk_bf16_f32 (round-to-nearest-even)
    const float2_t v = {a, b};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<const uint32_t *>(&r);
}

// rows [0,256) x cols [k0, k0+KCH) of a row-major [256][ldw] bf16 matrix, spread over the workgroup
constexpr int kChunkTPR = kThreads / HID;            // threads per weight row: 2 (512 threads) or 1 (256 threads)
template <int KCH> struct ChunkRegs { uint4 v[KCH / 8 / kChunkTPR]; };

// NR < HID: only the first NR rows are streamed (the rest of the matrix is padding nobody multiplies by; their LDS rows keep stale data)
template <int KCH, int NR = HID>
__device__ __forceinline__ ChunkRegs<KCH> load_chunk(const uint16_t *__restrict__ W, int ldw, int k0) {
    ChunkRegs<KCH> r;
    constexpr int NV = KCH / 8 / kChunkTPR;
    const int row = threadIdx.x / kChunkTPR, part = threadIdx.x % kChunkTPR;
    const uint16_t *src = W + (size_t)row * ldw + k0 + part * (8 * NV);
    if (NR < HID && row >= NR) return r;
#pragma unroll
    for (int i = 0; i < NV; i++) r.v[i] = *reinterpret_cast<const uint4 *>(src + 8 * i);
    return r;
}

template <int KCH, int NR = HID>
__device__ __forceinline__ void store_chunk(uint16_t *Wc, const ChunkRegs<KCH> &r) {
    constexpr int NV = KCH / 8 / kChunkTPR;
    const int row = threadIdx.x / kChunkTPR, part = threadIdx.x % kChunkTPR;
    uint16_t *dst = Wc + (size_t)row * (KCH + 8) + part * (8 * NV);
    if (NR < HID && row >= NR) return;
#pragma unroll
    for (int i = 0; i < NV; i++) *reinterpret_cast<uint4 *>(dst + 8 * i) = r.v[i];
}

struct Frags { bf16x8 a[2], b[2]; };

// operand fragments of k-step `s` (16 wide) of a layer streamed in KCH-deep chunks: weights from the chunk buffer, activations
// from H (row pitch AP)
template <int AP, int KCH>
__device__ __forceinline__ Frags load_frags(const uint16_t *Wc, const uint16_t *H, int s, int nq, int ph, int lane) {
    Frags f;
    constexpr int KS = KCH / 16;
    const uint16_t *wbuf = Wc + (size_t)((s / KS) & 1) * HID * WP + (s % KS) * 16 + (lane >> 5) * 8;
    const uint16_t *hrow = H + s * 16 + (lane >> 5) * 8;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        f.a[i] = *reinterpret_cast<const bf16x8 *>(wbuf + (size_t)(nq * 64 + i * 32 + (lane & 31)) * (KCH + 8));
        f.b[i] = *reinterpret_cast<const bf16x8 *>(hrow + (size_t)(ph * 64 + i * 32 + (lane & 31)) * AP);
    }
    return f;
}

// (the one-chunk-ahead form of layer_mma below, kept verbatim for the kernel that has no registers for a second chunk set)
template <int AP, int KCH, int NR>
__device__ __forceinline__ void layer_mma_shallow(const uint16_t *__restrict__ W, int ldw, int K, const uint16_t *H, uint16_t *Wc, f32x16 acc[2][2],
                                          int nq, int ph, int lane, bool compute = true) {
    constexpr int KS = KCH / 16;
    const int nchunks = K / KCH;
    ChunkRegs<KCH> pre = load_chunk<KCH, NR>(W, ldw, 0);
    store_chunk<KCH, NR>(Wc, pre);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
#ifndef HS_EXP_NO_WLOAD
        if (c + 1 < nchunks) pre = load_chunk<KCH, NR>(W, ldw, (c + 1) * KCH);
#endif
#ifndef HS_EXP_NO_MMA
        if (compute) {
            Frags f[2];
            f[0] = load_frags<AP, KCH>(Wc, H, c * KS, nq, ph, lane);
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
                if (kk + 1 < KS) f[(kk + 1) & 1] = load_frags<AP, KCH>(Wc, H, c * KS + kk + 1, nq, ph, lane);   // in flight under this step's MFMAs
                const Frags &g = f[kk & 1];
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int pt = 0; pt < 2; pt++) acc[nt][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.a[nt], g.b[pt], acc[nt][pt], 0, 0, 0);
            }
        }
#endif
        if (c + 1 < nchunks) store_chunk<KCH, NR>(Wc + (size_t)((c + 1) & 1) * HID * WP, pre);
        __syncthreads();
    }
}

// One hidden layer: acc[nt][pt] += W[neurons][K] . H[points][K]^T, K a multiple of KCH.  wave -> neurons [nq*64,+64), points [ph*64,+64)
// compute = false: the wave only helps stream the weights and keeps the barriers (its neurons are padding)
// first: the layer's chunk 0 if the caller requested it earlier (first_chunk below, issued before the previous layer's epilogue): every
// layer otherwise opens with one fully exposed L2 latency (~2.5 k cycles) before its first MFMA
template <int AP = HP, int KCH = KC, int NR = HID, bool DEEP = true>
__device__ __forceinline__ void layer_mma(const uint16_t *__restrict__ W, int ldw, int K, const uint16_t *H, uint16_t *Wc, f32x16 acc[2][2],
                                          int nq, int ph, int lane, bool compute = true, const ChunkRegs<KCH> *first = nullptr) {
    if constexpr (!DEEP) { layer_mma_shallow<AP, KCH, NR>(W, ldw, K, H, Wc, acc, nq, ph, lane, compute); return; }
    constexpr int KS = KCH / 16;
    const int nchunks = K / KCH;
    // Weight chunks travel TWO rounds ahead of their use (two register sets, chunk k in set k & 1): a round is as long as the L2
    // latency of the chunk it waits for (~2.5 k cycles on the loaded chip against 0.5 k of MFMA per 64-deep chunk), so with one chunk
    // in flight a 256-deep layer costs four latencies (tools/exp/tbwd_prof.hip: 10 k cycles for 2 k of MFMA).
    //   round c:  request chunk c + 2 (its set held chunk c, stored to LDS a round ago) | MFMAs of chunk c from buffer c & 1 |
    //             store chunk c + 1 (requested a round ago) to buffer (c + 1) & 1 -- last read in round c - 1, before its barrier | barrier
    // (two named register sets and the loop unrolled by hand in pairs: indexed as pre[c & 1] the sets live in scratch memory)
    auto compute_chunk = [&](int c) {
#ifndef HS_EXP_NO_MMA
        if (compute) {
            Frags f[2];
            f[0] = load_frags<AP, KCH>(Wc, H, c * KS, nq, ph, lane);
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
                if (kk + 1 < KS) f[(kk + 1) & 1] = load_frags<AP, KCH>(Wc, H, c * KS + kk + 1, nq, ph, lane);   // in flight under this step's MFMAs
                const Frags &g = f[kk & 1];
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int pt = 0; pt < 2; pt++) acc[nt][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.a[nt], g.b[pt], acc[nt][pt], 0, 0, 0);
            }
        }
#endif
    };
    uint16_t *buf0 = Wc, *buf1 = Wc + (size_t)HID * WP;
    ChunkRegs<KCH> pre0 = first ? *first : load_chunk<KCH, NR>(W, ldw, 0), pre1;
    if (nchunks > 1) pre1 = load_chunk<KCH, NR>(W, ldw, KCH);
    store_chunk<KCH, NR>(buf0, pre0);
    __syncthreads();
    for (int c = 0; c < nchunks; c += 2) {
        if (c + 2 < nchunks) pre0 = load_chunk<KCH, NR>(W, ldw, (c + 2) * KCH);
        compute_chunk(c);
        if (c + 1 < nchunks) store_chunk<KCH, NR>(buf1, pre1);
        __syncthreads();
        if (c + 1 >= nchunks) break;
        if (c + 3 < nchunks) pre1 = load_chunk<KCH, NR>(W, ldw, (c + 3) * KCH);
        compute_chunk(c + 1);
        if (c + 2 < nchunks) store_chunk<KCH, NR>(buf0, pre0);
        __syncthreads();
    }
}

template <int KCH = KC, int NR = HID>
__device__ __forceinline__ ChunkRegs<KCH> first_chunk(const uint16_t *__restrict__ W, int ldw) { return load_chunk<KCH, NR>(W, ldw, 0); }

__device__ __forceinline__ void zero_acc(f32x16 acc[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
}


__device__ __forceinline__ float quad_bcast0(float v) {  // value held by lane (lane & ~3) of each quad
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, true));
}


template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }

struct TileRegs { uint4 v[BM * (HID / 8) / kThreads]; };   // one [BM][HID] bf16 tile spread over the workgroup (8 x 16 B per thread)

__device__ __forceinline__ TileRegs load_tile_regs(const uint16_t *__restrict__ src, int64_t r0, int64_t M) {
    TileRegs t;
#pragma unroll
    for (int i = 0; i < BM * (HID / 8) / kThreads; i++) {
        const int idx = threadIdx.x + i * kThreads, row = idx / (HID / 8), seg = idx - row * (HID / 8);
        t.v[i] = r0 + row < M ? *reinterpret_cast<const uint4 *>(src + (size_t)(r0 + row) * HID + seg * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
    return t;
}

__device__ __forceinline__ void store_tile_regs(uint16_t *H, const TileRegs &t) {
#pragma unroll
    for (int i = 0; i < BM * (HID / 8) / kThreads; i++) {
        const int idx = threadIdx.x + i * kThreads, row = idx / (HID / 8), seg = idx - row * (HID / 8);
        *reinterpret_cast<uint4 *>(H + (size_t)row * HP + seg * 8) = t.v[i];
    }
}


#ifndef HS_NT_TILE_STORE
#define HS_NT_TILE_STORE 0
#endif
// activation tile -> global, 16 B per lane, rows contiguous (coalesced 512 B per row)
__device__ __forceinline__ void store_tile(const uint16_t *H, uint16_t *__restrict__ dst, int64_t r0, int64_t M) {
#ifdef HS_EXP_NO_STORE
    if (r0 >= 0) return;
#endif
    for (int idx = threadIdx.x; idx < BM * (HID / 8); idx += kThreads) {
        const int row = idx / (HID / 8), seg = idx - row * (HID / 8);
        if (r0 + row < M) {
#if HS_NT_TILE_STORE       // non-temporal: the tile's reader is another kernel, far away
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(*reinterpret_cast<const u32x4_t *>(H + (size_t)row * HP + seg * 8), reinterpret_cast<u32x4_t *>(dst + (size_t)(r0 + row) * HID + seg * 8));
#else
            *reinterpret_cast<uint4 *>(dst + (size_t)(r0 + row) * HID + seg * 8) = *reinterpret_cast<const uint4 *>(H + (size_t)row * HP + seg * 8);
#endif
        }
    }
}


// bias gradient: column sums of the cotangent tile over every STRIDE-th row (4: the value rows of the 4-row trunk; 1: all rows)
template <int STRIDE>
__device__ __forceinline__ float tile_colsum(const uint16_t *H) {
    float s = 0.f;
    if (threadIdx.x < HID) {
#pragma unroll 8
        for (int r = 0; r < BM; r += STRIDE) s += __uint_as_float((uint32_t)H[(size_t)r * HP + threadIdx.x] << 16);
    }
    return s;
}


}  // namespace
