// holoscene_amd/csrc/gemm_split.hip -- fp32 matrix products on the bf16 matrix cores: every fp32 operand is split into P bf16 planes
// (x = x0 + x1 (+ x2), each plane the bf16 rounding of what the planes before it left over) and the product is the sum of the plane
// products whose combined weight is above the fp32 rounding error, accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//     P = 3: six plane products (x0 y0, x0 y1, x1 y0, x0 y2, x1 y1, x2 y0) -- 24 mantissa bits per operand, the error of an fp32 FMA chain;
//     P = 2: three (x0 y0, x0 y1, x1 y0)                                   -- 16 mantissa bits per operand (relative error ~1e-5).
// This is the reference's own precision (training/holoscene_train.py:45 runs fp32; torch.mm on fp32 operands) at 1/6 (1/3) of the bf16
// MFMA rate = 2.6x (5x) the fp32 MFMA rate (v_mfma_f32_32x32x2_f32), which is what the library's fp32 GEMMs run on.  It replaces the three
// products of model/network.py's `_linear_rows` (the fp32 path's only GEMM call site):
//     NT:  C [M, N] = A [M, K] . B [N, K]^T (+ bias [N])        forward (A = x, B = W) and data gradient (A = g, B = W^T)
//     TN:  C_s [N, K] = sum over the rows m of slice s of A [m, N]^T . B [m, K]   weight gradient as split-M partials (A = g, B = x)
// Workgroup = 4 waves = one 128 x 128 tile of C (a wave: 64 x 64, four accumulators); the reduction runs in chunks of 32 through LDS: fp32 from
// global memory (next chunk requested before this chunk's products), split in registers, planes stored as bf16 tiles; fragments by
// ds_read_b128 (NT: the reduction index is contiguous) or by gfx950's transposing ds_read_b64_tr_b16 (TN: it is the row index).
#include "launch_util.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "holoscene_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));

constexpr int kT = 256;            // threads
constexpr int BT = 128;            // tile of C: BT x BT
constexpr int RC = 32;             // reduction chunk
constexpr int PNT = RC + 8;        // NT: LDS row pitch in elements (80 B: rows 4 apart fall on different bank groups)
constexpr int PTN = BT + 24;       // TN: pitch of a [RC][BT] tile (wgrad_pairs.hip: 2-way instead of 4-way conflicts on the transposing reads)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {   // one v_cvt_pk_bf16_f32 (round-to-nearest-even): a in the low half
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}

// (x, y) -> P words of two bf16 each (plane p of x in the low half, of y in the high half); plane p + 1 is the bf16 rounding of the remainder
// after planes 0..p (the subtractions are exact)
template <int P> __device__ __forceinline__ void split2(float x, float y, uint32_t (&w)[P]) {
#pragma unroll
    for (int p = 0; p < P; p++) {
        w[p] = pack2(x, y);
        if (p + 1 < P) {
            x -= __uint_as_float(w[p] << 16);
            y -= __uint_as_float(w[p] & 0xffff0000u);
        }
    }
}

// four consecutive elements of row `r` starting at column `c` of a row-major fp32 matrix [rows, cols] (zero outside)
__device__ __forceinline__ float4 load4(const float *__restrict__ p, int64_t ld, int64_t r, int c, int64_t rows, int cols, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= rows || c >= cols) return v;
    const float *q = p + r * ld + c;
    if (vec && c + 3 < cols) return *reinterpret_cast<const float4 *>(q);
    v.x = q[0];
    if (c + 1 < cols) v.y = q[1];
    if (c + 2 < cols) v.z = q[2];
    if (c + 3 < cols) v.w = q[3];
    return v;
}

// the plane products (i, j) kept, most significant first: P = 2 -> (0,0) (0,1) (1,0);  P = 3 -> those + (0,2) (1,1) (2,0)
__host__ __device__ constexpr int term_count(int P) { return P == 2 ? 3 : 6; }
__host__ __device__ constexpr int term_a(int k) { return k == 0 ? 0 : k == 1 ? 0 : k == 2 ? 1 : k == 3 ? 0 : k == 4 ? 1 : 2; }
__host__ __device__ constexpr int term_b(int k) { return k == 0 ? 0 : k == 1 ? 1 : k == 2 ? 0 : k == 3 ? 2 : k == 4 ? 1 : 0; }

__device__ __forceinline__ bf16x8 tr_frag(uint32_t addr, uint32_t step) {
    const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4 *)(uintptr_t)addr);
    const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4 *)(uintptr_t)(addr + step));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// C tile (i0.., j0..) += sum over r of A(i, r) B(j, r).
//   NT: A [I, R] row-major (lda), B [J, R] row-major (ldb): the reduction index is the column.
//   TN: A [R, I] row-major (lda), B [R, J] row-major (ldb): the reduction index is the row, r in [r_begin, r_end).
template <int P, bool TN>
__global__ __launch_bounds__(kT) void k_gemm_split(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb, float *__restrict__ C,
                                                   int64_t ldc, const float *__restrict__ bias, int64_t I, int J, int64_t R, int64_t rows_per_slice,
                                                   int tiles_j) {
    constexpr int PITCH = TN ? PTN : PNT;
    constexpr int TILE = (TN ? RC * PTN : BT * PNT);         // elements of one plane of one operand
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *At = lds, *Bt = lds + P * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wi = wave & 1, wj = wave >> 1;
    const int bj = blockIdx.x % tiles_j, bi = blockIdx.x / tiles_j;
    const int64_t i0 = (int64_t)bi * BT;
    const int j0 = bj * BT;
    const int64_t r_begin = TN ? (int64_t)blockIdx.y * rows_per_slice : 0;
    const int64_t r_end = TN ? (r_begin + rows_per_slice < R ? r_begin + rows_per_slice : R) : R;
    const bool veca = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0), vecb = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
    // ---- staging: 128 x 32 (NT) or 32 x 128 (TN) fp32 per operand and chunk = four float4 per thread
    float4 ra[4], rb[4];
    // a chunk whose columns are all inside the matrices takes unconditional 16-byte loads (rows clamped, then zeroed): written with
    // per-element bounds the eight loads of a chunk sit in as many basic blocks, each waiting for its own result before the next is issued
    auto tile4 = [](const float *__restrict__ p, int64_t ld, int64_t r, int64_t rows, int c, int cols) {       // cols % 4 == 0: c < cols => c + 3 < cols
        const bool in = r < rows && c < cols;
        const float4 v = *reinterpret_cast<const float4 *>(p + (r < rows ? r : rows - 1) * ld + (c < cols ? c : 0));
        return in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const bool fast = veca && vecb && (TN ? (I % 4 == 0 && J % 4 == 0) : R % 4 == 0);       // workgroup-uniform
    auto fetch = [&](int64_t r0) {
        if (fast) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int idx = tid + kT * q;
                if constexpr (TN) {
                    const int rr = idx >> 5, c4 = (idx & 31) * 4;
                    ra[q] = tile4(A, lda, r0 + rr, r_end, (int)i0 + c4, (int)I);
                    rb[q] = tile4(B, ldb, r0 + rr, r_end, j0 + c4, J);
                } else {
                    const int rr = idx >> 3, c4 = (idx & 7) * 4;
                    ra[q] = tile4(A, lda, i0 + rr, I, (int)r0 + c4, (int)R);
                    rb[q] = tile4(B, ldb, j0 + rr, J, (int)r0 + c4, (int)R);
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = tid + kT * q;
            if constexpr (TN) {
                const int rr = idx >> 5, c4 = (idx & 31) * 4;
                ra[q] = load4(A, lda, r0 + rr < r_end ? r0 + rr : R, (int)i0 + c4, R, (int)I, veca);
                rb[q] = load4(B, ldb, r0 + rr < r_end ? r0 + rr : R, j0 + c4, R, J, vecb);
            } else {
                const int rr = idx >> 3, c4 = (idx & 7) * 4;
                ra[q] = load4(A, lda, i0 + rr, (int)r0 + c4, I, (int)R, veca);
                rb[q] = load4(B, ldb, j0 + rr, (int)r0 + c4, J, (int)R, vecb);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = tid + kT * q;
            const int rr = TN ? idx >> 5 : idx >> 3, c4 = TN ? (idx & 31) * 4 : (idx & 7) * 4;
            const float va[4] = {ra[q].x, ra[q].y, ra[q].z, ra[q].w}, vb[4] = {rb[q].x, rb[q].y, rb[q].z, rb[q].w};
            uint32_t pa[2][P], pb[2][P];
            split2<P>(va[0], va[1], pa[0]); split2<P>(va[2], va[3], pa[1]);
            split2<P>(vb[0], vb[1], pb[0]); split2<P>(vb[2], vb[3], pb[1]);
#pragma unroll
            for (int p = 0; p < P; p++) {
                *reinterpret_cast<uint2 *>(At + p * TILE + rr * PITCH + c4) = make_uint2(pa[0][p], pa[1][p]);
                *reinterpret_cast<uint2 *>(Bt + p * TILE + rr * PITCH + c4) = make_uint2(pb[0][p], pb[1][p]);
            }
        }
    };
    // ---- fragment addresses (bytes, plane 0, this wave's first 32-row tile, k-step 0)
    const int L16 = lane & 15, cg = (lane >> 4) & 1, rg = lane >> 5;
    const uint32_t a_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)At, b_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)Bt;
    const uint32_t fa = TN ? (uint32_t)(((rg * 8 + (L16 >> 2)) * PTN + wi * 64 + 16 * cg + 4 * (L16 & 3)) * 2) : (uint32_t)(((wi * 64 + (lane & 31)) * PNT + 8 * rg) * 2);
    const uint32_t fb = TN ? (uint32_t)(((rg * 8 + (L16 >> 2)) * PTN + wj * 64 + 16 * cg + 4 * (L16 & 3)) * 2) : (uint32_t)(((wj * 64 + (lane & 31)) * PNT + 8 * rg) * 2);
    auto frag = [&](uint32_t base, uint32_t lane_off, int plane, int tile, int ks) -> bf16x8 {
        if constexpr (TN) return tr_frag(base + lane_off + (uint32_t)((plane * TILE + ks * 16 * PTN + tile * 32) * 2), 4 * PTN * 2);
        else return *(__attribute__((address_space(3))) const bf16x8 *)(uintptr_t)(base + lane_off + (uint32_t)((plane * TILE + tile * 32 * PNT + ks * 16) * 2));
    };
    const int64_t nchunks = r_end > r_begin ? (r_end - r_begin + RC - 1) / RC : 0;
    if (nchunks > 0) fetch(r_begin);
    for (int64_t c = 0; c < nchunks; c++) {
        __syncthreads();            // everybody is done with the previous chunk's tiles
        stage();
        __syncthreads();
        if (c + 1 < nchunks) fetch(r_begin + (c + 1) * RC);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 af[2][P], bfr[2][P];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int p = 0; p < P; p++) { af[t][p] = frag(a_base, fa, p, t, ks); bfr[t][p] = frag(b_base, fb, p, t, ks); }
#pragma unroll
            for (int k = term_count(P) - 1; k >= 0; k--)
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][term_a(k)], bfr[b][term_b(k)], acc[a][b], 0, 0, 0);
        }
    }
    // ---- C: register r of accumulator (a, b) <-> row i0 + 64 wi + 32 a + 8 (r >> 2) + 4 (lane >> 5) + (r & 3), column j0 + 64 wj + 32 b + (lane & 31)
    float *Cs = C + (TN ? (size_t)blockIdx.y * (size_t)I * ldc : 0);
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const int j = j0 + 64 * wj + 32 * b + (lane & 31);
        if (j >= J) continue;
        const float bv = bias ? bias[j] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t i = i0 + 64 * wi + 32 * a + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (i < I) Cs[i * ldc + j] = acc[a][b][r] + bv;
            }
    }
}

template <int P, bool TN> int launch(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias, int64_t I, int J, int64_t R,
                                     int slices, hipStream_t stream) {
    const size_t lds = 2 * (size_t)P * (TN ? RC * PTN : BT * PNT) * sizeof(uint16_t);
    static hsLdsAttrOnce attr;
    attr.set((const void *)k_gemm_split<P, TN>, (int)lds);
    const int tiles_j = (J + BT - 1) / BT;
    const int64_t tiles_i = (I + BT - 1) / BT;
    if (tiles_i * tiles_j > 0x7fffffff) return HS_ERR_ARG;
    const int64_t per = TN ? ((R + slices - 1) / slices + RC - 1) / RC * RC : 0;
    k_gemm_split<P, TN><<<dim3((unsigned)(tiles_i * tiles_j), TN ? slices : 1), kT, lds, stream>>>(A, lda, B, ldb, C, ldc, bias, I, J, R, per, tiles_j);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int hs_gemm_split_nt(const float *A, int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias, int64_t M, int32_t N, int32_t K,
                     int32_t planes, void *stream) {
    if (M < 0 || N < 0 || K < 0 || lda < K || ldb < K || ldc < N || (planes != 2 && planes != 3)) return HS_ERR_ARG;
    if (M == 0 || N == 0) return HS_OK;
    if (!A || !B || !C) return HS_ERR_NULL;
    return planes == 3 ? launch<3, false>(A, lda, B, ldb, C, ldc, bias, M, N, K, 1, (hipStream_t)stream)
                       : launch<2, false>(A, lda, B, ldb, C, ldc, bias, M, N, K, 1, (hipStream_t)stream);
}

int hs_gemm_split_tn(const float *A, int64_t lda, const float *B, int64_t ldb, float *C_parts, int64_t M, int32_t N, int32_t K, int32_t slices, int32_t planes,
                     void *stream) {
    if (M < 0 || N < 0 || K < 0 || lda < N || ldb < K || slices < 1 || (planes != 2 && planes != 3)) return HS_ERR_ARG;
    if (N == 0 || K == 0) return HS_OK;
    if (!A || !B || !C_parts) return HS_ERR_NULL;
    return planes == 3 ? launch<3, true>(A, lda, B, ldb, C_parts, K, nullptr, N, K, M, slices, (hipStream_t)stream)
                       : launch<2, true>(A, lda, B, ldb, C_parts, K, nullptr, N, K, M, slices, (hipStream_t)stream);
}

}  // extern "C"
