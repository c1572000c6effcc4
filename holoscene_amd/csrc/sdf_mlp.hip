// holoscene_amd/csrc/sdf_mlp.hip -- fused SDF-trunk inference on the CDNA4 matrix cores (gfx950).
//
// The error-bounded sampler evaluates the scene SDF at up to 5 x 128 points per ray per iteration -- 85 % of all
// network evaluations of a Stage-1 step.  The reference runs ObjectImplicitNetworkGrid.forward for
// that (model/network.py:169-210, 305-311): ~20 ATen kernels per sweep, colour branch included.  This kernel does
// the whole SDF branch for a tile of 128 points without leaving the CU:
//
//   in  = [x, sin/cos(2^k x) k<6, hash features(32)]            71 values, zero-padded to 96   (model/embedder.py:11-36)
//   h0  = softplus100(W0 in + b0)   256                          (network.py:203-206, Softplus(beta=100) :163)
//   h1  = softplus100(W1 h0 + b1)   256
//   y   = W2 h1 + b2                d_out (<= 64)
//   out = min_k y_k  (or y_idx)                                   (network.py:305-311 / 316-318)
//
// bf16 operands, fp32 accumulation: v_mfma_f32_32x32x16_bf16.  The product is formed transposed, D[neuron][point] =
// W . H^T, so both operands are 16-byte row reads from row-major LDS images (weights [neuron][k], activations
// [point][k]) and each lane ends up with 4 consecutive neurons of ONE point per accumulator quad -> the epilogue
// (bias, softplus, bf16 pack) writes 8-byte runs straight back into the activation tile.
//   LDS: activation tile 128 x (256+8) bf16 = 66 KB (in place across layers) + double-buffered weight chunks
//   2 x 256 x (32+8) bf16 = 40 KB; row pitches of +8 bf16 make every ds_read_b128 conflict-free.
//   4 waves = 2 (neuron halves) x 2 (point halves): 64 points x 128 neurons per wave, 8 accumulator tiles (128 VGPR);
//   weight chunks stream L2 -> registers -> LDS one chunk ahead of the MFMAs.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>

#include "holoscene_hip.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kThreads = 256;
constexpr int BM = 128;           // points per tile
constexpr int HID = 256;          // hidden width
constexpr int HP = HID + 8;       // activation row pitch (bf16)
constexpr int KC = 32;            // weight chunk depth
constexpr int WP = KC + 8;        // weight chunk row pitch (bf16)
constexpr int K0 = 96;            // padded input width (71 -> 96)
constexpr int NFREQ = 6;
constexpr int NPE = 3 + 6 * NFREQ;  // 39
constexpr int NFEAT = 32;

__device__ __forceinline__ uint16_t f2bf(float f) {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__device__ __forceinline__ float softplus100(float v) {
    const float t = v * 100.f;
    if (t > 20.f) return v;
    return __logf(1.f + __expf(t)) * 0.01f;
}

// stage rows [0,256) x cols [k0, k0+KC) of a row-major [256][ldw] bf16 matrix: 16 KB = 64 B per thread
struct ChunkRegs { uint4 v[4]; };

__device__ __forceinline__ ChunkRegs load_chunk(const uint16_t *__restrict__ W, int ldw, int k0) {
    ChunkRegs r;
    // thread t -> row t (256 rows), 4 x 16 B = the row's 32 bf16
    const uint16_t *src = W + (size_t)threadIdx.x * ldw + k0;
#pragma unroll
    for (int i = 0; i < 4; i++) r.v[i] = *reinterpret_cast<const uint4 *>(src + 8 * i);
    return r;
}

__device__ __forceinline__ void store_chunk(uint16_t *Wc, const ChunkRegs &r) {
    uint16_t *dst = Wc + (size_t)threadIdx.x * WP;
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<uint4 *>(dst + 8 * i) = r.v[i];
}

// One hidden layer: acc[nt][pt] += W[neurons][K] . H[points][K]^T over K (multiple of KC), all 256 neurons, BM points.
// wave -> neurons [nh*128, +128), points [ph*64, +64)
__device__ __forceinline__ void layer_mma(const uint16_t *__restrict__ W, int ldw, int K, const uint16_t *H, uint16_t *Wc, f32x16 acc[4][2],
                                          int nh, int ph, int lane) {
    const int nchunks = K / KC;
    ChunkRegs pre = load_chunk(W, ldw, 0);
    store_chunk(Wc, pre);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        uint16_t *cur = Wc + (size_t)(c & 1) * HID * WP;
        uint16_t *nxt = Wc + (size_t)((c + 1) & 1) * HID * WP;
        if (c + 1 < nchunks) pre = load_chunk(W, ldw, (c + 1) * KC);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ks++) {
            bf16x8 b[2];
#pragma unroll
            for (int pt = 0; pt < 2; pt++)
                b[pt] = *reinterpret_cast<const bf16x8 *>(H + (size_t)(ph * 64 + pt * 32 + (lane & 31)) * HP + c * KC + ks * 16 + (lane >> 5) * 8);
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(cur + (size_t)(nh * 128 + nt * 32 + (lane & 31)) * WP + ks * 16 + (lane >> 5) * 8);
#pragma unroll
                for (int pt = 0; pt < 2; pt++) acc[nt][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[pt], acc[nt][pt], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) store_chunk(nxt, pre);
        __syncthreads();
    }
}

// bias + softplus + bf16 pack, written back into the activation tile (all waves have passed the barrier that ends layer_mma)
__device__ __forceinline__ void epilogue_softplus(const float *__restrict__ bias, uint16_t *H, f32x16 acc[4][2], int nh, int ph, int lane) {
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int n0 = nh * 128 + nt * 32 + q * 8 + 4 * (lane >> 5);  // 4 consecutive neurons
            const float4 bi = *reinterpret_cast<const float4 *>(bias + n0);
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                const int p = ph * 64 + pt * 32 + (lane & 31);
                const float v0 = softplus100(acc[nt][pt][q * 4 + 0] + bi.x), v1 = softplus100(acc[nt][pt][q * 4 + 1] + bi.y);
                const float v2 = softplus100(acc[nt][pt][q * 4 + 2] + bi.z), v3 = softplus100(acc[nt][pt][q * 4 + 3] + bi.w);
                uint2 pk;
                pk.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                pk.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
                *reinterpret_cast<uint2 *>(H + (size_t)p * HP + n0) = pk;
            }
        }
    }
}

template <int NOUT_TILES>  // d_out padded to 32 * NOUT_TILES
__global__ __launch_bounds__(kThreads) void k_sdf_mlp(const float *__restrict__ x, const float *__restrict__ feat, const uint16_t *__restrict__ W0,
                                                       const float *__restrict__ b0, const uint16_t *__restrict__ W1,
                                                       const float *__restrict__ b1, const uint16_t *__restrict__ W2,
                                                       const float *__restrict__ b2, int d_out, int select, float *__restrict__ out_min,
                                                       float *__restrict__ out_raw, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t *H = lds;                      // [BM][HP]
    uint16_t *Wc = lds + (size_t)BM * HP;   // 2 x [HID][WP]  (also holds W2 [32*NOUT_TILES][HP] for the last layer)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nh = wave & 1, ph = wave >> 1;
    const int64_t ntiles = (B + BM - 1) / BM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t p0 = tile * BM;
        // ---- input features: H[p][0..K0)
        for (int idx = threadIdx.x; idx < BM * K0; idx += kThreads) {
            const int p = idx / K0, c = idx - p * K0;
            const int64_t gp = p0 + p;
            float v = 0.f;
            if (gp < B) {
                if (c < 3) v = x[gp * 3 + c];
                else if (c < NPE) {
                    const int k = (c - 3) / 6, r = (c - 3) - 6 * k;
                    const float a = x[gp * 3 + (r % 3)] * (float)(1 << k);
                    v = (r < 3) ? sinf(a) : cosf(a);
                } else if (c < NPE + NFEAT) v = feat[gp * NFEAT + (c - NPE)];
            }
            H[(size_t)p * HP + c] = f2bf(v);
        }
        __syncthreads();
        f32x16 acc[4][2];
        // ---- layer 0
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
        layer_mma(W0, K0, K0, H, Wc, acc, nh, ph, lane);
        epilogue_softplus(b0, H, acc, nh, ph, lane);
        __syncthreads();
        // ---- layer 1
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
        layer_mma(W1, HID, HID, H, Wc, acc, nh, ph, lane);
        epilogue_softplus(b1, H, acc, nh, ph, lane);
        // ---- layer 2: stage W2 [32*NOUT_TILES][HID] with pitch HP into the chunk area
        for (int idx = threadIdx.x; idx < 32 * NOUT_TILES * (HID / 8); idx += kThreads) {
            const int row = idx / (HID / 8), seg = idx - row * (HID / 8);
            *reinterpret_cast<uint4 *>(Wc + (size_t)row * HP + seg * 8) = *reinterpret_cast<const uint4 *>(W2 + (size_t)row * HID + seg * 8);
        }
        __syncthreads();
        f32x16 y[NOUT_TILES];
#pragma unroll
        for (int t = 0; t < NOUT_TILES; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) y[t][i] = 0.f;
        const int prow = wave * 32 + (lane & 31);  // this wave's 32 points
#pragma unroll 4
        for (int ks = 0; ks < HID / 16; ks++) {
            const bf16x8 b = *reinterpret_cast<const bf16x8 *>(H + (size_t)prow * HP + ks * 16 + (lane >> 5) * 8);
#pragma unroll
            for (int t = 0; t < NOUT_TILES; t++) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(Wc + (size_t)(t * 32 + (lane & 31)) * HP + ks * 16 + (lane >> 5) * 8);
                y[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, y[t], 0, 0, 0);
            }
        }
        // lane holds, for point prow, neurons t*32 + (i&3) + 8*(i>>2) + 4*(lane>>5)
        const int64_t gp = p0 + prow;
        float best = INFINITY;
#pragma unroll
        for (int t = 0; t < NOUT_TILES; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int n = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (n < d_out) {
                    const float v = y[t][i] + b2[n];
                    if (out_raw && gp < B) out_raw[gp * d_out + n] = v;
                    if (select < 0) best = fminf(best, v);
                    else if (n == select) best = v;
                }
            }
        const float other = __shfl_xor(best, 32);
        best = fminf(best, other);  // for `select`, exactly one half holds the value, the other +inf
        if (lane < 32 && gp < B) out_min[gp] = best;
        __syncthreads();  // H and the chunk area are rewritten by the next tile
    }
}

int check_launch() { return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH; }

}  // namespace

extern "C" {

int hs_sdf_mlp_fwd(const float *x, const float *feat, const void *W0, const float *b0, const void *W1, const float *b1, const void *W2,
                   const float *b2, int32_t d_out, int32_t select, float *out_min, float *out_raw, int64_t B, void *stream) {
    if (d_out < 1 || d_out > 64 || select >= d_out) return HS_ERR_ARG;
    if (B == 0) return HS_OK;
    if (!x || !feat || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !out_min) return HS_ERR_NULL;
    const size_t lds = ((size_t)BM * HP + 2 * (size_t)HID * WP) * sizeof(uint16_t);
    const int64_t ntiles = (B + BM - 1) / BM;
    const int grid = (int)(ntiles < 256 ? ntiles : 256);  // one workgroup per CU (108 KB LDS), tiles strided across the grid
    hipStream_t st = (hipStream_t)stream;
    if (d_out <= 32) {
        static bool attr1 = false;
        if (!attr1) { (void)hipFuncSetAttribute((const void *)k_sdf_mlp<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr1 = true; }
        k_sdf_mlp<1><<<grid, kThreads, lds, st>>>(x, feat, (const uint16_t *)W0, b0, (const uint16_t *)W1, b1, (const uint16_t *)W2, b2, d_out, select,
                                                   out_min, out_raw, B);
    } else {
        static bool attr2 = false;
        if (!attr2) { (void)hipFuncSetAttribute((const void *)k_sdf_mlp<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr2 = true; }
        k_sdf_mlp<2><<<grid, kThreads, lds, st>>>(x, feat, (const uint16_t *)W0, b0, (const uint16_t *)W1, b1, (const uint16_t *)W2, b2, d_out, select,
                                                   out_min, out_raw, B);
    }
    return check_launch();
}

}  // extern "C"
