// experiment: HBM write rate of a [M, 256] bf16 activation image as a function of how much of a row one store instruction covers.
// Each wave owns 32-row blocks (16 KB) and writes them with 16 global_store_dwordx4; SEG = contiguous bytes per row per instruction:
//   32  -> the wave-tile kernels' pattern (two lanes per row, 32 rows per instruction)
//   64 / 128 / 512 -> 4 / 8 / 32 lanes per row (16 / 8 / 2 rows per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int SEG>
__global__ __launch_bounds__(512) void k_store(uint4 *__restrict__ H0, uint4 *__restrict__ H1, int64_t nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = SEG / 16;            // lanes per row
    constexpr int RPI = 64 / LPR;            // rows per instruction
    for (int64_t blk = (int64_t)blockIdx.x * 8 + wave; blk < nblocks; blk += (int64_t)gridDim.x * 8) {
        char *b0 = reinterpret_cast<char *>(H0) + blk * 16384, *b1 = reinterpret_cast<char *>(H1) + blk * 16384;
        const uint4 v = make_uint4((uint32_t)blk, lane, 1, 2);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            // instruction i covers rows (i % (32 / RPI)) * RPI .. and segment i / (32 / RPI) of those rows
            constexpr int groups = 32 / RPI;
            const int row = (i % groups) * RPI + lane / LPR, seg = i / groups;
            const size_t off = (size_t)row * 512 + (size_t)seg * SEG + (lane % LPR) * 16;
            *reinterpret_cast<uint4 *>(b0 + off) = v;
            *reinterpret_cast<uint4 *>(b1 + off) = v;
        }
    }
}
// the same 32-byte pieces, but the pieces of one 128-byte line are written CHUNK bytes at a time with the rest of the wave's 8 row blocks
// (128 KB) in between -- the wave-tile kernels finish a line in two 64-byte halves a phase apart
template <int CHUNK>
__global__ __launch_bounds__(512) void k_store_apart(uint4 *__restrict__ H0, uint4 *__restrict__ H1, int64_t nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 31, h = lane >> 5;
    constexpr int PIECES = CHUNK / 32;        // 32-byte pieces written together
    for (int64_t blk0 = ((int64_t)blockIdx.x * 8 + wave) * 8; blk0 < nblocks; blk0 += (int64_t)gridDim.x * 64) {
        const uint4 v = make_uint4((uint32_t)blk0, lane, 1, 2);
        for (int c = 0; c < 512 / CHUNK; c++)
            for (int b = 0; b < 8; b++) {
                if (blk0 + b >= nblocks) continue;
                char *b0 = reinterpret_cast<char *>(H0) + (blk0 + b) * 16384, *b1 = reinterpret_cast<char *>(H1) + (blk0 + b) * 16384;
#pragma unroll
                for (int p = 0; p < PIECES; p++) {
                    const size_t off = (size_t)row * 512 + (size_t)c * CHUNK + p * 32 + h * 16;
                    *reinterpret_cast<uint4 *>(b0 + off) = v;
                    *reinterpret_cast<uint4 *>(b1 + off) = v;
                }
            }
    }
}
template <int CHUNK> static void run_apart(uint4 *H0, uint4 *H1, int64_t nblocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) k_store_apart<CHUNK><<<256, 512>>>(H0, H1, nblocks);
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) k_store_apart<CHUNK><<<256, 512>>>(H0, H1, nblocks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * nblocks * 16384;
    printf("pieces of a line %3d B at a time, 128 KB apart: %.1f us per launch, %.2f TB/s\n", CHUNK, ms * 100, bytes / (ms * 1e-4) / 1e12);
}
template <int SEG> static void run(uint4 *H0, uint4 *H1, int64_t nblocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; r++) k_store<SEG><<<256, 512>>>(H0, H1, nblocks);
    hipEventRecord(e0);
    for (int r = 0; r < 10; r++) k_store<SEG><<<256, 512>>>(H0, H1, nblocks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * nblocks * 16384;
    printf("SEG %3d B/row/instr: %.1f us per launch, %.2f TB/s\n", SEG, ms * 100, bytes / (ms * 1e-4) / 1e12);
}
int main() {
    const int64_t M = 417792, nblocks = M / 32;
    uint4 *H0, *H1; hipMalloc(&H0, M * 512); hipMalloc(&H1, M * 512);
    run<32>(H0, H1, nblocks); run<64>(H0, H1, nblocks); run<128>(H0, H1, nblocks); run<512>(H0, H1, nblocks);
    run<32>(H0, H1, nblocks);
    run_apart<32>(H0, H1, nblocks); run_apart<64>(H0, H1, nblocks); run_apart<128>(H0, H1, nblocks); run_apart<256>(H0, H1, nblocks);
    return 0;
}
