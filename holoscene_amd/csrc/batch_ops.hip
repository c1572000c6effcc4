// holoscene_amd/csrc/batch_ops.hip -- the per-iteration pixel draw of a training batch, on the device (gfx950).
//
// Reference: NSDataset.__getitem__ (datasets/ns_dataset.py:409-430), run in 8 DataLoader worker processes: half of the R rays split
// evenly over the instance classes present in the frame (class 0 takes the remainder; a class with fewer pixels than its quota gives all
// of them), each share = the first `want` entries of torch.randperm over the class's pixel list, i.e. a uniformly random subset; the
// other half = the first entries of a permutation of ALL pixels.  On the host that is ~4.5 ms per batch (one 262 144-element
// permutation plus one per class) -- twice a whole training iteration here.  On the device it is one launch: workgroup c draws the
// subset of class c (the last workgroup the uniform half) by rejection into an LDS hash set -- `want` is at most a few hundred out of
// thousands, so a round accepts nearly every candidate and two or three rounds finish --, then orders the accepted set by a hash of
// (pixel, stream position): which thread won a duplicate's insertion must not show in the output, so that a (seed, counter) pair names
// ONE batch whatever the scheduling.  Same distribution as the reference's rule (a uniformly random subset per class; the order inside
// a batch is immaterial to every consumer); the reference's own permutations remain injectable on the host path for the parity fixtures.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "holoscene_hip.h"

namespace {

constexpr int kDrawThreads = 256;
constexpr uint32_t kEmpty = 0xffffffffu;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {      // splitmix64 finaliser
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// LDS: table[cap] (open addressing, linear probing) | keys[2 want_max] | vals[2 want_max]
__global__ __launch_bounds__(kDrawThreads) void k_draw_pixels(const int32_t *__restrict__ class_ptr, const int32_t *__restrict__ class_pix,
                                                               const int32_t *__restrict__ out_off, int32_t n_cls, int32_t per_class, int32_t n_bg,
                                                               int32_t n_uniform, int32_t total_pixels, uint64_t seed, uint64_t counter,
                                                               int64_t *__restrict__ out, uint32_t cap, uint32_t want_max) {
    extern __shared__ uint32_t sm[];
    uint32_t *table = sm, *keys = sm + cap, *vals = keys + 2 * want_max;
    __shared__ uint32_t count;
    const int c = blockIdx.x;
    const bool uniform = c == n_cls;
    const int32_t first = uniform ? 0 : class_ptr[c];
    const int32_t n = uniform ? total_pixels : class_ptr[c + 1] - first;
    const int32_t quota = uniform ? n_uniform : (c == 0 ? n_bg : per_class);
    int64_t *dst = out + out_off[c];
    if (n <= quota) {           // the whole class (ns_dataset.py:422-427); never taken by the uniform half of a real image
        for (int i = threadIdx.x; i < n; i += kDrawThreads) dst[i] = uniform ? i : class_pix[first + i];
        return;
    }
    const uint64_t stream = mix64(seed ^ mix64(counter * 0x100000001b3ull + (uint64_t)c));
    // a class not much larger than its quota: rejection would spend its time re-drawing members (coupon collector); instead every member
    // gets a key and the `quota` smallest win -- the list fits the sort arrays (n <= 2 quota <= 2 want_max)
    const bool all_members = n <= 2 * quota;
    uint32_t have = all_members ? (uint32_t)n : 0u;
    if (all_members)
        for (uint32_t i = threadIdx.x; i < (uint32_t)n; i += kDrawThreads) vals[i] = i;
    else
        for (uint32_t i = threadIdx.x; i < cap; i += kDrawThreads) table[i] = kEmpty;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    for (uint32_t round = 0; !all_members && have < (uint32_t)quota; round++) {
        const uint32_t need = (uint32_t)quota - have;
        for (uint32_t t = threadIdx.x; t < need; t += kDrawThreads) {
            const uint64_t r = mix64(stream + ((uint64_t)round << 32) + t);
            const uint32_t cand = (uint32_t)(((r >> 32) * (uint64_t)n) >> 32);       // position in the class's list, [0, n)
            uint32_t slot = (uint32_t)mix64(cand) & (cap - 1);
            for (;;) {
                const uint32_t prev = atomicCAS(&table[slot], kEmpty, cand);
                if (prev == kEmpty) {           // new member: any free list slot, the final order comes from the sort below
                    const uint32_t at = atomicAdd(&count, 1u);
                    vals[at] = cand;
                    break;
                }
                if (prev == cand) break;        // already a member (an earlier round, or a twin in this one)
                slot = (slot + 1) & (cap - 1);
            }
        }
        __syncthreads();
        have = count;
        __syncthreads();
    }
    // order: by hash of (member, stream) -- a pseudo-random order that depends on the SET only (bitonic sort of the 64-bit (hash, member)
    // pairs split into two 32-bit arrays; quota <= want_max, padded with maximal keys)
    uint32_t m = 1;
    while (m < have) m <<= 1;
    for (uint32_t i = threadIdx.x; i < m; i += kDrawThreads) {
        if (i < have) keys[i] = (uint32_t)(mix64(stream ^ ((uint64_t)vals[i] << 20)) >> 32);
        else { keys[i] = 0xffffffffu; vals[i] = 0xffffffffu; }
    }
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < m; i += kDrawThreads) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const uint32_t ki = keys[i], kl = keys[l], vi = vals[i], vl = vals[l];
                    const bool gt = ki > kl || (ki == kl && vi > vl);
                    if (gt == up) { keys[i] = kl; keys[l] = ki; vals[i] = vl; vals[l] = vi; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < quota; i += kDrawThreads) dst[i] = uniform ? (int64_t)vals[i] : (int64_t)class_pix[first + vals[i]];
}

}  // namespace

extern "C" {

int hs_draw_pixels(const int32_t *class_ptr, const int32_t *class_pix, const int32_t *out_off, int32_t n_cls, int32_t per_class, int32_t n_bg,
                   int32_t n_uniform, int32_t total_pixels, uint64_t seed, uint64_t counter, int64_t *out, void *stream) {
    if (n_cls < 1 || per_class < 0 || n_bg < 0 || n_uniform < 0 || total_pixels < 1) return HS_ERR_ARG;
    if (!class_ptr || !class_pix || !out_off || !out) return HS_ERR_NULL;
    int32_t want = n_bg > per_class ? n_bg : per_class;
    if (n_uniform > want) want = n_uniform;
    if (want > HS_DRAW_MAX_WANT) return HS_ERR_ARG;
    uint32_t want_max = 1;
    while (want_max < (uint32_t)(want > 1 ? want : 1)) want_max <<= 1;      // the sort pads to a power of two
    const uint32_t cap = 4 * want_max;                                       // load factor <= 1/4
    const size_t lds = ((size_t)cap + 4 * (size_t)want_max) * sizeof(uint32_t);
    // (per call, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE, and the call is cheap)
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_draw_pixels, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(8 * (size_t)HS_DRAW_MAX_WANT * sizeof(uint32_t)));
    k_draw_pixels<<<n_cls + 1, kDrawThreads, lds, (hipStream_t)stream>>>(class_ptr, class_pix, out_off, n_cls, per_class, n_bg, n_uniform, total_pixels,
                                                                      seed, counter, out, cap, want_max);
    return hipGetLastError() == hipSuccess ? HS_OK : HS_ERR_LAUNCH;
}

}  // extern "C"
