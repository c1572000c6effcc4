// wave_ops.h -- wave64 prefix sums and reductions on DPP operands (gfx9 row_shr / row_bcast), shared by the sampler and compositing
// kernels.  HIP's __shfl_up / __shfl_xor compile to ds_bpermute_b32: an address computation, a trip through the LDS crossbar (~64+
// cycles of latency each, six in a row per scan) and an lgkmcnt wait per step.  The sampler's line search runs two scans and a maximum
// per evaluation, eleven evaluations per ray and round; as DPP the same scan is six dependent VALU instructions.
// The association of the floating-point sums differs from the shuffle form (rows of 16 first, then the row totals); every caller's
// result is a cumulative sum the reference builds with torch.cumsum, whose association is not specified either.
#pragma once
#include <hip/hip_runtime.h>

namespace hs_wave {

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_or(float fallback, float v) {      // lanes without a source (or masked out) get `fallback`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fallback), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}

constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118, kRowBcast15 = 0x142, kRowBcast31 = 0x143;

__device__ __forceinline__ float incl_scan(float v) {
    v += dpp_or<kRowShr1>(0.f, v);
    v += dpp_or<kRowShr2>(0.f, v);
    v += dpp_or<kRowShr4>(0.f, v);
    v += dpp_or<kRowShr8>(0.f, v);                 // inclusive scan inside each row of 16 lanes
    v += dpp_or<kRowBcast15, 0xa>(0.f, v);         // rows 1, 3 += total of rows 0, 2
    v += dpp_or<kRowBcast31, 0xc>(0.f, v);         // rows 2, 3 += total of rows 0 + 1
    return v;
}

__device__ __forceinline__ float sum(float v) {     // the same value on every lane
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl_scan(v)), 63));
}

__device__ __forceinline__ float max(float v) {     // the same value on every lane
    v = fmaxf(v, dpp_or<kRowShr1>(v, v));
    v = fmaxf(v, dpp_or<kRowShr2>(v, v));
    v = fmaxf(v, dpp_or<kRowShr4>(v, v));
    v = fmaxf(v, dpp_or<kRowShr8>(v, v));
    v = fmaxf(v, dpp_or<kRowBcast15, 0xa>(v, v));
    v = fmaxf(v, dpp_or<kRowBcast31, 0xc>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The same scan / maximum with the DPP operand folded INTO the arithmetic instruction (v_add_f32_dpp / v_max_f32_dpp, hand-written: the compiler
// emits v_mov_b32 (fallback) + v_mov_b32_dpp + the add, and for fmaxf two canonicalising v_max more -- 5 instructions and a hazard nop per step).
// A lane without a source keeps its value (a disabled DPP lane is not written), which is "+ 0" / "max with itself": the sums associate exactly as
// in incl_scan above, so the results are bit-identical.  A DPP source written by the previous VALU instruction needs two wait states that nothing
// inserts inside an asm block: the two interleaved chains of incl_scan2 supply one, an s_nop the other.  Used by the sampler's line search, which
// is bound by its instruction count (11 evaluations per ray and round, two scans and a maximum each).
__device__ __forceinline__ void incl_scan2(float &a, float &b) {
    asm volatile(
        "s_nop 4\n\t"      /* 5 wait states: covers VALU-writes-EXEC -> DPP as well as VALU-write -> DPP read (the hazard recogniser does not see inside the block) */
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}

__device__ __forceinline__ float max_dpp(float v) {     // hs_wave::max for values that are not NaN on any lane (v_max_f32 then IS fmaxf)
    asm volatile(
        "s_nop 4\n\t"      /* 5 wait states: covers VALU-writes-EXEC -> DPP as well as VALU-write -> DPP read (the hazard recogniser does not see inside the block) */
        "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

}  // namespace hs_wave
