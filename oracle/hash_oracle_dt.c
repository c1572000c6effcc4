/*
 * oracle/hash_oracle_dt.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The hash-grid encoder of oracle/hash_oracle.c in the reference's OTHER scalar types, double and half
 * (AT_DISPATCH_FLOATING_TYPES_AND_HALF, hashencoder/src/hashencoder.cu:747, 778, 817): sequential plain C, the reference's
 * order of operations with C++'s promotions written out (float * double -> double; at::Half: every operation in float,
 * rounded to half).  Half is emulated on IEEE bits (this image's gcc 11 has no _Float16 on x86): h2f / f2h below,
 * round-to-nearest-even.  Checks csrc/hash_encode_dt.hip (tests/test_hash_dtypes_gpu.py); pinned only by its agreement with
 * hash_oracle.c's float results on float-representable inputs (tests/test_oracle_golden.py) -- "parity unpinned" like that file.
 *
 * Algorithm source (behaviour restated, no code copied): hashencoder.cu:36-72, 87-93, 104-254, 258-343, 347-372, 376-428, 432-595.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MAXD 3
#define MAXC 8

typedef struct { uint32_t table; float scale; uint32_t resolution; } level_t;

static level_t level_params(const int32_t *offsets, uint32_t level, float S, uint32_t H) {
    level_t p;
    p.table = (uint32_t)(offsets[level + 1] - offsets[level]);
    p.scale = exp2f((float)level * S) * (float)H - 1.0f;
    p.resolution = (uint32_t)ceilf(p.scale) + 1u;
    return p;
}

static uint32_t cell_index(uint32_t D, const level_t *p, const uint32_t g[MAXD]) {
    static const uint32_t primes[3] = {1u, 2654435761u, 805459861u};
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= p->table; d++) {
        index += g[d] * stride;
        stride *= p->resolution;
    }
    if (stride > p->table) {
        index = 0;
        for (uint32_t d = 0; d < D; d++) index ^= g[d] * primes[d];
    }
    return index % p->table;
}

static float smooth(float t) { return t * t * (3.0f - 2.0f * t); }
static float dsmooth(float t) { return 6 * t * (1.0f - t); }

/* IEEE binary16 <-> binary32 */
static float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {                                  /* subnormal: normalise */
            int e = -1;
            uint32_t m = man;
            do { e++; m <<= 1; } while ((m & 0x400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
static uint16_t f2h(float f) {                  /* round to nearest even */
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0u));      /* inf / nan */
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                          /* >= 65520: overflows to inf */
    if (ax < 0x33000001u) return sign;                                                                 /* <= 2^-25: zero */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;     /* 24-bit significand */
    int shift;                                     /* bits to drop */
    uint32_t hexp;
    if (e < -14) { shift = 13 + (-14 - e); hexp = 0; }      /* subnormal half */
    else { shift = 13; hexp = (uint32_t)(e + 15); }
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t h;
    if (hexp == 0) h = q;                          /* a carry into bit 10 makes the smallest normal: right as it is */
    else h = ((hexp - 1) << 10) + q;               /* q has its leading 1 at bit 10: adds one to the exponent field; a carry out of the mantissa adds another */
    return (uint16_t)(sign | h);
}
static float hround(float v) { return h2f(f2h(v)); }
uint16_t hs_oracle_f2h(float f) { return f2h(f); }       /* (exported for the conversion's own test) */
float hs_oracle_h2f(uint16_t h) { return h2f(h); }

/* ---- double */
#define ST double
#define VT double
#define PT double
#define LD(p, i) ((p)[i])
#define ST_(p, i, v) ((p)[i] = (v))
#define RT(x) (x)
#define FN(n) n##_f64
#include "hash_oracle_dt_impl.h"
#undef ST
#undef VT
#undef PT
#undef LD
#undef ST_
#undef RT
#undef FN

/* ---- half (bits in uint16_t; values held in float, every result rounded) */
#define ST uint16_t
#define VT float
#define PT float
#define LD(p, i) h2f((p)[i])
#define ST_(p, i, v) ((p)[i] = f2h(v))
#define RT(x) hround(x)
#define FN(n) n##_f16
#include "hash_oracle_dt_impl.h"
