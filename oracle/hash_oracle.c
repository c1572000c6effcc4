/*
 * oracle/hash_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential plain-C restatement of the multiresolution hash-grid encoder of
 * HoloScene's Stage-1 path (float32, any D in {2,3}, any C in {1,2,4,8}).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (holoscene_amd/) never does.
 *
 * Algorithm source (behaviour restated, no code copied):
 *   hashencoder/src/hashencoder.cu:36-72    index math (dense stride walk, xor hash, modulo)
 *   hashencoder/src/hashencoder.cu:87-93    smoothstep / its derivative
 *   hashencoder/src/hashencoder.cu:104-254  forward + dy_dx
 *   hashencoder/src/hashencoder.cu:258-343  first backward (scatter into grid grads)
 *   hashencoder/src/hashencoder.cu:347-372  input backward
 *   hashencoder/src/hashencoder.cu:376-428  second backward, grad_grad part
 *   hashencoder/src/hashencoder.cu:432-595  second backward, embedding part
 *
 * Pinning status: the reference kernels are CUDA-only and cannot be built in
 * this image without stand-in headers, so this file is pinned by (a) self
 * consistency tests (dy_dx vs finite differences, backward vs the transpose of
 * forward, second backward vs the derivative of the first) and (b) the golden
 * vectors produced by importing the reference's Python layer on top of it
 * (tests/golden/make_golden.py).  DESIGN.md says the same.
 *
 * One deliberate deviation, shared with the product: the per-level scale
 * exp2f(level*S)*H-1 is evaluated on the host (glibc exp2f) instead of with the
 * device's exp2f, so oracle and HIP kernels see bit-identical scales.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MAXD 3
#define MAXC 8

typedef struct {
    uint32_t table;      /* entries in this level            (hashencoder.cu:151) */
    float    scale;      /* exp2f(level*S)*H - 1             (hashencoder.cu:152) */
    uint32_t resolution; /* (uint32_t)ceil(scale) + 1        (hashencoder.cu:153) */
} level_t;

static level_t level_params(const int32_t *offsets, uint32_t level, float S, uint32_t H) {
    level_t p;
    p.table = (uint32_t)(offsets[level + 1] - offsets[level]);
    p.scale = exp2f((float)level * S) * (float)H - 1.0f;
    p.resolution = (uint32_t)ceilf(p.scale) + 1u;
    return p;
}

/* hashencoder.cu:36-72 */
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

static float smooth(float t) { return t * t * (3.0f - 2.0f * t); }      /* :87-89 */
static float dsmooth(float t) { return 6 * t * (1.0f - t); }             /* :91-93 */

/* returns 1 when the point is outside [0,1]^D (hashencoder.cu:124-131) */
static int locate(uint32_t D, const float *x, const level_t *p, uint32_t g[MAXD], float w[MAXD], float dw[MAXD]) {
    for (uint32_t d = 0; d < D; d++)
        if (x[d] < 0 || x[d] > 1) return 1;
    for (uint32_t d = 0; d < D; d++) {
        float pos = x[d] * p->scale;
        float fl = floorf(pos);
        g[d] = (uint32_t)fl;
        pos -= (float)g[d];
        dw[d] = dsmooth(pos);
        w[d] = smooth(pos);
    }
    return 0;
}

/*
 * Forward.  x [B,D]; emb [sum table, C]; offsets [L+1]; out [L,B,C];
 * dydx [B,L,D,C] (written when calc_dydx).        hashencoder.cu:104-254
 */
int hs_oracle_hash_fwd(const float *x, const float *emb, const int32_t *offsets, float *out,
                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                       int calc_dydx, float *dydx) {
    if (D < 2 || D > MAXD || C > MAXC || C == 0) return -1;
    for (uint32_t level = 0; level < L; level++) {
        const level_t p = level_params(offsets, level, S, H);
        const float *grid = emb + (size_t)(uint32_t)offsets[level] * C;
        for (uint32_t b = 0; b < B; b++) {
            float *o = out + ((size_t)level * B + b) * C;
            float *j = calc_dydx ? dydx + ((size_t)b * L + level) * D * C : 0;
            uint32_t g[MAXD];
            float w[MAXD], dw[MAXD];
            if (locate(D, x + (size_t)b * D, &p, g, w, dw)) {
                for (uint32_t c = 0; c < C; c++) o[c] = 0;
                if (j) for (uint32_t i = 0; i < D * C; i++) j[i] = 0;
                continue;
            }
            float acc[MAXC] = {0};
            for (uint32_t corner = 0; corner < (1u << D); corner++) {
                float wt = 1;
                uint32_t gl[MAXD];
                for (uint32_t d = 0; d < D; d++) {
                    if ((corner & (1u << d)) == 0) { wt *= 1 - w[d]; gl[d] = g[d]; }
                    else                           { wt *= w[d];     gl[d] = g[d] + 1; }
                }
                const float *e = grid + (size_t)cell_index(D, &p, gl) * C;
                for (uint32_t c = 0; c < C; c++) acc[c] += wt * e[c];
            }
            for (uint32_t c = 0; c < C; c++) o[c] = acc[c];
            if (!j) continue;
            for (uint32_t gd = 0; gd < D; gd++) {                 /* :214-252 */
                float ga[MAXC] = {0};
                for (uint32_t corner = 0; corner < (1u << (D - 1)); corner++) {
                    float wt = p.scale;
                    uint32_t gl[MAXD];
                    for (uint32_t nd = 0; nd < D - 1; nd++) {
                        const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                        if ((corner & (1u << nd)) == 0) { wt *= 1 - w[d]; gl[d] = g[d]; }
                        else                            { wt *= w[d];     gl[d] = g[d] + 1; }
                    }
                    gl[gd] = g[gd];
                    const float *lo = grid + (size_t)cell_index(D, &p, gl) * C;
                    gl[gd] = g[gd] + 1;
                    const float *hi = grid + (size_t)cell_index(D, &p, gl) * C;
                    for (uint32_t c = 0; c < C; c++) ga[c] += wt * (hi[c] - lo[c]) * dw[gd];
                }
                for (uint32_t c = 0; c < C; c++) j[gd * C + c] = ga[c];
            }
        }
    }
    return 0;
}

/*
 * First backward.  grad [L,B,C]; grad_emb accumulates (caller zeroes);
 * grad_x [B,D] written when calc_dydx.        hashencoder.cu:258-372
 */
int hs_oracle_hash_bwd(const float *grad, const float *x, const float *emb, const int32_t *offsets,
                       float *grad_emb, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                       int calc_dydx, const float *dydx, float *grad_x) {
    (void)emb;
    if (D < 2 || D > MAXD || C > MAXC || C == 0) return -1;
    for (uint32_t level = 0; level < L; level++) {
        const level_t p = level_params(offsets, level, S, H);
        float *gg = grad_emb + (size_t)(uint32_t)offsets[level] * C;
        for (uint32_t b = 0; b < B; b++) {
            uint32_t g[MAXD];
            float w[MAXD], dw[MAXD];
            if (locate(D, x + (size_t)b * D, &p, g, w, dw)) continue;
            const float *go = grad + ((size_t)level * B + b) * C;
            for (uint32_t corner = 0; corner < (1u << D); corner++) {
                float wt = 1;
                uint32_t gl[MAXD];
                for (uint32_t d = 0; d < D; d++) {
                    if ((corner & (1u << d)) == 0) { wt *= 1 - w[d]; gl[d] = g[d]; }
                    else                           { wt *= w[d];     gl[d] = g[d] + 1; }
                }
                float *e = gg + (size_t)cell_index(D, &p, gl) * C;
                for (uint32_t c = 0; c < C; c++) e[c] += wt * go[c];
            }
        }
    }
    if (calc_dydx) {                                              /* :347-372 */
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t d = 0; d < D; d++) {
                float r = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t c = 0; c < C; c++)
                        r += grad[((size_t)l * B + b) * C + c] * dydx[(((size_t)b * L + l) * D + d) * C + c];
                grad_x[(size_t)b * D + d] = r;
            }
    }
    return 0;
}

/*
 * Second backward.  ggx [B,D] is the cotangent of grad_x.  Writes
 * grad_grad [L,B,C] and accumulates grad2_emb (caller zeroes).  The reference
 * returns no gradient w.r.t. x here and omits smoothstep''.   hashencoder.cu:376-595
 */
int hs_oracle_hash_bwd2(const float *grad, const float *x, const float *emb, const int32_t *offsets,
                        uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                        const float *dydx, const float *ggx, float *grad_grad, float *grad2_emb) {
    (void)emb;
    if (D < 2 || D > MAXD || C > MAXC || C < 2) return -1;         /* C==1 unsupported (:678-684) */
    for (uint32_t level = 0; level < L; level++) {
        const level_t p = level_params(offsets, level, S, H);
        float *g2 = grad2_emb + (size_t)(uint32_t)offsets[level] * C;
        for (uint32_t b = 0; b < B; b++) {
            const float *gx = ggx + (size_t)b * D;
            const float *j = dydx + ((size_t)b * L + level) * D * C;
            float *o = grad_grad + ((size_t)level * B + b) * C;
            for (uint32_t c = 0; c < C; c++) {                    /* :400-413 */
                float r = 0;
                for (uint32_t d = 0; d < D; d++) r += gx[d] * j[d * C + c];
                o[c] = r;
            }
            uint32_t g[MAXD];
            float w[MAXD], dw[MAXD];
            if (locate(D, x + (size_t)b * D, &p, g, w, dw)) continue;
            const float *go = grad + ((size_t)level * B + b) * C;
            float cache[(1u << MAXD) * MAXC];
            memset(cache, 0, sizeof cache);
            for (uint32_t gd = 0; gd < D; gd++) {                 /* :511-549 */
                for (uint32_t corner = 0; corner < (1u << (D - 1)); corner++) {
                    float wt = p.scale;
                    uint32_t bits = 0;
                    for (uint32_t nd = 0; nd < D - 1; nd++) {
                        const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                        if ((corner & (1u << nd)) == 0) wt *= 1 - w[d];
                        else { wt *= w[d]; bits |= 1u << d; }
                    }
                    const uint32_t lo = bits, hi = bits | (1u << gd);
                    for (uint32_t c = 0; c < C; c++) {
                        const float v = wt * go[c] * gx[gd] * dw[gd];
                        cache[hi * C + c] += v;
                        cache[lo * C + c] -= v;
                    }
                }
            }
            for (uint32_t corner = 0; corner < (1u << D); corner++) { /* :555-594 */
                uint32_t gl[MAXD];
                for (uint32_t d = 0; d < D; d++) gl[d] = g[d] + ((corner >> d) & 1u);
                float *e = g2 + (size_t)cell_index(D, &p, gl) * C;
                for (uint32_t c = 0; c < C; c++) e[c] += cache[corner * C + c];
            }
        }
    }
    return 0;
}

/* Per-level table exposed so tests can pin scale/resolution/table sizes. */
int hs_oracle_level_table(const int32_t *offsets, uint32_t L, float S, uint32_t H,
                          float *scale, uint32_t *resolution, uint32_t *table) {
    for (uint32_t l = 0; l < L; l++) {
        level_t p = level_params(offsets, l, S, H);
        scale[l] = p.scale; resolution[l] = p.resolution; table[l] = p.table;
    }
    return 0;
}
