/*
 * oracle/hash_oracle_dt_impl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Body of hash_oracle_dt.c, included once per scalar type.
 *
 * The including file defines:  ST   storage type of the tensors (double, or uint16_t holding IEEE half bits)
 *                              VT   type a loaded element is held in (double; float holding a half-representable value)
 *                              PT   type of `float * element` under C++'s promotions (double; float)
 *                              LD(p, i) / ST_(p, i, v)   load / store element i
 *                              RT(x)   round a PT value to the element type (identity; float -> half -> float)
 *                              FN(name)   name ## suffix
 */

/* hashencoder.cu:104-254: forward and (dy_dx != NULL) dy_dx [B, L, D, C]; outputs [L, B, C] */
int FN(hs_oracle_dt_fwd)(const ST *x, const ST *emb, const int32_t *offsets, ST *out, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                         ST *dydx) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return -1;
    for (uint32_t level = 0; level < L; level++) {
        const level_t p = level_params(offsets, level, S, H);
        for (uint32_t b = 0; b < B; b++) {
            ST *o = out + ((size_t)level * B + b) * C;
            ST *j = dydx ? dydx + ((size_t)b * L + level) * D * C : NULL;
            uint32_t g[MAXD];
            float w[MAXD], dw[MAXD];
            int inside = p.table != 0;
            for (uint32_t d = 0; d < D; d++) {
                const VT v = LD(x, (size_t)b * D + d);
                if (v < 0 || v > 1) inside = 0;
            }
            if (!inside) {
                for (uint32_t c = 0; c < C; c++) ST_(o, c, (VT)0);
                if (j) for (uint32_t i = 0; i < D * C; i++) ST_(j, i, (VT)0);
                continue;
            }
            for (uint32_t d = 0; d < D; d++) {
                float pos = (float)LD(x, (size_t)b * D + d) * p.scale;           /* :158 */
                g[d] = (uint32_t)floorf(pos);
                pos -= (float)g[d];
                dw[d] = dsmooth(pos);
                w[d] = smooth(pos);
            }
            const ST *grid = emb + (size_t)offsets[level] * C;
            VT res[MAXC];
            for (uint32_t c = 0; c < C; c++) res[c] = 0;
            for (uint32_t idx = 0; idx < (1u << D); idx++) {                     /* :174-204 */
                float wt = 1;
                uint32_t gl[MAXD];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { wt *= 1 - w[d]; gl[d] = g[d]; }
                    else { wt *= w[d]; gl[d] = g[d] + 1; }
                }
                const uint32_t cell = cell_index(D, &p, gl);
                for (uint32_t c = 0; c < C; c++) {
                    const PT prod = (PT)wt * (PT)LD(grid, (size_t)cell * C + c);
                    res[c] = RT((PT)res[c] + prod);
                }
            }
            for (uint32_t c = 0; c < C; c++) ST_(o, c, res[c]);
            if (!j) continue;
            for (uint32_t gd = 0; gd < D; gd++) {                                 /* :214-252 */
                VT rg[MAXC];
                for (uint32_t c = 0; c < C; c++) rg[c] = 0;
                for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                    float wt = p.scale;
                    uint32_t gl[MAXD];
                    for (uint32_t nd = 0; nd < D - 1; nd++) {
                        const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                        if ((idx & (1u << nd)) == 0) { wt *= 1 - w[d]; gl[d] = g[d]; }
                        else { wt *= w[d]; gl[d] = g[d] + 1; }
                    }
                    gl[gd] = g[gd];
                    const uint32_t left = cell_index(D, &p, gl);
                    gl[gd] = g[gd] + 1;
                    const uint32_t right = cell_index(D, &p, gl);
                    for (uint32_t c = 0; c < C; c++) {
                        const VT diff = RT((PT)LD(grid, (size_t)right * C + c) - (PT)LD(grid, (size_t)left * C + c));      /* T - T -> T */
                        const PT term = (PT)wt * (PT)diff * (PT)dw[gd];
                        rg[c] = RT((PT)rg[c] + term);
                    }
                }
                for (uint32_t c = 0; c < C; c++) ST_(j, gd * C + c, rg[c]);
            }
        }
    }
    return 0;
}

/* hashencoder.cu:258-343 (scatter, sequential here: point order within a level) and :347-372 (input backward) */
int FN(hs_oracle_dt_bwd)(const ST *grad, const ST *x, const int32_t *offsets, ST *gemb, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                         const ST *dydx, ST *gx) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return -1;
    if (gemb) {
        for (uint32_t level = 0; level < L; level++) {
            const level_t p = level_params(offsets, level, S, H);
            if (p.table == 0) continue;
            ST *gg = gemb + (size_t)offsets[level] * C;
            for (uint32_t b = 0; b < B; b++) {
                uint32_t g[MAXD];
                float w[MAXD];
                int inside = 1;
                for (uint32_t d = 0; d < D; d++) {
                    const VT v = LD(x, (size_t)b * D + d);
                    if (v < 0 || v > 1) inside = 0;
                }
                if (!inside) continue;
                for (uint32_t d = 0; d < D; d++) {
                    float pos = (float)LD(x, (size_t)b * D + d) * p.scale;
                    g[d] = (uint32_t)floorf(pos);
                    pos -= (float)g[d];
                    w[d] = smooth(pos);
                }
                for (uint32_t idx = 0; idx < (1u << D); idx++) {
                    float wt = 1;
                    uint32_t gl[MAXD];
                    for (uint32_t d = 0; d < D; d++) {
                        if ((idx & (1u << d)) == 0) { wt *= 1 - w[d]; gl[d] = g[d]; }
                        else { wt *= w[d]; gl[d] = g[d] + 1; }
                    }
                    const uint32_t cell = cell_index(D, &p, gl);
                    for (uint32_t c = 0; c < C; c++) {
                        const VT add = RT((PT)wt * (PT)LD(grad, ((size_t)level * B + b) * C + c));       /* (T)(w * grad): the atomic's operand */
                        ST_(gg, (size_t)cell * C + c, RT((PT)LD(gg, (size_t)cell * C + c) + (PT)add));
                    }
                }
            }
        }
    }
    if (gx) {
        for (uint32_t b = 0; b < B; b++)
            for (uint32_t d = 0; d < D; d++) {
                VT r = 0;
                for (uint32_t l = 0; l < L; l++)
                    for (uint32_t c = 0; c < C; c++) {
                        const VT prod = RT((PT)LD(grad, ((size_t)l * B + b) * C + c) * (PT)LD(dydx, ((size_t)b * L + l) * D * C + d * C + c));   /* T * T -> T */
                        r = RT((PT)r + (PT)prod);
                    }
                ST_(gx, (size_t)b * D + d, r);
            }
    }
    return 0;
}

/* hashencoder.cu:376-428 (grad_grad [L, B, C]) and :432-595 (second-backward scatter) */
int FN(hs_oracle_dt_bwd2)(const ST *grad, const ST *x, const int32_t *offsets, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                          const ST *dydx, const ST *ggx, ST *gg_out, ST *g2emb) {
    if (D < 1 || D > MAXD || C < 1 || C > MAXC) return -1;
    if (gg_out) {
        for (uint32_t level = 0; level < L; level++)
            for (uint32_t b = 0; b < B; b++)
                for (uint32_t c = 0; c < C; c++) {
                    VT r = 0;
                    for (uint32_t d = 0; d < D; d++) {
                        const VT prod = RT((PT)LD(ggx, (size_t)b * D + d) * (PT)LD(dydx, ((size_t)b * L + level) * D * C + d * C + c));
                        r = RT((PT)r + (PT)prod);
                    }
                    ST_(gg_out, ((size_t)level * B + b) * C + c, r);
                }
    }
    if (g2emb) {
        for (uint32_t level = 0; level < L; level++) {
            const level_t p = level_params(offsets, level, S, H);
            if (p.table == 0) continue;
            ST *gt = g2emb + (size_t)offsets[level] * C;
            for (uint32_t b = 0; b < B; b++) {
                uint32_t g[MAXD];
                float w[MAXD], dw[MAXD];
                int inside = 1;
                for (uint32_t d = 0; d < D; d++) {
                    const VT v = LD(x, (size_t)b * D + d);
                    if (v < 0 || v > 1) inside = 0;
                }
                if (!inside) continue;
                for (uint32_t d = 0; d < D; d++) {
                    float pos = (float)LD(x, (size_t)b * D + d) * p.scale;
                    g[d] = (uint32_t)floorf(pos);
                    pos -= (float)g[d];
                    dw[d] = dsmooth(pos);
                    w[d] = smooth(pos);
                }
                VT cache[(1 << MAXD) * MAXC];
                for (uint32_t i = 0; i < (1u << D) * C; i++) cache[i] = 0;
                for (uint32_t gd = 0; gd < D; gd++) {
                    const VT g2 = LD(ggx, (size_t)b * D + gd);
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float wt = p.scale;
                        uint32_t bits = 0;
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? nd + 1 : nd;
                            if ((idx & (1u << nd)) == 0) wt *= 1 - w[d];
                            else { wt *= w[d]; bits |= 1u << d; }
                        }
                        const uint32_t left = bits, right = bits | (1u << gd);
                        for (uint32_t c = 0; c < C; c++) {
                            const PT v = (PT)wt * (PT)LD(grad, ((size_t)level * B + b) * C + c) * (PT)g2 * (PT)dw[gd];
                            cache[right * C + c] = RT((PT)cache[right * C + c] + v);
                            cache[left * C + c] = RT((PT)cache[left * C + c] - v);
                        }
                    }
                }
                for (uint32_t idx = 0; idx < (1u << D); idx++) {
                    uint32_t gl[MAXD];
                    for (uint32_t d = 0; d < D; d++) gl[d] = g[d] + ((idx >> d) & 1u);
                    const uint32_t cell = cell_index(D, &p, gl);
                    for (uint32_t c = 0; c < C; c++)
                        ST_(gt, (size_t)cell * C + c, RT((PT)LD(gt, (size_t)cell * C + c) + (PT)cache[idx * C + c]));
                }
            }
        }
    }
    return 0;
}
