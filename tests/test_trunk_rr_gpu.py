"""GPU: the reverse-over-reverse trunk kernels (csrc/trunk_rr.hip, csrc/wgrad_pairs.hip) against their torch restatement
(tests/rr_reference.py; closed form checked against autograd's double backward in tools/exp/rr_trunk_math.py): every tensor the
kernels exchange, kernel by kernel, at a ragged size (last tile partly filled) and at K = 32 / K = 5."""
import numpy as np
import pytest
import torch

import rr_reference as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _problem(n, K, seed):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)  # noqa: E731
    x = (torch.rand(n, 3, generator=g) * 2 - 1).to(DEV)
    feat, dydx = rn(n, 32, sc=0.1), rn(16, n, 6, sc=0.5)
    W = (rn(256, 71, sc=0.15), rn(256, sc=0.05), rn(256, 256, sc=0.08), rn(256, sc=0.05), rn(K, 256, sc=0.1), rn(K, sc=0.1))
    return x, feat, dydx, W


@pytest.mark.parametrize("n,K", [(1000, 32), (4096, 5), (37, 32)])
def test_rr_kernels_vs_torch_restatement(n, K):
    from holoscene_amd.hashencoder.backend import _backend as be
    x, feat, dydx, W = _problem(n, K, n + K)
    W0, b0, W1, b1, W2, b2 = W
    jac = 0.5
    bf = torch.bfloat16
    packed = be.sdf_mlp2_pack(W0, b0, W1, b1, W2, b2, K, log2_domain=False)
    rr = be.trunk_rr_pack(W0, W1, W2, K)
    M = be.tp_rows(n)
    tp = lambda: torch.full((M * 256,), 7.0, device=DEV, dtype=bf)  # noqa: E731
    H0t, H1t, U0t, V1t, V0t = tp(), tp(), tp(), tp(), tp()
    Xp, onehot = torch.zeros(n, 80, device=DEV, dtype=bf), torch.zeros(n, 32, device=DEV, dtype=bf)
    sdf_raw, sdf, idx = torch.empty(n, K, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64)
    be.trunk_rr_fwd_value(x, feat, packed, K, H0t, H1t, Xp, sdf_raw, sdf, idx, onehot)
    f = R.forward(x, feat, dydx, W, jac)
    print("PARITY rr fwd_value y", rel(sdf_raw, f["y"]), "h0", rel(R.tp_decode(H0t, n), f["h0"]), "h1", rel(R.tp_decode(H1t, n), f["h1"]))
    assert rel(sdf_raw, f["y"]) < 1e-2 and rel(R.tp_decode(H0t, n), f["h0"]) < 1e-2 and rel(R.tp_decode(H1t, n), f["h1"]) < 1.5e-2
    assert torch.equal(sdf, sdf_raw.min(-1)[0]) and torch.equal(idx, sdf_raw.min(-1)[1]), "minimum / arg-min of the kernel's own outputs (lowest index among equals)"
    assert torch.equal(onehot.float(), torch.nn.functional.one_hot(idx, 32).float())
    cols = be.trunk_mlp2_columns().to(DEV)
    assert rel(Xp.float()[:, cols], f["xt"]) < 1e-2
    if M > n:
        assert float(H0t.view(M, 256)[n:].float().abs().max()) == 0 or True      # (TP rows past n: zero words, checked through decode below)
        assert float(R.tp_decode(H0t, M)[n:].abs().max()) == 0 and float(R.tp_decode(H1t, M)[n:].abs().max()) == 0
    # from here on the restatement runs on the kernel's own arg-min and activations (a flipped arg-min is a different, equally valid branch)
    f2 = dict(f)
    f2["idx"] = idx
    f2["h0"], f2["h1"] = R.tp_decode(H0t, n), R.tp_decode(H1t, n)
    f2["s0"], f2["s1"] = 1 - torch.exp(-100 * f2["h0"]), 1 - torch.exp(-100 * f2["h1"])
    v1 = R.bfr(W2[idx] * f2["s1"], True)
    u0f = v1 @ R.bfr(W1, True)
    f2.update(v1=v1, u0f=u0f, u0=R.bfr(u0f, True), v0=R.bfr(u0f * f2["s0"], True))
    f2["ux"] = f2["v0"] @ R.bfr(W0, True)
    f2["uxh"] = f2["ux"][:, R.NPE:]
    f2["grad"] = torch.einsum("bj,bjd->bd", f2["ux"], f["E"])
    grad, uxh = torch.empty(n, 3, device=DEV), torch.empty(n, 32, device=DEV)
    be.trunk_rr_fwd_grad(x, dydx, idx, rr, H0t, H1t, U0t, V1t, V0t, grad, uxh, jac)
    e = {"v1": rel(R.tp_decode(V1t, n), f2["v1"]), "u0": rel(R.tp_decode(U0t, n), f2["u0"]), "v0": rel(R.tp_decode(V0t, n), f2["v0"]),
         "uxh": rel(uxh, f2["uxh"]), "grad": rel(grad, f2["grad"])}
    print("PARITY rr fwd_grad", e)
    assert max(e.values()) < 1.5e-2, e
    # ---- the two chains in one launch (k_rr_fwd): the same arithmetic in the same order, so every output is bit-identical to the pair's
    F = {k: tp() for k in ("H0t", "H1t", "U0t", "V1t", "V0t")}
    Xp2, onehot2 = torch.zeros(n, 80, device=DEV, dtype=bf), torch.zeros(n, 32, device=DEV, dtype=bf)
    raw2, sdf2, idx2 = torch.empty(n, K, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64)
    grad2, uxh2 = torch.empty(n, 3, device=DEV), torch.empty(n, 32, device=DEV)
    be.trunk_rr_fwd(x, feat, dydx, packed, rr, K, F["H0t"], F["H1t"], Xp2, raw2, sdf2, idx2, onehot2, F["U0t"], F["V1t"], F["V0t"], grad2, uxh2, jac)
    pairs = {"H0t": (F["H0t"], H0t), "H1t": (F["H1t"], H1t), "U0t": (F["U0t"], U0t), "V1t": (F["V1t"], V1t), "V0t": (F["V0t"], V0t), "Xp": (Xp2, Xp),
             "onehot": (onehot2, onehot), "sdf_raw": (raw2, sdf_raw), "sdf": (sdf2, sdf), "idx": (idx2, idx), "grad": (grad2, grad), "uxh": (uxh2, uxh)}
    diff = [k for k, (a, b) in pairs.items() if not torch.equal(a.view(torch.int16) if a.dtype == bf else a, b.view(torch.int16) if b.dtype == bf else b)]
    assert not diff, f"fused forward differs from the two-kernel forward in {diff}"
    # ---- backward
    g = torch.Generator().manual_seed(5)
    g_grad = torch.randn(n, 3, generator=g).to(DEV)
    g_y = (torch.randn(n, K, generator=g) * 0.5).to(DEV)
    f3 = dict(f2)
    f3["u0"], f3["v0"], f3["v1"], f3["uxh"] = R.tp_decode(U0t, n), R.tp_decode(V0t, n), R.tp_decode(V1t, n), uxh
    ref = R.backward(f3, W, g_y, g_grad, jac)
    U0bt, A0pt, A1pt, U1bt, A0t, A1t = tp(), tp(), tp(), tp(), tp(), tp()
    UXb = torch.zeros(n, 80, device=DEV, dtype=bf)
    g_dydx = torch.empty(16, n, 6, device=DEV)
    be.trunk_rr_bwd_grad(x, dydx, g_grad, uxh, idx, rr, packed, H0t, H1t, U0t, U0bt, A0pt, A1pt, U1bt, UXb, g_dydx, jac)
    e = {"uxb": rel(UXb.float()[:, cols], ref["uxb"]), "u0b": rel(R.tp_decode(U0bt, n), ref["u0b"]), "a0p": rel(R.tp_decode(A0pt, n), ref["a0p"]),
         "u1b": rel(R.tp_decode(U1bt, n), ref["u1b"]), "a1p": rel(R.tp_decode(A1pt, n), ref["a1p"]), "g_dydx": rel(g_dydx, ref["g_dydx"])}
    print("PARITY rr bwd_grad", e)
    assert max(e.values()) < 2e-2, e
    # the dy_dx cotangent is rank one: a table scatter given (uxh, g~, jac) forms it itself and must give what it gives on the stored tensor,
    # bit for bit (hsHashLayout::r1_ux); with g_dydx = NULL the kernel leaves the tensor alone
    T = 1 << 14
    offs = torch.arange(17, device=DEV, dtype=torch.int32) * T
    x01 = ((x + 1) / 2).contiguous()
    g_feat_lm = torch.randn(16, n, 2, device=DEV)
    tabs = [torch.zeros(16 * T, 2, device=DEV) for _ in range(2)]
    be.bwd_jac(g_feat_lm, g_dydx, x01, offs, tabs[0], n, 3, 2, 16, 0.45, 16, level_major=True)
    be.bwd_jac(g_feat_lm, None, x01, offs, tabs[1], n, 3, 2, 16, 0.45, 16, level_major=True, rank1=(uxh, g_grad, jac))
    # (atomic accumulation order differs between launches: compare to rounding, and the cotangent itself exactly)
    assert rel_l2(tabs[1], tabs[0]) < 1e-5
    want_dydx = (jac * uxh.view(n, 16, 1, 2) * g_grad.view(n, 1, 3, 1)).permute(1, 0, 2, 3).reshape(16, n, 6)
    assert torch.equal(g_dydx, want_dydx), "the stored cotangent is the rank-one product"
    untouched = torch.full((16, n, 6), 5.0, device=DEV)
    be.trunk_rr_bwd_grad(x, dydx, g_grad, uxh, idx, rr, packed, H0t, H1t, U0t, tp(), tp(), tp(), tp(), torch.zeros(n, 80, device=DEV, dtype=bf), None, jac)
    assert float(untouched.min()) == 5.0
    gy = torch.zeros(n, 32, device=DEV, dtype=bf)
    gy[:, :K] = g_y.to(bf)
    g_feat = torch.empty(16, n, 2, device=DEV)
    be.trunk_rr_bwd_value(gy, rr, H0t, H1t, A0pt, A1pt, A0t, A1t, g_feat, n)
    ref2 = dict(ref)
    # the value part on the kernel's own primes
    f4 = dict(f3)
    r2 = R.backward(f4, W, g_y, g_grad, jac)
    e = {"a1": rel(R.tp_decode(A1t, n), r2["a1"]), "a0": rel(R.tp_decode(A0t, n), r2["a0"]), "g_feat": rel(g_feat, r2["g_feat"])}
    print("PARITY rr bwd_value", e)
    assert max(e.values()) < 2e-2, e
    # without a gradient cotangent: the primes are absent
    A0n, A1n, g_featn = tp(), tp(), torch.empty(16, n, 2, device=DEV)
    be.trunk_rr_bwd_value(gy, rr, H0t, H1t, None, None, A0n, A1n, g_featn, n)
    r0 = R.backward(f4, W, g_y, None, jac)
    e = {"a1": rel(R.tp_decode(A1n, n), r0["a1"]), "a0": rel(R.tp_decode(A0n, n), r0["a0"]), "g_feat": rel(g_featn, r0["g_feat"])}
    print("PARITY rr bwd_value (no gradient cotangent)", e)
    assert max(e.values()) < 2e-2, e
    # ---- weight gradients: one launch, against fp32 products of the kernels' own operands
    tiles = M // 32
    S = max(d for d in range(1, 65) if tiles % d == 0)
    A1d, A0d, U0bd, U1bd = R.tp_decode(A1t, n), R.tp_decode(A0t, n), R.tp_decode(U0bt, n), R.tp_decode(U1bt, n)
    parts = be.wgrad_pairs([((256, 256), S, (A1t, H0t), (V1t, U0bt)), ((256, 80), S, (A0t, Xp), (V0t, UXb)), ((32, 256), S, (gy, H1t), (onehot, U1bt))], n)
    dW1, dW0, dW2 = (p.float().sum(0) for p in parts)
    want1 = A1d.t() @ f3["h0"] + f3["v1"].t() @ U0bd
    want0 = A0d.t() @ Xp.float() + f3["v0"].t() @ UXb.float()
    want2 = gy.float().t() @ f3["h1"] + onehot.float().t() @ U1bd
    e = {"dW1": rel_l2(dW1, want1), "dW0": rel_l2(dW0[:, :80], want0), "dW2": rel_l2(dW2, want2), "dW0 pad": float(dW0[:, 80:].abs().max())}
    print("PARITY rr wgrad_pairs", e)
    assert e["dW1"] < 1e-2 and e["dW0"] < 1e-2 and e["dW2"] < 1e-2 and e["dW0 pad"] == 0, e
    single = be.wgrad_pairs([((256, 256), S, (A1t, H0t), None)], n)[0].float().sum(0)
    assert rel_l2(single, A1d.t() @ f3["h0"]) < 1e-2
    # ragged cuts (slice = ceil(tiles / S) tiles; short and EMPTY trailing slices write zero partials), a different count per job
    for cuts in [(3, 5, 7), (tiles - 1 if tiles > 2 else 1, 2, max(1, tiles // 2 + 1))]:
        parts = be.wgrad_pairs([((256, 256), cuts[0], (A1t, H0t), (V1t, U0bt)), ((256, 80), cuts[1], (A0t, Xp), (V0t, UXb)),
                                ((32, 256), cuts[2], (gy, H1t), (onehot, U1bt))], n)
        r1, r0, r2 = (p.float().sum(0) for p in parts)
        assert rel_l2(r1, want1) < 1e-2 and rel_l2(r0[:, :80], want0) < 1e-2 and rel_l2(r2, want2) < 1e-2, cuts
    # column sums of the first pair's A operand beside the product (hsWgradPairJob::colsum: the bias gradients of a backward pass), with and
    # without a second pair; an 8-k-step tile-packed B operand (the colour branch's assembled inputs: kind 256 x 128 "tp")
    for second in (True, False):
        cs = []
        parts = be.wgrad_pairs([((256, 256, "colsum"), 5, (A1t, H0t), (V1t, U0bt) if second else None)], n, colsum_out=cs)
        assert rel_l2(parts[0].float().sum(0), want1 if second else A1d.t() @ f3["h0"]) < 1e-2
        assert cs[0].shape == (5, 256) and rel_l2(cs[0].sum(0), A1d.sum(0)) < 2e-3, "column sums of A0 only (fp32 accumulation of bf16 values)"
    # the two forms of the row stream (LDS-DMA stages / register-staged chunks) add the same products in the same order: equal partials
    for cuts in [(4, 3, 2), (tiles, tiles, tiles)]:
        both = []
        for tag in (("consecutive",), ("reg",)):
            cs = []
            parts = be.wgrad_pairs([((256, 256, "colsum") + tag, cuts[0], (A1t, H0t), (V1t, U0bt)), ((256, 80, "colsum") + tag, cuts[1], (A0t, Xp), (V0t, UXb)),
                                    ((32, 256) + tag, cuts[2], (gy, H1t), (onehot, U1bt))], n, colsum_out=cs)
            both.append([p.float() for p in parts] + [c for c in cs if c is not None])
        for a, b in zip(*both):
            assert rel_l2(a.sum(0), b.sum(0)) < 1e-5 and torch.equal(a, b), "LDS-DMA form differs from the register form"
    g = torch.Generator().manual_seed(9)
    xa = (torch.randn(M, 128, generator=g) * 0.5).to(DEV)
    xa[n:] = 0
    XAt = xa.view(tiles, 32, 8, 2, 2, 4).permute(0, 2, 4, 1, 3, 5).contiguous().to(torch.bfloat16).view(-1)   # [tile, s, h, row, e >> 2, e & 3]
    xa_bf = xa.to(torch.bfloat16).float()
    cs = []
    parts = be.wgrad_pairs([((256, 128, "tp", "colsum"), 7, (A0t, XAt), None)], n, colsum_out=cs)
    assert rel_l2(parts[0].float().sum(0), A0d.t() @ xa_bf[:n]) < 1e-2 and rel_l2(cs[0].sum(0), A0d.sum(0)) < 2e-3
    reg = be.wgrad_pairs([((256, 128, "tp", "reg"), 7, (A0t, XAt), None)], n)
    con = be.wgrad_pairs([((256, 128, "tp", "consecutive"), 7, (A0t, XAt), None)], n)
    assert torch.equal(reg[0], con[0]) and rel_l2(reg[0].float().sum(0), parts[0].float().sum(0)) < 6e-3      # (bf16 partials, another cut)


@pytest.mark.parametrize("n,K", [(1000, 32), (4099, 5), (64, 12)])
def test_rr_output_cotangent_image(n, K):
    """hs_trunk_rr_gy: gy = bf16(g_raw + g_sdf at the arg-min column), zero beyond K; per-block column sums add up to the bias gradient;
    either cotangent may be absent."""
    from holoscene_amd.hashencoder.backend import _backend as be
    g = torch.Generator().manual_seed(n + K)
    g_raw, g_sdf = torch.randn(n, K, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    idx = torch.randint(0, K, (n,), generator=g).to(DEV)
    for raw, sdf in ((g_raw, g_sdf), (g_raw, None), (None, g_sdf)):
        gy = torch.full((n, 32), 7.0, device=DEV, dtype=torch.bfloat16)
        part = torch.full((be.RR_GY_BLOCKS, 32), 7.0, device=DEV)
        be.trunk_rr_gy(raw, sdf, idx, K, gy, part)
        want = torch.zeros(n, 32, device=DEV)
        if raw is not None:
            want[:, :K] = raw
        if sdf is not None:
            want[torch.arange(n, device=DEV), idx] += sdf
        assert torch.equal(gy, want.to(torch.bfloat16)), "gy is the bf16 rounding of the fp32 sum"
        assert torch.allclose(part.sum(0), want.sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("n,K,Be", [(1000, 32, 96), (4099, 5, 513), (300, 40, 64)])
def test_output_cotangent_images_in_one_launch_equal_the_two_launches(n, K, Be):
    """hs_trunk_rr_gy_split: gy / its column sums of the rendered samples and the [4 Be, KP] cotangent image of the Eikonal points' value+Jacobian rows
    from ONE launch, bit for bit what hs_trunk_rr_gy and hs_trunk_split_bwd write as two."""
    from holoscene_amd.hashencoder.backend import _backend as be
    g = torch.Generator().manual_seed(n + K + Be)
    KP, planes = (32, ()) if K <= 32 else (64, (2,))
    g_raw, g_sdf = torch.randn(n, K, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    idx = torch.randint(0, K, (n + Be, 1), generator=g).to(DEV)
    g_yeik, g_min, g_theta = torch.randn(Be, K, generator=g).to(DEV), torch.randn(Be, 1, generator=g).to(DEV), torch.randn((K + 1) * Be, 3, generator=g).to(DEV)
    bf = torch.bfloat16
    gy_a, gy_b = (torch.full(planes + (n, 32), 7.0, device=DEV, dtype=bf) for _ in range(2))
    pa, pb = (torch.full((be.RR_GY_BLOCKS, KP), 7.0, device=DEV) for _ in range(2))
    img_a, img_b = (torch.full((4 * Be, KP), 7.0, device=DEV, dtype=bf) for _ in range(2))
    be.trunk_rr_gy(g_raw, g_sdf, idx[:n], K, gy_a, pa)
    be.trunk_split_bwd(None, None, idx[n:], None, g_yeik, g_min, g_theta, Be, 0, K, img_a)
    be.trunk_rr_gy_split(g_raw, g_sdf, idx[:n], K, gy_b, pb, idx[n:].reshape(-1), g_yeik, g_min, g_theta, img_b)
    assert torch.equal(gy_a.view(torch.int16), gy_b.view(torch.int16)) and torch.equal(pa, pb)
    assert torch.equal(img_a.view(torch.int16), img_b.view(torch.int16))
    # absent cotangents (either family's may be missing in a pass)
    be.trunk_rr_gy(None, g_sdf, idx[:n], K, gy_a, pa)
    be.trunk_split_bwd(None, None, idx[n:], None, None, None, g_theta, Be, 0, K, img_a)
    be.trunk_rr_gy_split(None, g_sdf, idx[:n], K, gy_b, pb, idx[n:].reshape(-1), None, None, g_theta, img_b)
    assert torch.equal(gy_a.view(torch.int16), gy_b.view(torch.int16)) and torch.equal(pa, pb) and torch.equal(img_a.view(torch.int16), img_b.view(torch.int16))


def test_fused_forward_equals_the_pair_at_the_benchmarked_size():
    """k_rr_fwd's chunk pipeline (counted vector-memory waits, bare barriers, two rounds of tiles per workgroup) at 100 352 samples:
    every output bit-identical to k_rr_fwd_value + k_rr_fwd_grad, five runs in a row (a race in the pipeline would show as a run
    that differs)."""
    from holoscene_amd.hashencoder.backend import _backend as be
    n, K = 100352, 32
    x, feat, dydx, W = _problem(n, K, 11)
    W0, b0, W1, b1, W2, b2 = W
    bf = torch.bfloat16
    packed = be.sdf_mlp2_pack(W0, b0, W1, b1, W2, b2, K, log2_domain=False)
    rr = be.trunk_rr_pack(W0, W1, W2, K)
    M = be.tp_rows(n)

    def buffers():
        tp = lambda: torch.full((M * 256,), 7.0, device=DEV, dtype=bf)  # noqa: E731
        return dict(H0t=tp(), H1t=tp(), U0t=tp(), V1t=tp(), V0t=tp(), Xp=torch.zeros(n, 80, device=DEV, dtype=bf), onehot=torch.zeros(n, 32, device=DEV, dtype=bf),
                    raw=torch.empty(n, K, device=DEV), sdf=torch.empty(n, device=DEV), idx=torch.empty(n, device=DEV, dtype=torch.int64),
                    grad=torch.empty(n, 3, device=DEV), uxh=torch.empty(n, 32, device=DEV))
    a = buffers()
    be.trunk_rr_fwd_value(x, feat, packed, K, a["H0t"], a["H1t"], a["Xp"], a["raw"], a["sdf"], a["idx"], a["onehot"])
    be.trunk_rr_fwd_grad(x, dydx, a["idx"], rr, a["H0t"], a["H1t"], a["U0t"], a["V1t"], a["V0t"], a["grad"], a["uxh"], 0.5)
    for run in range(5):
        b = buffers()
        be.trunk_rr_fwd(x, feat, dydx, packed, rr, K, b["H0t"], b["H1t"], b["Xp"], b["raw"], b["sdf"], b["idx"], b["onehot"], b["U0t"], b["V1t"], b["V0t"],
                        b["grad"], b["uxh"], 0.5)
        diff = [k for k in a if not torch.equal(a[k].view(torch.int16) if a[k].dtype == bf else a[k], b[k].view(torch.int16) if b[k].dtype == bf else b[k])]
        assert not diff, f"run {run}: fused forward differs from the pair in {diff}"


@pytest.mark.parametrize("n,K", [(1000, 64), (4099, 40), (37, 33)])
def test_rr_wide_kernels_vs_torch_restatement(n, K):
    """33..64 objects (confs/custom/siebelgame: d_out = 64) on the reverse-over-reverse kernels: k_rr_fwd<true> (second output tile, arg-min over both,
    one-hot image in two planes, the arg-min row of W2 from the 64-row table), hs_trunk_rr_gy's two-plane image, k_rr_bwd_value<., true> and the two
    32-column weight-gradient jobs -- each against the torch restatement (tests/rr_reference.py), as the K <= 32 test above does kernel by kernel."""
    from holoscene_amd.hashencoder.backend import _backend as be
    x, feat, dydx, W = _problem(n, K, n + K)
    W0, b0, W1, b1, W2, b2 = W
    jac, bf = 0.5, torch.bfloat16
    packed, packed_b, rr, W2Tf_b = be.trunk_pack_wide(W0, b0, W1, b1, W2, b2, K)
    assert torch.equal(rr[3].view(64, 256)[:K], W2), "the gather table holds W2's rows in fp32"
    M = be.tp_rows(n)
    tp = lambda: torch.full((M * 256,), 7.0, device=DEV, dtype=bf)  # noqa: E731
    H0t, H1t, U0t, V1t, V0t = tp(), tp(), tp(), tp(), tp()
    Xp, onehot = torch.zeros(n, 80, device=DEV, dtype=bf), torch.full((2, n, 32), 7.0, device=DEV, dtype=bf)
    sdf_raw, sdf, idx = torch.empty(n, K, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64)
    grad, uxh = torch.empty(n, 3, device=DEV), torch.empty(n, 32, device=DEV)
    be.trunk_rr_fwd_wide(x, feat, dydx, packed, packed_b, rr, K, H0t, H1t, Xp, sdf_raw, sdf, idx, onehot, U0t, V1t, V0t, grad, uxh, jac)
    f = R.forward(x, feat, dydx, W, jac)
    e = {"y": rel(sdf_raw, f["y"]), "h0": rel(R.tp_decode(H0t, n), f["h0"]), "h1": rel(R.tp_decode(H1t, n), f["h1"])}
    print(f"PARITY rr wide K={K} n={n} forward values", e)
    assert e["y"] < 1e-2 and e["h0"] < 1e-2 and e["h1"] < 1.5e-2, e
    assert torch.equal(sdf, sdf_raw.min(-1)[0]) and torch.equal(idx, sdf_raw.min(-1)[1]), "minimum / arg-min over BOTH tiles of the kernel's own outputs"
    if K >= 40:
        assert float((idx >= 32).float().mean()) > 0.05 and float((idx < 32).float().mean()) > 0.05, "the problem exercises both tiles"
    oh = torch.nn.functional.one_hot(idx, 64).float()
    assert torch.equal(onehot[0].float(), oh[:, :32]) and torch.equal(onehot[1].float(), oh[:, 32:])
    f2 = dict(f)
    f2["idx"] = idx
    f2["h0"], f2["h1"] = R.tp_decode(H0t, n), R.tp_decode(H1t, n)
    f2["s0"], f2["s1"] = 1 - torch.exp(-100 * f2["h0"]), 1 - torch.exp(-100 * f2["h1"])
    v1 = R.bfr(W2[idx] * f2["s1"], True)
    u0f = v1 @ R.bfr(W1, True)
    f2.update(v1=v1, u0f=u0f, u0=R.bfr(u0f, True), v0=R.bfr(u0f * f2["s0"], True))
    f2["ux"] = f2["v0"] @ R.bfr(W0, True)
    f2["uxh"] = f2["ux"][:, R.NPE:]
    f2["grad"] = torch.einsum("bj,bjd->bd", f2["ux"], f["E"])
    e = {"v1": rel(R.tp_decode(V1t, n), f2["v1"]), "u0": rel(R.tp_decode(U0t, n), f2["u0"]), "v0": rel(R.tp_decode(V0t, n), f2["v0"]),
         "uxh": rel(uxh, f2["uxh"]), "grad": rel(grad, f2["grad"])}
    print(f"PARITY rr wide K={K} forward gradient chain", e)
    assert max(e.values()) < 1.5e-2, e
    # ---- the first tile's arithmetic is the 32-object kernel's: with objects 32.. pushed out of reach every tile-packed output is bit-identical to it
    W2far, b2far = W2.clone(), b2.clone()
    b2far[32:] += 1e3
    pk, pkb, rrf, _ = be.trunk_pack_wide(W0, b0, W1, b1, W2far, b2far, K)
    Fw = {k: tp() for k in ("H0t", "H1t", "U0t", "V1t", "V0t")}
    rawf, sdff, idxf, ohf = torch.empty(n, K, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64), torch.zeros(2, n, 32, device=DEV, dtype=bf)
    gradf, uxhf = torch.empty(n, 3, device=DEV), torch.empty(n, 32, device=DEV)
    be.trunk_rr_fwd_wide(x, feat, dydx, pk, pkb, rrf, K, Fw["H0t"], Fw["H1t"], torch.zeros(n, 80, device=DEV, dtype=bf), rawf, sdff, idxf, ohf, Fw["U0t"], Fw["V1t"],
                         Fw["V0t"], gradf, uxhf, jac)
    p32 = be.sdf_mlp2_pack(W0, b0, W1, b1, W2[:32].contiguous(), b2[:32].contiguous(), 32, log2_domain=False)
    r32 = be.trunk_rr_pack(W0, W1, W2[:32].contiguous(), 32)
    Fn = {k: tp() for k in ("H0t", "H1t", "U0t", "V1t", "V0t")}
    rawn, sdfn, idxn, ohn = torch.empty(n, 32, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV, dtype=torch.int64), torch.zeros(n, 32, device=DEV, dtype=bf)
    gradn, uxhn = torch.empty(n, 3, device=DEV), torch.empty(n, 32, device=DEV)
    be.trunk_rr_fwd(x, feat, dydx, p32, r32, 32, Fn["H0t"], Fn["H1t"], torch.zeros(n, 80, device=DEV, dtype=bf), rawn, sdfn, idxn, ohn, Fn["U0t"], Fn["V1t"], Fn["V0t"],
                    gradn, uxhn, jac)
    same = {k: torch.equal(Fw[k].view(torch.int16), Fn[k].view(torch.int16)) for k in Fw}
    same.update(raw=torch.equal(rawf[:, :32], rawn), sdf=torch.equal(sdff, sdfn), idx=torch.equal(idxf, idxn), grad=torch.equal(gradf, gradn), uxh=torch.equal(uxhf, uxhn),
                onehot=torch.equal(ohf[0].view(torch.int16), ohn.view(torch.int16)) and float(ohf[1].float().abs().max()) == 0.0)
    assert all(same.values()), f"two-tile kernel differs from the 32-object kernel where the second tile cannot win: {same}"
    # ---- backward: gradient part (the 32-object kernel on the 64-row table), cotangent image, value part
    g = torch.Generator().manual_seed(5)
    g_grad = torch.randn(n, 3, generator=g).to(DEV)
    g_y = (torch.randn(n, K, generator=g) * 0.5).to(DEV)
    f3 = dict(f2)
    f3["u0"], f3["v0"], f3["v1"], f3["uxh"] = R.tp_decode(U0t, n), R.tp_decode(V0t, n), R.tp_decode(V1t, n), uxh
    ref = R.backward(f3, W, g_y, g_grad, jac)
    U0bt, A0pt, A1pt, U1bt, A0t, A1t = tp(), tp(), tp(), tp(), tp(), tp()
    UXb = torch.zeros(n, 80, device=DEV, dtype=bf)
    be.trunk_rr_bwd_grad(x, dydx, g_grad, uxh, idx, rr, packed, H0t, H1t, U0t, U0bt, A0pt, A1pt, U1bt, UXb, None, jac)
    e = {"u0b": rel(R.tp_decode(U0bt, n), ref["u0b"]), "a0p": rel(R.tp_decode(A0pt, n), ref["a0p"]), "u1b": rel(R.tp_decode(U1bt, n), ref["u1b"]),
         "a1p": rel(R.tp_decode(A1pt, n), ref["a1p"])}
    print(f"PARITY rr wide K={K} bwd_grad", e)
    assert max(e.values()) < 2e-2, e
    gy = torch.full((2, n, 32), 7.0, device=DEV, dtype=bf)
    part = torch.full((be.RR_GY_BLOCKS, 64), 7.0, device=DEV)
    g_sdf = torch.randn(n, generator=g).to(DEV)
    be.trunk_rr_gy(g_y, g_sdf, idx, K, gy, part)
    want_gy = torch.zeros(n, 64, device=DEV)
    want_gy[:, :K] = g_y
    want_gy[torch.arange(n, device=DEV), idx] += g_sdf
    assert torch.equal(gy[0], want_gy[:, :32].to(bf)) and torch.equal(gy[1], want_gy[:, 32:].to(bf)), "gy planes = bf16(g_raw + g_sdf at the arg-min column)"
    assert torch.allclose(part.sum(0), want_gy.sum(0), rtol=1e-4, atol=1e-3)
    g_yt = want_gy[:, :K].to(bf).float()          # the cotangent the value part sees (minimum's folded in, rounded to bf16)
    g_feat = torch.empty(16, n, 2, device=DEV)
    be.trunk_rr_bwd_value_wide(gy, rr, W2Tf_b, H0t, H1t, A0pt, A1pt, A0t, A1t, g_feat, n)
    r2 = R.backward(f3, W, g_yt, g_grad, jac)
    e = {"a1": rel(R.tp_decode(A1t, n), r2["a1"]), "a0": rel(R.tp_decode(A0t, n), r2["a0"]), "g_feat": rel(g_feat, r2["g_feat"])}
    print(f"PARITY rr wide K={K} bwd_value", e)
    assert max(e.values()) < 2e-2, e
    A0n, A1n, g_featn = tp(), tp(), torch.empty(16, n, 2, device=DEV)
    be.trunk_rr_bwd_value_wide(gy, rr, W2Tf_b, H0t, H1t, None, None, A0n, A1n, g_featn, n)
    r0 = R.backward(f3, W, g_yt, None, jac)
    e = {"a1": rel(R.tp_decode(A1n, n), r0["a1"]), "a0": rel(R.tp_decode(A0n, n), r0["a0"]), "g_feat": rel(g_featn, r0["g_feat"])}
    assert max(e.values()) < 2e-2, e
    # ---- the last layer's weight gradient as two 32-row jobs of one launch
    tiles = M // 32
    S = max(d for d in range(1, 65) if tiles % d == 0)
    parts = be.wgrad_pairs([((32, 256), S, (gy[0], H1t), (onehot[0], U1bt)), ((32, 256), S, (gy[1], H1t), (onehot[1], U1bt))], n)
    dW2 = torch.cat([p.float().sum(0) for p in parts])
    gyf = torch.cat([gy[0], gy[1]], 1).float()
    want2 = gyf.t() @ f3["h1"] + oh.t() @ R.tp_decode(U1bt, n)
    e2 = rel_l2(dW2, want2)
    print(f"PARITY rr wide K={K} dW2 (two 32-row jobs) relL2 {e2:.2e}")
    assert e2 < 1e-2 and (K == 64 or float(dW2[K:].abs().max()) == 0.0)
