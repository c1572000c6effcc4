"""GPU: HIP hash-grid kernels (through the C ABI) vs the C oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from helpers import load
from oracle import hash_oracle

pytestmark = pytest.mark.gpu


def _be():
    from holoscene_amd.hashencoder.backend import _backend
    return _backend


def _cfg(r):
    return float(r["S"]), int(r["base"])


@pytest.mark.parametrize("name", ["hash_small", "hash_mid"])
def test_reference_abi_vs_golden(name):
    """hs_hash_encode_{forward,backward,second_backward}: reference layouts ([L,B,C], dy_dx [B,L*D*C])."""
    r = load(name)
    dev = "cuda"
    x, emb, offs = (torch.from_numpy(r[k]).to(dev) for k in ("x", "emb", "offsets"))
    S, H = _cfg(r)
    B, D = x.shape
    C, L = emb.shape[1], offs.numel() - 1
    out = torch.empty(L, B, C, device=dev)
    dydx = torch.empty(B, L * D * C, device=dev)
    _be().hash_encode_forward(x, emb, offs, out, B, D, C, L, S, H, True, dydx)
    assert torch.equal(out.cpu(), torch.from_numpy(r["out"])), "forward must be bit-exact"
    assert torch.equal(dydx.cpu(), torch.from_numpy(r["dydx"])), "dy_dx must be bit-exact"
    grad = torch.from_numpy(r["grad"]).to(dev)
    gemb = torch.zeros_like(emb)
    gx = torch.zeros_like(x)
    _be().hash_encode_backward(grad, x, emb, offs, gemb, B, D, C, L, S, H, True, dydx, gx)
    assert torch.equal(gx.cpu(), torch.from_numpy(r["grad_x"])), "input gradient must be bit-exact"
    # scatter: float atomics, order differs -> tolerance scaled by the number of contributors
    ref = torch.from_numpy(r["grad_emb"])
    tol = 1e-6 * np.sqrt(B) * float(ref.abs().max())
    assert (gemb.cpu() - ref).abs().max() <= tol
    ggx = torch.from_numpy(r["ggx"]).to(dev)
    gg = torch.zeros_like(grad)
    g2 = torch.zeros_like(emb)
    _be().hash_encode_second_backward(grad, x, emb, offs, B, D, C, L, S, H, True, dydx, ggx, gg, g2)
    assert torch.equal(gg.cpu(), torch.from_numpy(r["grad_grad"]))
    ref2 = torch.from_numpy(r["grad2_emb"])
    assert (g2.cpu() - ref2).abs().max() <= 1e-6 * np.sqrt(B) * float(ref2.abs().max())


@pytest.mark.parametrize("schedule", [0, 1])
@pytest.mark.parametrize("cfg", [dict(L=8, base=16, end=256, logmap=13, B=3001), dict(L=16, base=16, end=2048, logmap=19, B=20000)])
def test_strided_variants_vs_oracle(cfg, schedule, monkeypatch):
    """Point-major features / level-major dy_dx, both block schedules, ragged B, incl. OOB points."""
    from holoscene_amd.hashencoder import backend
    monkeypatch.setattr(backend, "SCHEDULE", schedule)
    g = torch.Generator().manual_seed(7)
    L, base = cfg["L"], cfg["base"]
    pls = hash_oracle.per_level_scale_for(base, cfg["end"], L)
    offs = torch.from_numpy(hash_oracle.level_offsets(L, base, pls, cfg["logmap"]))
    emb = (torch.rand(int(offs[-1]), 2, generator=g) * 2 - 1)
    B = cfg["B"]
    x = torch.rand(B, 3, generator=g) * 1.1 - 0.05
    x[:3] = torch.tensor([[0.0, 0, 0], [1.0, 1, 1], [1.0, 0.5, 0.0]])
    S = float(np.log2(pls))
    ref_out, ref_dydx = hash_oracle.fwd(x, emb, offs, S, base, True)
    dev = "cuda"
    xd, ed, od = x.to(dev), emb.to(dev), offs.to(dev)
    out = torch.empty(B, L * 2, device=dev)
    dydx = torch.empty(L, B, 6, device=dev)
    _be().fwd(xd, ed, od, out, B, 3, 2, L, S, base, dydx)
    assert torch.equal(out.cpu(), ref_out.permute(1, 0, 2).reshape(B, -1))
    assert torch.equal(dydx.cpu(), ref_dydx.view(B, L, 6).permute(1, 0, 2))
    out2 = torch.empty_like(out)
    _be().fwd(xd, ed, od, out2, B, 3, 2, L, S, base, None)      # no dy_dx variant
    assert torch.equal(out2, out)
    grad = torch.randn(L, B, 2, generator=g)
    ref_gx, ref_ge = hash_oracle.bwd(grad, x, emb, offs, S, base, True, ref_dydx)
    gpm = grad.permute(1, 0, 2).reshape(B, -1).contiguous().to(dev)
    ge = torch.zeros_like(ed)
    gx = torch.empty_like(xd)
    _be().bwd(gpm, xd, od, ge, B, 3, 2, L, S, base, dydx, gx)
    assert torch.equal(gx.cpu(), ref_gx)
    assert (ge.cpu() - ref_ge).abs().max() <= 2e-6 * np.sqrt(B) * float(ref_ge.abs().max())
    gx_only = torch.empty_like(xd)
    _be().bwd(gpm, xd, od, None, B, 3, 2, L, S, base, dydx, gx_only)  # scatter skipped
    assert torch.equal(gx_only, gx)
    ggx = torch.randn(B, 3, generator=g)
    ref_gg, ref_g2 = hash_oracle.bwd2(grad, x, emb, offs, S, base, ref_dydx, ggx)
    gg = torch.empty_like(gpm)
    g2 = torch.zeros_like(ed)
    _be().bwd2(gpm, xd, od, B, 3, 2, L, S, base, dydx, ggx.to(dev), gg, g2)
    assert torch.equal(gg.cpu(), ref_gg.permute(1, 0, 2).reshape(B, -1))
    assert (g2.cpu() - ref_g2).abs().max() <= 2e-6 * np.sqrt(B) * float(ref_g2.abs().max())


@pytest.mark.parametrize("D,C", [(2, 1), (2, 2), (3, 1), (3, 4), (3, 8), (2, 8)])
def test_other_dims_vs_oracle(D, C):
    g = torch.Generator().manual_seed(D * 10 + C)
    L, base = 5, 8
    pls = 1.5
    offs = torch.from_numpy(hash_oracle.level_offsets(L, base, pls, 11, D))
    emb = torch.rand(int(offs[-1]), C, generator=g) - 0.5
    B = 777
    x = torch.rand(B, D, generator=g)
    S = float(np.log2(pls))
    ref_out, ref_dydx = hash_oracle.fwd(x, emb, offs, S, base, True)
    dev = "cuda"
    out = torch.empty(L, B, C, device=dev)
    dydx = torch.empty(B, L * D * C, device=dev)
    _be().hash_encode_forward(x.to(dev), emb.to(dev), offs.to(dev), out, B, D, C, L, S, base, True, dydx)
    assert torch.equal(out.cpu(), ref_out) and torch.equal(dydx.cpu(), ref_dydx)
    grad = torch.randn(L, B, C, generator=g)
    ref_gx, ref_ge = hash_oracle.bwd(grad, x, emb, offs, S, base, True, ref_dydx)
    ge = torch.zeros(emb.shape, device=dev)
    gx = torch.zeros(B, D, device=dev)
    _be().hash_encode_backward(grad.to(dev), x.to(dev), emb.to(dev), offs.to(dev), ge, B, D, C, L, S, base, True, dydx, gx)
    assert torch.equal(gx.cpu(), ref_gx)
    assert (ge.cpu() - ref_ge).abs().max() <= 2e-6 * np.sqrt(B) * float(ref_ge.abs().max())


def test_empty_batch_and_bad_args():
    dev = "cuda"
    offs = torch.from_numpy(hash_oracle.level_offsets(4, 4, 2.0, 10)).to(dev)
    emb = torch.zeros(int(offs[-1]), 2, device=dev)
    out = torch.empty(0, 8, device=dev)
    _be().fwd(torch.empty(0, 3, device=dev), emb, offs, out, 0, 3, 2, 4, 1.0, 4, None)
    with pytest.raises(RuntimeError):
        _be().fwd(torch.empty(4, 3, device=dev), emb.double(), offs, torch.empty(4, 8, device=dev), 4, 3, 2, 4, 1.0, 4, None)
    with pytest.raises(RuntimeError):
        _be().fwd(torch.empty(4, 5, device=dev), emb, offs, torch.empty(4, 8, device=dev), 4, 5, 2, 4, 1.0, 4, None)


def test_full_size_properties():
    """BASELINE config-2 size (131 072 sampler points, stock 16-level grid): size-independent checks."""
    from holoscene_amd.hashencoder import HashEncoder
    torch.manual_seed(0)
    enc = HashEncoder(desired_resolution=2048).cuda()
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    B = 1024 * 128
    x = (torch.rand(B, 3, device="cuda") * 2 - 1)
    y1 = enc(x)
    assert y1.shape == (B, 32) and torch.isfinite(y1).all()
    # linear in the table: enc_{2E}(x) == 2 enc_E(x) exactly (power-of-two scaling)
    with torch.no_grad():
        enc.embeddings.mul_(2)
    assert torch.equal(enc(x), 2 * y1)
    # points outside the cube give exactly zero
    far = x.clone()
    far[::2, 0] = 1.5
    yz = enc(far)
    assert (yz[::2] == 0).all() and torch.equal(yz[1::2], enc(x)[1::2])
    # adjoint: <enc(x), g> == <E, dE>
    g = torch.randn_like(y1)
    xr = x.clone().requires_grad_(True)
    y = enc(xr)
    (ge,) = torch.autograd.grad(y, enc.embeddings, g, retain_graph=True)
    lhs = (y.double() * g.double()).sum()
    rhs = (enc.embeddings.double() * ge.double()).sum()
    assert abs(lhs - rhs) <= 1e-4 * abs(lhs)
    # d/dx against central differences (interior points of a coarse-only table), via double-backward plumbing
    (gx,) = torch.autograd.grad(y, xr, g, create_graph=True)
    assert gx.shape == x.shape and torch.isfinite(gx).all()
    v = torch.randn_like(gx)
    (g2,) = torch.autograd.grad((gx * v).sum(), enc.embeddings)
    # (gx . v) is linear in E: <E, d(gx.v)/dE> == gx.v
    assert abs((enc.embeddings.double() * g2.double()).sum() - (gx.double() * v.double()).sum()) <= 1e-4 * abs((gx.double() * v.double()).sum())


def test_full_size_binned_scatter_properties():
    """Binned scatter at BASELINE config-2 size (stock grid, 100 352 ray-ordered points + level-major cotangents): the adjoint
    identity <enc(x), g> == <E, dE> that ties it to the forward kernel, equality with the atomic path, and accumulation on top
    of existing content -- size-independent properties, no oracle involved."""
    from holoscene_amd.hashencoder import HashEncoder
    torch.manual_seed(1)
    enc = HashEncoder(desired_resolution=2048).cuda()
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    be = _be()
    R, N = 1024, 98
    o = torch.rand(R, 1, 3, device="cuda") * 0.2 + 0.4
    d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device="cuda"), dim=-1)
    z = torch.sort(torch.rand(R, N, 1, device="cuda") ** 3 * 0.5, 1)[0]        # crowded near the origin of each ray, like real samples
    x = (o + z * d).reshape(-1, 3).clamp(0, 1).contiguous()
    B, L, C = x.shape[0], 16, 2
    S, H = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
    feat = torch.empty(L, B, C, device="cuda")
    be.fwd(x, enc.embeddings, enc.offsets, feat, B, 3, C, L, S, H, None, level_major=True)
    g = torch.randn(L, B, C, device="cuda")
    g[:, ::5] = 0
    ws = be.scatter_workspace(B, 3, C, L, "cuda")
    prior = torch.randn_like(enc.embeddings)
    ge = prior.clone()
    be.bwd(g, x, enc.offsets, ge, B, 3, C, L, S, H, None, None, ws=ws, level_major=True)
    ge_atomic = prior.clone()
    be.bwd(g, x, enc.offsets, ge_atomic, B, 3, C, L, S, H, None, None, level_major=True)
    dE = (ge - prior).double()
    lhs = (feat.double() * g.double()).sum()
    rhs = (enc.embeddings.double() * dE).sum()
    assert abs(lhs - rhs) <= 1e-4 * abs(lhs)
    assert (ge - ge_atomic).abs().max() <= 2e-6 * np.sqrt(B) * float(dE.abs().max())
    # the work space contract (hsHashLayout::ws_clean): the reduce kernel returns every bin counter it consumed to zero, so the next scatter
    # through this (persistent) work space needs no clearing launch -- and a second scatter through it gives the same result
    counts = ws[0][:32 * 128 * 4].view(torch.int32)
    assert int(counts.abs().sum()) == 0
    assert len(ws) > 2 and ws[2], "scatter_workspace hands out persistent, zero-left work spaces"
    ge2 = prior.clone()
    be.bwd(g, x, enc.offsets, ge2, B, 3, C, L, S, H, None, None, ws=be.scatter_workspace(B, 3, C, L, "cuda"), level_major=True)
    assert (ge2 - ge).abs().max() <= 2e-6 * np.sqrt(B) * float(dE.abs().max())
    assert int(counts.abs().sum()) == 0


def _adam_state(dev, n_group0, steps=3):
    from holoscene_amd.hashencoder import backend as B
    st = B.hsAdamState()
    st.step = 0
    st.group_end[0], st.group_end[1] = n_group0, n_group0
    for i, v in enumerate((1e-2, 5e-4, 5e-4)):
        st.lr0[i] = v
        st.lr[i] = v
    state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).to(dev)
    for _ in range(steps):
        _be().adam_tick(state, 0.9, 0.99, 0.9999)
    return state


@pytest.mark.parametrize("case", ["stock", "slab", "small", "jac", "empty", "no_ws"])
def test_reduce_and_step_equals_scatter_then_adam(case):
    """hsTableStep (k_hash_bin_step): the Adam step taken inside the scatter's reduction against the same scatter followed by hs_adam_flat over
    the table -- every entry stepped exactly once (entries without a gradient included), the gradient table all zero afterwards, the bin
    counters returned to zero.  stock: the stock grid, every level through the record bins.  slab: all points in a thin slab, so the dense
    levels' bins overflow into atomics on the gradient table.  small: a grid whose coarse levels are not binned.  jac: through
    hs_hash_bwd_jac.  empty: B = 0 (the table still steps).  no_ws: no work space, every contribution an atomic."""
    from holoscene_amd.hashencoder import HashEncoder
    from holoscene_amd.hashencoder import backend as B_
    torch.manual_seed(5)
    be = _be()
    enc = (HashEncoder(num_levels=8, base_resolution=4, desired_resolution=128, log2_hashmap_size=12) if case == "small"
           else HashEncoder(desired_resolution=2048)).cuda()
    L, C = enc.num_levels, 2
    S, H = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
    n = (enc.embeddings.numel() + 3) // 4 * 4        # (hs_adam_flat walks whole quads; the table itself may end inside one)
    B = 0 if case == "empty" else 60000
    x = torch.rand(max(B, 1), 3, device="cuda")[:B].contiguous()
    if case == "slab":
        x[:, 2] = 0.5 + 0.002 * x[:, 2]
    g = torch.randn(L, B, C, device="cuda")
    g[:, ::7] = 0
    gj = torch.randn(L, B, 3 * C, device="cuda") if case == "jac" else None
    p0 = torch.randn(n, device="cuda") * 0.1
    m0 = torch.randn(n, device="cuda") * 1e-3
    v0 = (0.5 + torch.rand(n, device="cuda")) * 1e-5
    state = _adam_state("cuda", n)

    def scatter(target):
        ws = None if case in ("no_ws", "empty") else be.scatter_workspace(max(B, 1), 3, C, L, "cuda")
        if case == "jac":
            be.bwd_jac(g, gj, x, enc.offsets, target, B, 3, C, L, S, H, ws=ws, level_major=True)
        else:
            be.bwd(g, x, enc.offsets, target, B, 3, C, L, S, H, None, None, ws=ws, level_major=True)
        return ws

    # the plain way
    ge = torch.zeros(n, device="cuda")
    scatter(ge[:enc.embeddings.numel()].view(-1, C))
    p1, m1, v1 = p0.clone(), m0.clone(), v0.clone()
    be.adam_flat(p1, ge, m1, v1, 0, n, state, 0.9, 0.99, 1e-15, 1.0)
    # reduce-and-step
    g2 = torch.zeros(n, device="cuda")
    p2, m2, v2 = p0.clone(), m0.clone(), v0.clone()
    ts = B_.hsTableStep(p2.data_ptr(), m2.data_ptr(), v2.data_ptr(), state.data_ptr(), 0.9, 0.99, 1e-15, 1.0, 0)
    B_.TABLE_STEPS[g2.data_ptr()] = [ts, 0, 1]
    try:
        ws = scatter(g2[:enc.embeddings.numel()].view(-1, C))
        assert B_.TABLE_STEPS[g2.data_ptr()][1] == 1
        with pytest.raises(RuntimeError, match="more gradient producers"):
            scatter(g2[:enc.embeddings.numel()].view(-1, C))
    finally:
        B_.TABLE_STEPS.pop(g2.data_ptr())
    torch.cuda.synchronize()
    assert not bool(g2.any()), "the gradient table is all zero again"
    # two producers of one iteration (the background-patch iteration's geometry table): the first accumulates into the gradient table the plain
    # way, the LAST steps and adds what it finds there (hsTableStep.prior) = scatter twice, then Adam on the sum
    ge2 = torch.zeros(n, device="cuda")
    scatter(ge2[:enc.embeddings.numel()].view(-1, C))
    scatter(ge2[:enc.embeddings.numel()].view(-1, C))
    p3, m3, v3 = p0.clone(), m0.clone(), v0.clone()
    be.adam_flat(p3, ge2, m3, v3, 0, n, state, 0.9, 0.99, 1e-15, 1.0)
    g4 = torch.zeros(n, device="cuda")
    p4, m4, v4 = p0.clone(), m0.clone(), v0.clone()
    ts2 = B_.hsTableStep(p4.data_ptr(), m4.data_ptr(), v4.data_ptr(), state.data_ptr(), 0.9, 0.99, 1e-15, 1.0, 0, 1)
    B_.TABLE_STEPS[g4.data_ptr()] = [ts2, 0, 2]
    try:
        scatter(g4[:enc.embeddings.numel()].view(-1, C))
        assert B_.TABLE_STEPS[g4.data_ptr()][1] == 1 and (B == 0 or bool(g4.any())), "the first producer accumulated the plain way"
        scatter(g4[:enc.embeddings.numel()].view(-1, C))
        assert B_.TABLE_STEPS[g4.data_ptr()][1] == 2
    finally:
        B_.TABLE_STEPS.pop(g4.data_ptr())
    torch.cuda.synchronize()
    assert not bool(g4.any()), "two producers: the gradient table is all zero again"
    kk = enc.embeddings.numel()
    bigg = 1 + float(ge2[:kk].abs().max())
    for name, a, b, tol in (("m", m4[:kk], m3[:kk], 1e-6 * m3[:kk].abs() + 2e-6 * bigg), ("p", p4[:kk], p3[:kk], 4e-5 * bigg)):
        err = (a - b).abs()
        assert bool((err <= tol).all()), (case, "two producers", name, float(err.max()), float((err / tol).max()))
    if ws is not None:
        assert int(ws[0][:32 * 128 * 4].view(torch.int32).abs().sum()) == 0
    if case == "slab":      # the case is only worth its name if bins did overflow
        assert B * 8 > 4 * ws[1]
    # the sums inside a cell are formed in a different order (LDS atomics in both paths: not even run-to-run stable), so the gradient
    # agrees to rounding and the update to what that rounding does to m / sqrt(v)
    # (a gradient element is a sum of up to thousands of O(1) contributions: its rounding is ~1e-6 of its size; m moves by 0.1 of that,
    # v by 0.02 |g| of it, p by step_size / sqrt(v) of m's -- with sqrt(v0) ~ 3e-3 and step_size 0.037 that is ~17x)
    k = enc.embeddings.numel()
    ga = ge[:k].abs()
    big = 1 + float(ga.max())        # (scale of the sums of |contributions|: cancellation makes a cell's rounding exceed 1e-6 of its own value)
    for name, a, b, tol in (("m", m2[:k], m1[:k], 1e-6 * m1[:k].abs() + 1e-6 * big),
                            ("v", v2[:k], v1[:k], 1e-6 * v1[:k].abs() + 1e-7 * ga * big + 1e-10),
                            ("p", p2[:k], p1[:k], 2e-5 * big)):
        err = (a - b).abs()
        assert bool((err <= tol).all()), (case, name, float(err.max()), float((err / tol).max()))
    assert float((p2[:k] - p1[:k]).abs().mean()) < 1e-7
    moved = (p2[:k] != p0[:k]).float().mean()
    assert float(moved) > 0.999, "every entry is stepped (dense Adam), with or without a gradient"


def test_scatter_with_ray_ordered_points_vs_oracle():
    """Consecutive lanes in the same cell exercise the wave-merged scatter (runs of every length,
    runs crossing wave boundaries, OOB lanes splitting runs, a ragged last block)."""
    g = torch.Generator().manual_seed(11)
    L, base, end, logmap = 8, 4, 128, 12
    pls = hash_oracle.per_level_scale_for(base, end, L)
    offs = torch.from_numpy(hash_oracle.level_offsets(L, base, pls, logmap))
    emb = torch.rand(int(offs[-1]), 2, generator=g) - 0.5
    R, N = 37, 98
    o = torch.tensor([0.85, 0.5, 0.5])
    d = torch.nn.functional.normalize(torch.tensor([-1.0, 0, 0]) + (torch.rand(R, 3, generator=g) - 0.5) * 0.6, dim=-1)
    z = torch.sort(torch.rand(R, N, generator=g) ** 3 * 1.3, -1)[0]       # clustered near the origin of the ray
    z[:, 10:20] = z[:, 10:11]                                               # exact duplicates
    x = (o + z[..., None] * d[:, None]).reshape(-1, 3).contiguous()
    B = x.shape[0]
    S = float(np.log2(pls))
    _, ref_dydx = hash_oracle.fwd(x, emb, offs, S, base, True)
    grad = torch.randn(L, B, 2, generator=g)
    ref_gx, ref_ge = hash_oracle.bwd(grad, x, emb, offs, S, base, True, ref_dydx)
    ggx = torch.randn(B, 3, generator=g)
    _, ref_g2 = hash_oracle.bwd2(grad, x, emb, offs, S, base, ref_dydx, ggx)
    dev = "cuda"
    xd, od = x.to(dev), offs.to(dev)
    gpm = grad.permute(1, 0, 2).reshape(B, -1).contiguous().to(dev)
    dydx = ref_dydx.view(B, L, 6).permute(1, 0, 2).contiguous().to(dev)
    ge = torch.zeros(emb.shape, device=dev)
    _be().bwd(gpm, xd, od, ge, B, 3, 2, L, S, base, None, None)
    scale = float(ref_ge.abs().max())
    assert (ge.cpu() - ref_ge).abs().max() <= 1e-5 * scale
    g2 = torch.zeros(emb.shape, device=dev)
    _be().bwd2(gpm, xd, od, B, 3, 2, L, S, base, dydx, ggx.to(dev), None, g2)
    assert (g2.cpu() - ref_g2).abs().max() <= 1e-5 * float(ref_g2.abs().max())
    # fused value+Jacobian scatter == first-backward scatter + rank-one second-backward scatter
    G = (grad[:, :, None, :] * ggx[None, :, :, None]).reshape(L, B, 6).contiguous().to(dev)
    gj = torch.zeros(emb.shape, device=dev)
    _be().bwd_jac(gpm, G, xd, od, gj, B, 3, 2, L, S, base)
    ref = ref_ge + ref_g2
    assert (gj.cpu() - ref).abs().max() <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("cap", [None, 16])
@pytest.mark.parametrize("cfg", [dict(L=16, base=16, end=2048, logmap=19, B=30000), dict(L=8, base=16, end=512, logmap=13, B=5000)])
def test_binned_scatter_vs_oracle(cfg, cap):
    """Record-list + LDS-reduction scatter of the hashed levels (hs_hash_bwd / hs_hash_bwd_jac with a work space) vs the C oracle
    and vs the pure-atomic path; cap=16 forces nearly every record through the bin-overflow fallback.  Same tolerance as the
    atomic path (float summation order differs), and a gradient that is accumulated on top of existing content."""
    g = torch.Generator().manual_seed(11)
    L, base = cfg["L"], cfg["base"]
    pls = hash_oracle.per_level_scale_for(base, cfg["end"], L)
    offs = torch.from_numpy(hash_oracle.level_offsets(L, base, pls, cfg["logmap"]))
    emb = (torch.rand(int(offs[-1]), 2, generator=g) * 2 - 1)
    B = cfg["B"]
    x = torch.rand(B, 3, generator=g) * 1.1 - 0.05
    S = float(np.log2(pls))
    _, ref_dydx = hash_oracle.fwd(x, emb, offs, S, base, True)
    grad = torch.randn(L, B, 2, generator=g)
    grad[:, ::3] = 0          # exact zeros are skipped, not recorded
    _, ref_ge = hash_oracle.bwd(grad, x, emb, offs, S, base, True, ref_dydx)
    dev = "cuda"
    xd, od = x.to(dev), offs.to(dev)
    gpm = grad.permute(1, 0, 2).reshape(B, -1).contiguous().to(dev)
    be = _be()
    ws = be.scatter_workspace(B, 3, 2, L, dev)
    assert ws is not None
    if cap is not None:
        ws = (ws[0], cap)
    prior = torch.randn(int(offs[-1]), 2, generator=g).to(dev)
    ge = prior.clone()
    be.bwd(gpm, xd, od, ge, B, 3, 2, L, S, base, None, None, ws=ws)
    tol = 2e-6 * np.sqrt(B) * float(ref_ge.abs().max())
    assert ((ge - prior).cpu() - ref_ge).abs().max() <= tol
    ge_atomic = prior.clone()
    be.bwd(gpm, xd, od, ge_atomic, B, 3, 2, L, S, base, None, None)
    assert (ge - ge_atomic).abs().max() <= tol
    # value+Jacobian scatter: against the atomic path (itself checked against the oracle elsewhere)
    g_dydx = torch.randn(L, B, 6, generator=g).to(dev)
    a, b = torch.zeros_like(prior), torch.zeros_like(prior)
    be.bwd_jac(gpm, g_dydx, xd, od, a, B, 3, 2, L, S, base, ws=ws)
    be.bwd_jac(gpm, g_dydx, xd, od, b, B, 3, 2, L, S, base)
    assert (a - b).abs().max() <= 4e-6 * np.sqrt(B) * float(b.abs().max())


# ------------------------------------------------------------------------------------ many grids in one launch (hsHashLayout::grid_id)
@pytest.mark.parametrize("order", ["mixed", "sorted"])
@pytest.mark.parametrize("cfg", [dict(L=8, base=16, end=256, logmap=13, B=3001, G=5, C=2), dict(L=16, base=16, end=2048, logmap=19, B=12000, G=3, C=2),
                                 dict(L=6, base=8, end=96, logmap=11, B=2000, G=4, C=4)])
def test_batched_grids_vs_oracle_per_grid(cfg, order):
    """Points of G grids of one geometry in ONE launch each for forward(+dy_dx), scatter, second backward and the value+Jacobian
    scatter: every point's result is the ORACLE's on its own grid's table (forward bit-exact); the same cell of two grids in
    neighbouring lanes must not be merged (ray-like runs of equal points, alternating grids, are part of the batch)."""
    g = torch.Generator().manual_seed(11)
    L, base, G, C, B = cfg["L"], cfg["base"], cfg["G"], cfg["C"], cfg["B"]
    pls = hash_oracle.per_level_scale_for(base, cfg["end"], L)
    offs = torch.from_numpy(hash_oracle.level_offsets(L, base, pls, cfg["logmap"]))
    T = int(offs[-1])
    emb = torch.rand(G, T, C, generator=g) * 2 - 1
    x = torch.rand(B, 3, generator=g) * 1.1 - 0.05
    x[64:192] = x[64:65]                                         # 128 lanes on ONE cell ...
    gid = torch.randint(0, G, (B,), generator=g, dtype=torch.int32)
    gid[64:192] = (torch.arange(128) // 3 % G).to(torch.int32)  # ... in runs of three lanes per grid
    if order == "sorted":
        gid, perm = torch.sort(gid, stable=True)
        x = x[perm]
    S = float(np.log2(pls))
    dev = "cuda"
    xd, ed, od, gd = x.to(dev), emb.to(dev), offs.to(dev), gid.to(dev)
    out = torch.empty(B, L * C, device=dev)
    dydx = torch.empty(L, B, 3 * C, device=dev)
    _be().fwd(xd, ed, od, out, B, 3, C, L, S, base, dydx, grids=(gd, T))
    out_v = torch.empty_like(out)
    _be().fwd(xd, ed, od, out_v, B, 3, C, L, S, base, None, grids=(gd, T))
    assert torch.equal(out_v, out)
    grad = torch.randn(L, B, C, generator=g)
    gpm = grad.permute(1, 0, 2).reshape(B, -1).contiguous().to(dev)
    ggx = torch.randn(B, 3, generator=g)
    gj = torch.randn(L, B, 3 * C, generator=g)                  # an arbitrary Jacobian cotangent
    ge = torch.zeros_like(ed)
    gx = torch.empty_like(xd)
    _be().bwd(gpm, xd, od, ge, B, 3, C, L, S, base, dydx, gx, grids=(gd, T))
    g2 = torch.zeros_like(ed)
    gg = torch.empty_like(gpm)
    if C >= 2:
        _be().bwd2(gpm, xd, od, B, 3, C, L, S, base, dydx, ggx.to(dev), gg, g2, grids=(gd, T))
    gjac = torch.zeros_like(ed)
    _be().bwd_jac(gpm, gj.to(dev), xd, od, gjac, B, 3, C, L, S, base, grids=(gd, T))
    for k in range(G):
        sel = (gid == k).nonzero().flatten()
        n = sel.numel()
        ref_out, ref_dydx = hash_oracle.fwd(x[sel], emb[k], offs, S, base, True)
        assert torch.equal(out.cpu()[sel], ref_out.permute(1, 0, 2).reshape(n, -1)), f"grid {k}: forward must be bit-exact"
        assert torch.equal(dydx.cpu()[:, sel], ref_dydx.view(n, L, 3 * C).permute(1, 0, 2)), f"grid {k}: dy_dx must be bit-exact"
        ref_gx, ref_ge = hash_oracle.bwd(grad[:, sel].contiguous(), x[sel], emb[k], offs, S, base, True, ref_dydx)
        assert torch.equal(gx.cpu()[sel], ref_gx)
        assert (ge.cpu()[k] - ref_ge).abs().max() <= 2e-6 * np.sqrt(n) * float(ref_ge.abs().max()), f"grid {k}: scatter"
        if C >= 2:
            ref_gg, ref_g2 = hash_oracle.bwd2(grad[:, sel].contiguous(), x[sel], emb[k], offs, S, base, ref_dydx, ggx[sel])
            assert torch.equal(gg.cpu()[sel], ref_gg.permute(1, 0, 2).reshape(n, -1))
            assert (g2.cpu()[k] - ref_g2).abs().max() <= 2e-6 * np.sqrt(n) * float(ref_g2.abs().max()), f"grid {k}: second backward"
        # value+Jacobian scatter == the single-grid launch on this grid's points (itself oracle-checked in test_model_gpu / stock fixtures)
        one = torch.zeros(T, C, device=dev)
        sd = sel.to(dev)
        _be().bwd_jac(gpm[sd].contiguous(), gj.to(dev)[:, sd].contiguous(), xd[sd].contiguous(), od, one, n, 3, C, L, S, base)
        assert (gjac[k] - one).abs().max() <= 2e-6 * np.sqrt(n) * float(one.abs().max()), f"grid {k}: value+Jacobian scatter"


def test_batched_grids_module_and_refusals():
    """BatchedHashEncoder == the HashEncoders it was stacked from, gradient included; the binned scatter refuses grid ids."""
    from holoscene_amd.hashencoder import HashEncoder
    from holoscene_amd.hashencoder.hashgrid import BatchedHashEncoder
    torch.manual_seed(3)
    encs = [HashEncoder(num_levels=8, base_resolution=8, log2_hashmap_size=12, desired_resolution=128).cuda() for _ in range(4)]
    for e in encs:
        e.embeddings.data.uniform_(-1, 1)
    bank = BatchedHashEncoder.from_encoders(encs)
    B = 4099
    x = torch.rand(B, 3, device="cuda") * 2 - 1
    gid = torch.randint(0, 4, (B,), device="cuda", dtype=torch.int32)
    y = bank(x, gid)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    for k, e in enumerate(encs):
        sel = gid == k
        yk = e(x[sel])
        assert torch.equal(yk, y[sel])
        (yk * w[sel]).sum().backward()
        assert (bank.embeddings.grad[k] - e.embeddings.grad).abs().max() <= 1e-5 * float(e.embeddings.grad.abs().max())
    ws = _be().scatter_workspace(B, 3, 2, 8, "cuda")
    if ws is not None:
        with pytest.raises(ValueError):
            _be().bwd_jac(w, None, x, bank.offsets, torch.zeros_like(bank.embeddings), B, 3, 2, 8, 1.0, 8, ws=ws, grids=(gid, 10))
    lay = _be()._layout(B, 3, 2, 8, grids=(gid, 0))           # a zero stride is refused by the library itself
    from holoscene_amd.hashencoder import backend
    import ctypes
    rc = backend.load_library().hs_hash_fwd(backend._dev(x, "x"), backend._dev(bank.embeddings.data, "e"), backend._dev(bank.offsets, "o", torch.int32),
                                            backend._dev(y.detach(), "y"), B, 3, 2, 8, ctypes.c_float(1.0), 8, None, ctypes.byref(lay), None)
    assert rc != 0
