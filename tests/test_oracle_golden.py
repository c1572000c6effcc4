"""CPU: the oracle restatement (oracle/) against fixtures generated from the reference import."""
import numpy as np
import pytest
import torch

from helpers import load, oracle_cfg, rand_dict, section
from oracle import hash_oracle
from oracle.stage1_oracle import Stage1Oracle


def close(a, b, rtol, atol, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max err {err.max():.3e} (worst tol {float(tol.flatten()[err.argmax()]) if err.numel() else 0:.3e})"


@pytest.mark.parametrize("tag,cfg", [("stock", (16, 16, 2048, 19)), ("c1", (8, 16, 256, 15)), ("tiny", (4, 4, 32, 10))])
def test_offset_tables(tag, cfg):
    rec = load("hash_tables")
    L, base, end, logmap = cfg
    pls = hash_oracle.per_level_scale_for(base, end, L)
    assert np.float64(pls) == rec[f"{tag}.per_level_scale"]
    np.testing.assert_array_equal(hash_oracle.level_offsets(L, base, pls, logmap), rec[f"{tag}.offsets"])


def test_stock_offsets_match_survey():
    pls = hash_oracle.per_level_scale_for(16, 2048, 16)
    offs = hash_oracle.level_offsets(16, 16, pls, 19)
    assert list(offs[:7]) == [0, 4096, 16263, 46054, 125561, 330940, 855228] and offs[-1] == 6098108
    scale, res, tab = hash_oracle.level_table(torch.from_numpy(offs), float(np.log2(pls)), 16)
    assert res.tolist() == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]


@pytest.mark.parametrize("name", ["hash_small", "hash_mid"])
def test_hash_regression(name):
    r = load(name)
    x, emb, offs = (torch.from_numpy(r[k]) for k in ("x", "emb", "offsets"))
    S, H = float(r["S"]), int(r["base"])
    out, dydx = hash_oracle.fwd(x, emb, offs, S, H, True)
    assert torch.equal(out, torch.from_numpy(r["out"])) and torch.equal(dydx, torch.from_numpy(r["dydx"]))
    gx, gemb = hash_oracle.bwd(torch.from_numpy(r["grad"]), x, emb, offs, S, H, True, dydx)
    assert torch.equal(gx, torch.from_numpy(r["grad_x"])) and torch.equal(gemb, torch.from_numpy(r["grad_emb"]))
    gg, g2 = hash_oracle.bwd2(torch.from_numpy(r["grad"]), x, emb, offs, S, H, dydx, torch.from_numpy(r["ggx"]))
    assert torch.equal(gg, torch.from_numpy(r["grad_grad"])) and torch.equal(g2, torch.from_numpy(r["grad2_emb"]))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_hash_regression_stock_grid(seed):
    """G1 of SURVEY 8c at the stock grid (16 levels, T = 2^19, 16 -> 2048, B = 1000 incl. boundary and out-of-cube points)."""
    from helpers import load_hash_stock
    r = load_hash_stock(f"hash_stock_s{seed}")
    x, emb, offs = r["x"], r["emb"], r["offsets"]
    S, H = float(r["S"]), int(r["base"])
    out, dydx = hash_oracle.fwd(x, emb, offs, S, H, True)
    assert torch.equal(out, r["out"]) and torch.equal(dydx, r["dydx"])
    gx, gemb = hash_oracle.bwd(r["grad"], x, emb, offs, S, H, True, dydx)
    assert torch.equal(gx, r["grad_x"]) and torch.equal(gemb, r["grad_emb"])
    gg, g2 = hash_oracle.bwd2(r["grad"], x, emb, offs, S, H, dydx, r["ggx"])
    assert torch.equal(gg, r["grad_grad"]) and torch.equal(g2, r["grad2_emb"])


def test_hash_self_consistency():
    """dy_dx = d(out)/dx by central differences; bwd = transpose of fwd; bwd2 = d(grad_x . ggx)/d(grad, emb)."""
    r = load("hash_small")
    x, emb, offs = (torch.from_numpy(r[k]) for k in ("x", "emb", "offsets"))
    S, H = float(r["S"]), int(r["base"])
    L = offs.numel() - 1
    B = x.shape[0]
    inside = ((x > 0.02) & (x < 0.98)).all(-1)
    out, dydx = hash_oracle.fwd(x, emb, offs, S, H, True)
    dydx = dydx.view(B, L, 3, 2)
    h = 1e-4
    scale, _, _ = hash_oracle.level_table(offs, S, H)
    for d in range(3):
        e = torch.zeros(3)
        e[d] = h
        op, _ = hash_oracle.fwd((x + e).contiguous(), emb, offs, S, H, False)
        om, _ = hash_oracle.fwd((x - e).contiguous(), emb, offs, S, H, False)
        fd = ((op - om) / (2 * h)).permute(1, 0, 2)  # [B,L,C]
        # away from cell borders (smoothstep is C1 there too, but keep clear of floor() jumps)
        frac = (x[:, None, d] * scale[None]) % 1.0
        ok = inside[:, None] & (frac > 0.05) & (frac < 0.95)
        err = (fd - dydx[:, :, d]).abs()[ok]
        assert err.max() < 2e-2 * max(1.0, float(dydx.abs().max())), err.max()
    # transpose test: <fwd(emb), g> == <emb, bwd(g)>
    g = torch.from_numpy(r["grad"])
    _, gemb = hash_oracle.bwd(g, x, emb, offs, S, H, False, torch.empty(1))
    lhs = (out.double() * g.double()).sum()
    rhs = (emb.double() * gemb.double()).sum()
    assert abs(lhs - rhs) < 1e-3 * abs(lhs)
    # second backward: grad_x . ggx is bilinear in (grad, emb) -> directional derivatives are exact
    ggx = torch.from_numpy(r["ggx"])
    gg, g2 = hash_oracle.bwd2(g, x, emb, offs, S, H, dydx.reshape(B, -1).contiguous(), ggx)
    gx, _ = hash_oracle.bwd(g, x, emb, offs, S, H, True, dydx.reshape(B, -1).contiguous())
    val = (gx.double() * ggx.double()).sum()
    assert abs((gg.double() * g.double()).sum() - val) < 1e-3 * abs(val)        # linear in grad
    assert abs((g2.double() * emb.double()).sum() - val) < 1e-3 * abs(val)      # linear in emb


@pytest.mark.parametrize("name", [f"sampler_{i}" for i in range(5)] + ["sampler_eval", "stock_sampler_0"])
def test_sampler_matches_reference(name):
    rec = load(name)
    o = Stage1Oracle(oracle_cfg(rec), section(rec, "state."))
    o.training = bool(rec["meta.train"])
    ins = section(rec, "in.")
    z, z_eik, rounds = o.sample_z(ins["ray_dirs"], ins["cam_loc"], rand_dict(rec))
    assert rounds == int(rec["meta.rounds"])
    close(z, rec["out.z_vals"], 1e-5, 1e-5, "z_vals")
    close(z_eik, rec["out.z_samples_eik"], 1e-5, 1e-5, "z_eik")


@pytest.mark.parametrize("name", ["net_k2", "net_k21", "net_k32", "stock_net_k21", "stock_net_k32"])
def test_network_methods_match_reference(name):
    """G3 / G5 of SURVEY 8c: the oracle's network, rendering and compositing restatements against direct calls of the reference's
    ObjectImplicitNetworkGrid / RenderingNetwork / volume_rendering / occlusion_opacity (K = 2, 21, 32)."""
    rec = load(name)
    o = Stage1Oracle(oracle_cfg(rec), section(rec, "state."))
    o.training = False
    ins = section(rec, "in.")
    x, dirs = ins["x"], ins["dirs"]
    b = int(rec["meta.b"])
    close(o.implicit(x.clone()), rec["forward.ret0"], 1e-5, 1e-6, "forward")
    outs = o.get_outputs(x.clone())
    for i, v in enumerate(outs):
        close(v, rec[f"get_outputs.ret{i}"], 1e-4, 1e-6, f"get_outputs[{i}]")
    close(o.gradient(x.clone()), rec["gradient.ret0"], 1e-4, 1e-6, "gradient")
    close(o.sdf_vals(x.clone()), rec["get_sdf_vals.ret0"], 1e-5, 1e-6, "get_sdf_vals")
    close(o.sdf_raw(x.clone()), rec["get_sdf_raw.ret0"], 1e-5, 1e-6, "get_sdf_raw")
    close(o.object_sdf_vals(x.clone(), b), rec["get_object_sdf_vals.ret0"], 1e-5, 1e-6, "get_object_sdf_vals")
    ref = section(rec, "get_outputs.")
    close(o.rendering(x, ref["ret2"], dirs, ref["ret1"]), rec["rendering.ret0"], 1e-5, 1e-6, "rendering")
    vin = section(rec, "vr.in.")
    w, T, dists = o.volume_rendering(vin["z"], vin["sdf"])
    close(w, rec["vr.out.weights"], 1e-5, 1e-7, "weights")
    close(T, rec["vr.out.transmittance"], 1e-5, 1e-7, "transmittance")
    close(o.occlusion_opacity(T, dists, vin["raw"]), rec["vr.out.occlusion"], 1e-5, 1e-7, "occlusion")


@pytest.mark.parametrize("name", ["iter_k3_bg", "iter_k5", "stock_k32_bg", "stock_k21", "stock_k40", "stock_k64_bg", "stock_l8_k3_bg"])
def test_iteration_matches_reference(name):
    rec = load(name)
    sd = section(rec, "state.")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
    full = dict(sd)
    full.update(params)
    o = Stage1Oracle(oracle_cfg(rec), full)
    ins, gt = section(rec, "in."), section(rec, "gt.")
    out = o.forward(ins["uv"], ins["pose"], ins["intrinsics"], rand_dict(rec), iter_step=int(rec["meta.iter_step"]))
    ref_out = section(rec, "out.")
    assert ("bg_depth_values" in out) == bool(rec["meta.has_bg"])
    for k, v in ref_out.items():
        if k == "bg_mask":
            assert torch.equal(out[k], v)
            continue
        close(out[k], v, 2e-4, 2e-5, k)
    lo = o.loss(out, gt, call_reg=bool(rec["meta.call_reg"]))
    for k, v in section(rec, "loss.").items():
        close(lo[k], v, 1e-4, 1e-6, "loss." + k)
    lo["loss"].backward()
    for k, v in section(rec, "grad.").items():
        g = params[k].grad
        assert g is not None, k
        close(g, v, 2e-3, 1e-5 * max(1.0, float(v.abs().max())), "grad." + k)
    # one Adam step with the reference's groups (holoscene_train.py:156-164)
    lr = 5e-4
    groups = [
        {"params": [params[k] for k in params if k.endswith("embeddings")], "lr": lr * 20},
        {"params": [params[k] for k in params if not k.endswith("embeddings") and k != "density.beta"], "lr": lr},
        {"params": [params["density.beta"]], "lr": lr}]
    torch.optim.Adam(groups, betas=(0.9, 0.99), eps=1e-15).step()
    for k, v in section(rec, "adam1.").items():
        close(params[k].detach(), v, 1e-5, 1e-6, "adam1." + k)


@pytest.mark.parametrize("name", ["sampler_1", "sampler_4", "stock_sampler_0"])
def test_oracle_sampler_round_intermediates(name):
    """The oracle's d* (Theorem 1), error bound (ray_sampler.py:450-458) and line search against the reference's per-round
    intermediates (fixture keys round{i}.*: merged set, SDF, d*, bound at beta0, beta after the search)."""
    import torch
    from helpers import load, oracle_cfg, section
    from oracle.stage1_oracle import Stage1Oracle
    rec = load(name)
    orc = Stage1Oracle(oracle_cfg(rec), section(rec, "state."))
    c = orc.cfg
    beta0 = orc.beta().detach()
    beta = None
    for r in range(int(rec["meta.rounds"])):
        z, sdf, d_ref, e_ref, b_ref = (torch.from_numpy(rec[f"round{r}.{k}"]) for k in ("z", "sdf", "d_star", "err0", "beta"))
        dists = z[:, 1:] - z[:, :-1]
        if r == 0:
            beta = torch.sqrt((1.0 / (4.0 * torch.log(torch.tensor(c.eps + 1.0)))) * (dists ** 2).sum(-1))
        a, b, cc = dists, sdf[:, :-1].abs(), sdf[:, 1:].abs()
        first, second = a ** 2 + b ** 2 <= cc ** 2, a ** 2 + cc ** 2 <= b ** 2
        s_ = (a + b + cc) / 2
        heron = 2.0 * torch.sqrt(s_ * (s_ - a) * (s_ - b) * (s_ - cc)) / a
        d_star = torch.where(first, b, torch.zeros_like(a))
        d_star = torch.where(second, cc, d_star)
        d_star = torch.where(~first & ~second & (b + cc - a > 0), heron, d_star)
        d_star = (sdf[:, 1:].sign() * sdf[:, :-1].sign() == 1) * d_star
        assert torch.allclose(d_star, d_ref, rtol=1e-6, atol=1e-9), r
        err = orc.error_bound(beta0, sdf.reshape(-1, 1), z, dists, d_ref)
        assert torch.allclose(err, e_ref, rtol=1e-5, atol=1e-7), r
        beta = torch.where(err <= c.eps, beta0.expand_as(beta), beta)
        lo, hi = beta0.expand_as(beta).clone(), beta.clone()
        for _ in range(c.beta_iters):
            mid = (lo + hi) / 2
            e = orc.error_bound(mid[:, None], sdf.reshape(-1, 1), z, dists, d_ref)
            hi = torch.where(e <= c.eps, mid, hi)
            lo = torch.where(e > c.eps, mid, lo)
        assert torch.allclose(hi, b_ref, rtol=1e-6, atol=0), r
        beta = b_ref.clone()
