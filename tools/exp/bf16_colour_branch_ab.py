#!/usr/bin/env python3
"""Where does the bf16 colour branch lose 3-7 % of its parameter gradients?  

A/B on the CPU, plain torch, no product kernels: the colour branch of one reference iteration fixture -- colour-feature MLP 32 -> 256
(ReLU) -> 256, rendering network 337 -> 256 (ReLU) -> 256 (ReLU) -> 3 (sigmoid) -- is evaluated on the inputs and the upstream cotangent
d loss / d rgb that the fp32 whole-tensor path of the product produces on the fixture (hash features, points, view directions, normals
of every rendered sample), in these arithmetic schemes:

  fp32        everything in fp32                                                     (sanity: must reproduce the reference's gradients)
  fwd         GEMM operands of the FORWARD rounded to bf16 (fp32 accumulate); backward entirely fp32 on the saved (rounded) activations
  fwd+masks32 as `fwd`, but the three ReLU masks are taken from the fp32 forward     (-> what the mask flips alone cost)
  kernels     the scheme of csrc/appearance_mlp.hip + wgrad.hip: as `fwd`, plus cotangents rounded to bf16 between the layers, bf16 operands in
              the backward products, weight gradients as 128 bf16 per-slice partials summed in fp32
  kernels+c32 as `kernels` but fp32-stored cotangents between the layers
  kernels+p32 as `kernels` but fp32 per-slice partials

and every parameter gradient (and the cotangent of the colour hash features, whose scatter is the colour table's gradient) is compared
with the fp32 result in relative L2.

    python tools/exp/bf16_colour_branch_ab.py [fixture ...]        (default: stock_k32_bg stock_k21)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def capture(name):
    """Inputs of the colour branch and d loss / d rgb from the product's fp32 CPU path on the fixture (hash encoder = CPU oracle)."""
    import oracle_backend
    from helpers import load, rand_dict, section
    from model_helpers import build_loss, build_model
    from holoscene_amd.hashencoder import backend as be
    from holoscene_amd.model import loss as loss_mod, network, ray_sampler
    be._backend = oracle_backend.OracleBackend
    for mod in (network, ray_sampler):
        if hasattr(mod, "_be"):
            mod._be._backend = oracle_backend.OracleBackend
    ray_sampler.SAMPLER_IMPL, network.ops.COMPOSITE_IMPL, loss_mod.LOSS_IMPL = "torch", "torch", "torch"
    rec = load(name)
    model = build_model(rec).train()
    got = {}
    orig = model._rgb_at

    def spy(points_flat, dirs_flat, gradients, indices=None, x01=None):
        net = model.implicit_network
        feat = net.color_encoding(points_flat / net.divide_factor).detach()
        got.update(points=points_flat.detach(), dirs=dirs_flat.detach(), normals=gradients.detach(), feat=feat)
        rgb = orig(points_flat, dirs_flat, gradients, indices, x01)
        rgb.register_hook(lambda g: got.__setitem__("g_rgb", g.detach().clone()))
        return rgb
    model._rgb_at = spy
    out = model(section(rec, "in."), torch.tensor([0]), iter_step=int(rec["meta.iter_step"]), rng=rand_dict(rec))
    out["iter_step"] = int(rec["meta.iter_step"])
    lo = build_loss()(out, section(rec, "gt."), call_reg=bool(rec["meta.call_reg"]))
    lo["loss"].backward()
    from holoscene_amd.model.network import effective_weights
    net, rn = model.implicit_network, model.rendering_network
    mlp = net.color_grid_feature_map_mlp
    with torch.no_grad():
        R0, R1, R2 = effective_weights([rn.lin0, rn.lin1, rn.lin2])
    W = {"Wc0": mlp[0].weight.detach(), "bc0": mlp[0].bias.detach(), "Wc1": mlp[2].weight.detach(), "bc1": mlp[2].bias.detach(),
         "Wr0": R0.detach(), "br0": rn.lin0.bias.detach(), "Wr1": R1.detach(), "br1": rn.lin1.bias.detach(), "Wr2": R2.detach(),
         "br2": rn.lin2.bias.detach()}
    return got, W, rn


def posenc(x, nfreq=4):
    out = [x]
    for k in range(nfreq):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


def bf(t):
    return t.to(torch.bfloat16).float()


def run(inp, W, scheme, masks32=None, slices=128):
    """Manual forward + backward of the colour branch.  Returns ({name: gradient}, masks)."""
    fwd_bf = scheme != "fp32"
    bwd_bf = scheme.startswith("kernels")
    cot32 = scheme == "kernels+c32"
    part32 = scheme == "kernels+p32"
    r = bf if fwd_bf else (lambda t: t)
    rc = (lambda t: t) if (not bwd_bf or cot32) else bf            # cotangent storage between the layers
    rb = bf if bwd_bf else (lambda t: t)                           # operands of the backward products
    feat, pts, dirs, nrm, g_rgb = (inp[k].double().float() for k in ("feat", "points", "dirs", "normals", "g_rgb"))
    nrm = nrm.clone().requires_grad_(False)
    enc = torch.cat([posenc(pts), posenc(dirs), posenc(nrm)], -1)                   # 81 columns
    # ---- forward
    x0 = r(feat)
    a_c = x0 @ r(W["Wc0"]).t() + W["bc0"]
    m_c = a_c > 0
    hc = r(torch.relu(a_c))
    fv = r(hc @ r(W["Wc1"]).t() + W["bc1"])
    xin = torch.cat([r(enc), fv], -1)
    a0 = xin @ r(W["Wr0"]).t() + W["br0"]
    m0 = a0 > 0
    r0 = r(torch.relu(a0))
    a1 = r0 @ r(W["Wr1"]).t() + W["br1"]
    m1 = a1 > 0
    r1 = r(torch.relu(a1))
    y = r1 @ r(W["Wr2"]).t() + W["br2"]
    rgb = torch.sigmoid(y)
    masks = (m_c, m0, m1)
    if masks32 is not None:
        m_c, m0, m1 = masks32

    def wgrad(g, x):
        """g^T x with the kernels' split-M partials (bf16 or fp32 per slice) or plainly in fp32."""
        if not bwd_bf:
            return g.t() @ x
        M = x.shape[0]
        S = slices if M % slices == 0 else 1
        p = torch.bmm(rb(g).view(S, M // S, -1).transpose(1, 2), rb(x).view(S, M // S, -1))
        return (p if part32 else bf(p)).sum(0)
    # ---- backward
    gy = rc(g_rgb * rgb * (1 - rgb))
    G = {"Wr2": wgrad(gy, r1), "br2": gy.sum(0)}
    g1 = rc((rb(gy) @ rb(W["Wr2"])) * m1)
    G["Wr1"], G["br1"] = wgrad(g1, r0), g1.sum(0)
    g0 = rc((rb(g1) @ rb(W["Wr1"])) * m0)
    G["Wr0"], G["br0"] = wgrad(g0, xin), g0.sum(0)
    gx = rb(g0) @ rb(W["Wr0"])
    g_fv = rc(gx[:, 81:])
    G["Wc1"], G["bc1"] = wgrad(g_fv, hc), g_fv.sum(0)
    g_hc = rc((rb(g_fv) @ rb(W["Wc1"])) * m_c)
    G["Wc0"], G["bc0"] = wgrad(g_hc, x0), g_hc.sum(0)
    G["g_featc (-> colour table)"] = rb(g_hc) @ rb(W["Wc0"])
    return G, masks


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    names = sys.argv[1:] or ["stock_k32_bg", "stock_k21"]
    torch.manual_seed(0)
    for name in names:
        inp, W, _ = capture(name)
        ref, masks32 = run(inp, W, "fp32")
        B = inp["feat"].shape[0]
        print(f"== {name}: {B} rendered samples")
        res = {}
        for scheme, kw in (("fwd", {}), ("fwd+masks32", {"masks32": masks32}), ("kernels", {}), ("kernels+c32", {}), ("kernels+p32", {})):
            G, masks = run(inp, W, "fwd" if scheme == "fwd+masks32" else scheme, **kw)
            res[scheme] = {k: rel(G[k], ref[k]) for k in ref}
            if scheme == "fwd":
                flips = [float((a != b).float().mean()) for a, b in zip(masks, masks32)]
                print("   fraction of ReLU mask bits that differ from the fp32 forward (hc, r0, r1): " + ", ".join(f"{f:.2e}" for f in flips))
        keys = list(ref)
        print(f"   {'relative L2 vs fp32':28s}" + "".join(f"{s:>14s}" for s in res))
        for k in keys:
            print(f"   {k:28s}" + "".join(f"{res[s][k]:14.2e}" for s in res))


if __name__ == "__main__":
    main()
