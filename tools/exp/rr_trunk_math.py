#!/usr/bin/env python3
"""Reverse-over-reverse formulation of the trunk for the rendered samples ("V1b"): closed-form forward (values, min,
d min / dx) and closed-form backward (all parameter gradients + input-feature cotangents), checked against autograd's double backward in
float64.  Plain torch, no kernels: this is the arithmetic csrc/trunk_rr.hip implements."""
import torch

torch.manual_seed(0)
dt = torch.float64
n, K, F = 37, 5, 71
NPE, L, C = 39, 16, 2


def sp(a):
    return torch.nn.functional.softplus(a, beta=100)


def make():
    W0, W1, W2 = (torch.randn(256, F, dtype=dt) * 0.15).requires_grad_(), (torch.randn(256, 256, dtype=dt) * 0.08).requires_grad_(), \
        (torch.randn(K, 256, dtype=dt) * 0.1).requires_grad_()
    b0, b1, b2 = (torch.randn(256, dtype=dt) * 0.05).requires_grad_(), (torch.randn(256, dtype=dt) * 0.05).requires_grad_(), \
        (torch.randn(K, dtype=dt) * 0.1).requires_grad_()
    return W0, b0, W1, b1, W2, b2


def posenc(x):
    out = [x]
    for k in range(6):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


# a differentiable stand-in for the hash encoding: feat = G(x) with known Jacobian dydx (both "parameters" here, as the table is)
x = torch.randn(n, 3, dtype=dt) * 0.5
A = (torch.randn(3, L * C, dtype=dt) * 0.3).requires_grad_()      # feat = sin(x A): smooth, so that autograd can provide dfeat/dx
params = make()
W0, b0, W1, b1, W2, b2 = params
jac_scale = 0.5


def reference(x):
    """autograd formulation (what the reference does: network.py:273-301): values, min, d min / dx with create_graph."""
    x = x.clone().requires_grad_()
    feat = torch.sin(x @ A)
    xt = torch.cat([posenc(x), feat], -1)
    h0 = sp(xt @ W0.t() + b0)
    h1 = sp(h0 @ W1.t() + b1)
    y = h1 @ W2.t() + b2
    m, idx = y.min(-1)
    g = torch.autograd.grad(m.sum(), x, create_graph=True)[0]
    return y, m, idx, g, feat, xt


y, m, idx, g, feat, xt = reference(x)
c_y, c_m, c_g = torch.randn_like(y), torch.randn_like(m), torch.randn_like(g)
loss = (y * c_y).sum() + (m * c_m).sum() + (g * c_g).sum()
ref = torch.autograd.grad(loss, list(params) + [A])

# ---------------------------------------------------------------- closed form
with torch.no_grad():
    xa = x @ A
    feat, dfeat = torch.sin(xa), torch.cos(xa)[:, None, :] * A[None]          # dfeat[b, d, j] = d feat_j / d x_d
    xt = torch.cat([posenc(x), feat], -1)
    a0 = xt @ W0.t() + b0
    h0, s0 = sp(a0), torch.sigmoid(100 * a0)
    a1 = h0 @ W1.t() + b1
    h1, s1 = sp(a1), torch.sigmoid(100 * a1)
    y2 = h1 @ W2.t() + b2
    m2, idx2 = y2.min(-1)
    u1 = W2[idx2]
    v1 = u1 * s1
    u0 = v1 @ W1
    v0 = u0 * s0
    ux = v0 @ W0                                                               # [n, 71] = d min / d xt
    # E[b, j, d] = d xt_j / d x_d
    E = torch.zeros(n, F, 3, dtype=dt)
    for d in range(3):
        E[:, d, d] = 1
        for k in range(6):
            E[:, 3 + 6 * k + d, d] = 2.0 ** k * torch.cos(x[:, d] * 2.0 ** k)
            E[:, 3 + 6 * k + 3 + d, d] = -(2.0 ** k) * torch.sin(x[:, d] * 2.0 ** k)
    E[:, NPE:, :] = dfeat.transpose(1, 2)
    g2 = torch.einsum("bj,bjd->bd", ux, E)
    print(float((y2-y).abs().max()), torch.equal(idx2,idx), float((g2-g).abs().max()), float(g.abs().max()))
    assert torch.allclose(y2, y) and torch.equal(idx2, idx) and torch.allclose(g2, g, rtol=1e-7, atol=1e-7), "forward"   # (torch.softplus is linear above 100 a = 20: its derivative differs from the sigmoid by e^-20)
    # ---- backward
    gy = c_y.clone()
    gy[torch.arange(n), idx2] += c_m
    ux_bar = torch.einsum("bjd,bd->bj", E, c_g)                               # cotangent of ux
    g_dfeat = ux[:, NPE:, None] * c_g[:, None, :]                             # cotangent of E's feature block: [n, 32, 3]  (-> hash d/dx scatter)
    v0_bar = ux_bar @ W0.t()
    dW0 = v0.t() @ ux_bar
    u0_bar = v0_bar * s0
    a0_bar = v0_bar * u0 * (100 * s0 * (1 - s0))
    v1_bar = u0_bar @ W1.t()
    dW1 = v1.t() @ u0_bar
    u1_bar = v1_bar * s1
    a1_bar = v1_bar * u1 * (100 * s1 * (1 - s1))
    dW2 = torch.zeros_like(W2)
    dW2.index_add_(0, idx2, u1_bar)
    h1_bar = gy @ W2
    dW2 += gy.t() @ h1
    db2 = gy.sum(0)
    a1_bar = a1_bar + h1_bar * s1
    db1 = a1_bar.sum(0)
    h0_bar = a1_bar @ W1
    dW1 += a1_bar.t() @ h0
    a0_bar = a0_bar + h0_bar * s0
    db0 = a0_bar.sum(0)
    xt_bar = a0_bar @ W0
    dW0 += a0_bar.t() @ xt
    g_feat = xt_bar[:, NPE:]
    # stand-in encoding: feat = sin(xA), dfeat = cos(xA) A  ->  dA
    dA = x.t() @ (g_feat * torch.cos(xa))
    dA += torch.einsum("bjd,bj,dj->dj", g_dfeat, torch.cos(xa), torch.ones(3, L * C, dtype=dt))           # through the explicit A factor
    dA += x.t() @ (-(torch.sin(xa)) * torch.einsum("bjd,dj->bj", g_dfeat, A))                              # through cos(xA)
got = [dW0, db0, dW1, db1, dW2, db2, dA]
names = ["W0", "b0", "W1", "b1", "W2", "b2", "A (encoding stand-in)"]
for nme, a, b in zip(names, got, ref):
    err = float((a - b).abs().max() / b.abs().max())
    print(f"{nme:24s} max rel err {err:.2e}")
    assert err < 1e-6, nme
print("reverse-over-reverse closed form == autograd double backward")
