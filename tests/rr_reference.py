"""TEST-ONLY torch restatement of the reverse-over-reverse trunk (csrc/trunk_rr.hip): every intermediate the four kernels exchange, in
plain [n, 256] layout, optionally with the kernels' bf16 rounding points (operands of every product and every stored activation bf16,
accumulation and epilogue arithmetic fp32).  The closed form itself is checked against autograd's double backward in
tools/exp/rr_trunk_math.py."""
import torch

NPE = 39


def bfr(t, on):
    return t.to(torch.bfloat16).float() if on else t


def posenc(x):
    out = [x]
    for k in range(6):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


def sp100(a):
    return torch.nn.functional.softplus(a, beta=100)


def efac(x, dydx, jac):
    """E [n, 71, 3] = d xt / d x (positional encoding + jac * dy_dx; dydx [16, n, 6] = [level][sample][d * 2 + c])."""
    n = x.shape[0]
    E = torch.zeros(n, 71, 3, device=x.device, dtype=x.dtype)
    for d in range(3):
        E[:, d, d] = 1
        for k in range(6):
            E[:, 3 + 6 * k + d, d] = 2.0 ** k * torch.cos(x[:, d] * 2.0 ** k)
            E[:, 3 + 6 * k + 3 + d, d] = -(2.0 ** k) * torch.sin(x[:, d] * 2.0 ** k)
    dy = dydx.view(16, n, 3, 2).permute(1, 0, 3, 2).reshape(n, 32, 3)       # [n, level * 2 + c, d]
    E[:, NPE:, :] = jac * dy
    return E


def forward(x, feat, dydx, W, jac, bf16=True):
    W0, b0, W1, b1, W2, b2 = W
    r = lambda t: bfr(t, bf16)  # noqa: E731
    xt = torch.cat([posenc(x), feat], -1)
    a0 = r(xt) @ r(W0).t() + b0
    h0 = r(sp100(a0))
    a1 = h0 @ r(W1).t() + b1
    h1 = r(sp100(a1))
    W2v = r(W2) + r(W2 - r(W2)) if bf16 else W2       # the value product carries W2 as two bf16 planes (trunk_pack.h: the low plane)
    y = h1 @ W2v.t() + b2
    sdf, idx = y.min(-1)
    s1, s0 = 1 - torch.exp(-100 * h1), 1 - torch.exp(-100 * h0)
    v1 = r(W2[idx] * s1)
    u0f = v1 @ r(W1)
    u0 = r(u0f)
    v0 = r(u0f * s0)
    ux = v0 @ r(W0)
    E = efac(x, dydx, jac)
    grad = torch.einsum("bj,bjd->bd", ux, E)
    return dict(xt=r(xt), h0=h0, h1=h1, y=y, sdf=sdf, idx=idx, s0=s0, s1=s1, v1=v1, u0=u0, u0f=u0f, v0=v0, ux=ux, grad=grad, E=E, uxh=ux[:, NPE:])


def backward(f, W, g_y, g_grad, jac, bf16=True):
    """g_y [n, K] (the minimum's cotangent already folded in), g_grad [n, 3] or None."""
    W0, b0, W1, b1, W2, b2 = W
    r = lambda t: bfr(t, bf16)  # noqa: E731
    s0, s1, h0, h1 = f["s0"], f["s1"], f["h0"], f["h1"]
    out = {}
    n = h0.shape[0]
    if g_grad is not None:
        uxb = r(torch.einsum("bjd,bd->bj", f["E"], g_grad))
        v0b = uxb @ r(W0).t()
        u0b = r(v0b * s0)
        a0p = r(v0b * f["u0"] * (100 * s0 * (1 - s0)))
        v1b = u0b @ r(W1).t()
        u1b = r(v1b * s1)
        a1p = r(v1b * W2[f["idx"]] * (100 * s1 * (1 - s1)))
        g_dy = jac * f["uxh"][:, :, None] * g_grad[:, None, :]                  # [n, 32, 3]
        out.update(uxb=uxb, u0b=u0b, a0p=a0p, a1p=a1p, u1b=u1b,
                   g_dydx=g_dy.view(n, 16, 2, 3).permute(1, 0, 3, 2).reshape(16, n, 6))
    else:
        a0p = a1p = 0.0
    gy = r(g_y)
    h1b = gy @ r(W2)
    a1 = r(a1p + h1b * s1)
    h0b = a1 @ r(W1)
    a0 = r(a0p + h0b * s0)
    xtb = a0 @ r(W0)
    out.update(gy=gy, a1=a1, a0=a0, g_feat=xtb[:, NPE:].reshape(n, 16, 2).permute(1, 0, 2).contiguous())
    # weight gradients from the (rounded) operands
    dW1 = a1.t() @ h0
    dW0 = a0.t() @ f["xt"]
    dW2 = gy.t() @ h1
    if g_grad is not None:
        dW1 = dW1 + f["v1"].t() @ out["u0b"]
        dW0 = dW0 + f["v0"].t() @ out["uxb"]
        dW2 = dW2.index_add(0, f["idx"], out["u1b"])
    out.update(dW0=dW0, dW1=dW1, dW2=dW2, db0=a0.sum(0), db1=a1.sum(0), db2=gy.sum(0))
    return out


def tp_decode(T, n):
    """tile-packed [tiles, 16, 64, 8] bf16 -> [n, 256] fp32 (neuron nu(s, h, e) = 16 s + 8 (e >> 2) + 4 h + (e & 3), lane = 32 h + row)."""
    tiles = T.numel() // (16 * 64 * 8)
    t = T.view(tiles, 16, 2, 32, 2, 4).float()              # [tile, s, h, row, e >> 2, e & 3]
    out = t.permute(0, 3, 1, 4, 2, 5).reshape(tiles * 32, 256)   # [tile, row, s, e >> 2, h, e & 3] -> neuron 16 s + 8 (e>>2) + 4 h + (e&3)
    return out[:n]
