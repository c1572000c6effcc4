"""CPU: the checker of the hash encoder's double / half instantiations (oracle/hash_oracle_dt.c) against what can pin it here -- its half
conversions against torch's, its double results against the float oracle's (oracle/hash_oracle.c) on float inputs."""
import ctypes

import numpy as np
import torch

from oracle import hash_oracle as ho


def _grid(L=5, base=4, end=64, logmap=9, C=2, D=3, seed=0):
    pls = ho.per_level_scale_for(base, end, L)
    offs = torch.from_numpy(ho.level_offsets(L, base, pls, logmap, input_dim=D))
    g = torch.Generator().manual_seed(seed)
    emb = (torch.rand(int(offs[-1]), C, generator=g) * 2 - 1) * 0.5
    return emb, offs, float(np.log2(pls)), base


def test_half_conversions_equal_torch():
    lib = ho.lib()
    lib.hs_oracle_f2h.restype = ctypes.c_uint16
    lib.hs_oracle_h2f.restype = ctypes.c_float
    g = torch.Generator().manual_seed(1)
    vals = torch.cat([torch.randn(4000, generator=g) * s for s in (1e-8, 1e-6, 1e-4, 1e-2, 1.0, 100.0, 7e4)]
                     + [torch.tensor([0.0, -0.0, 65504.0, 65519.9, 65520.0, 6.1e-5, 5.96e-8, 2.98e-8, 2.99e-8, float("inf"), -float("inf")])])
    want = vals.half()
    got = torch.tensor([lib.hs_oracle_f2h(ctypes.c_float(float(v))) for v in vals], dtype=torch.int32).to(torch.int16).view(torch.float16)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    allh = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16)
    back = torch.tensor([lib.hs_oracle_h2f(ctypes.c_uint16(int(b) & 0xffff)) for b in torch.arange(0, 65536)], dtype=torch.float32)
    ok = ~torch.isnan(allh.float())
    assert torch.equal(back[ok], allh.float()[ok])


def test_double_oracle_agrees_with_the_float_oracle_and_half_with_both():
    emb, offs, S, H = _grid()
    g = torch.Generator().manual_seed(2)
    B = 300
    x = torch.rand(B, 3, generator=g)
    x[5] = torch.tensor([0.2, 1.5, 0.3])                     # outside: zeros, no gradient
    out32, dy32 = ho.fwd(x, emb, offs, S, H, True)           # float oracle: the same layouts
    o64, d64 = ho.fwd_dt(x.double(), emb.double(), offs, S, H, True)
    L, C = offs.shape[0] - 1, emb.shape[1]
    assert float((o64 - out32.double()).abs().max()) < 1e-6 and float((d64 - dy32.double()).abs().max()) < 1e-4 * float(dy32.abs().max())
    assert not o64[:, 5].any() and not d64[5].any()
    o16, d16 = ho.fwd_dt(x.half(), emb.half(), offs, S, H, True)
    o64h, _ = ho.fwd_dt(x.half().double(), emb.half().double(), offs, S, H, False)
    assert float((o16.double() - o64h).abs().max()) < 4e-3       # eight roundings to half on values below 0.5
    # backward: the transpose of the forward (double: to rounding)
    gr = torch.randn(L, B, C, generator=g, dtype=torch.float64)
    ge, gx = ho.bwd_dt(gr, x.double(), emb.double(), offs, S, H, d64, True)
    lhs = float((gr * o64).sum())
    rhs = float((ge * emb.double()).sum())
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))
    # second backward, grad_grad: the derivative of (grad . dy_dx) along ggx
    ggx = torch.randn(B, 3, generator=g, dtype=torch.float64)
    gg, g2 = ho.bwd2_dt(gr, x.double(), emb.double(), offs, S, H, d64, ggx)
    want = torch.einsum("bd,bldc->lbc", ggx, d64.view(B, L, 3, C))
    assert float((gg - want).abs().max()) < 1e-12
    # ... and its table part: d/d emb of sum(gx . ggx) -- linear in emb, so g2 . emb == gx . ggx
    assert abs(float((g2 * emb.double()).sum()) - float((gx * ggx).sum())) < 1e-9 * max(1.0, abs(float((gx * ggx).sum())))
