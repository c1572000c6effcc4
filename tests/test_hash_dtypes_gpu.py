"""GPU: the hash encoder's double and half instantiations (csrc/hash_encode_dt.hip behind hs_hash_encode_*_dt: the reference dispatches its kernels over
double / float / half, hashencoder.cu:747, 778, 817) against the CPU checker oracle/hash_oracle_dt.c.  Forward, dy_dx, input backward and grad_grad:
bit for bit.  The two scatters: to the order of their atomics (double: 1e-12; half: a few half ulps per colliding contribution)."""
import numpy as np
import pytest
import torch

from oracle import hash_oracle as ho

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _grid(L, base, end, logmap, C, D, seed):
    pls = ho.per_level_scale_for(base, end, L)
    offs = torch.from_numpy(ho.level_offsets(L, base, pls, logmap, input_dim=D))
    g = torch.Generator().manual_seed(seed)
    emb = (torch.rand(int(offs[-1]), C, generator=g) * 2 - 1) * 0.5
    return emb, offs, float(np.log2(pls)), base


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16])
@pytest.mark.parametrize("D,C", [(3, 2), (3, 1), (2, 4), (3, 8)])
def test_other_scalar_types_vs_oracle(dtype, D, C):
    from holoscene_amd.hashencoder import backend
    be = backend._backend
    L = 6
    emb, offs, S, H = _grid(L, 4, 96, 9, C, D, seed=D * 10 + C)      # dense coarse levels and hashed fine ones
    g = torch.Generator().manual_seed(5)
    B = 777                                                          # ragged: not a multiple of the workgroup
    x = torch.rand(B, D, generator=g)
    x[3, 0], x[9, D - 1], x[11, 0] = 1.25, -0.5, 1.0                 # outside (zeros, no gradient) and on the upper edge
    x, emb = x.to(dtype), emb.to(dtype)
    grad = torch.randn(L, B, C, generator=g).to(dtype)
    ggx = torch.randn(B, D, generator=g).to(dtype)
    # ---- oracle
    o_ref, d_ref = ho.fwd_dt(x, emb, offs, S, H, True)
    ge_ref, gx_ref = ho.bwd_dt(grad, x, emb, offs, S, H, d_ref, True)
    gg_ref, g2_ref = ho.bwd2_dt(grad, x, emb, offs, S, H, d_ref, ggx)
    # ---- product, through the native entry points
    xd, ed, od, gd, ggd = x.to(DEV), emb.to(DEV), offs.to(DEV), grad.to(DEV), ggx.to(DEV)
    out = torch.full((L, B, C), 3.0, device=DEV, dtype=dtype)
    dydx = torch.full((B, L * D * C), 3.0, device=DEV, dtype=dtype)
    be.encode_forward_dt(xd, ed, od, out, B, D, C, L, S, H, dydx)
    bits = lambda t: t.cpu().contiguous().view(torch.int64 if dtype == torch.float64 else torch.int16)  # noqa: E731
    assert torch.equal(bits(out), bits(o_ref)), "forward must be bit-identical"
    assert torch.equal(bits(dydx), bits(d_ref)), "dy_dx must be bit-identical"
    out2 = torch.empty_like(out)
    be.encode_forward_dt(xd, ed, od, out2, B, D, C, L, S, H, None)       # value only
    assert torch.equal(bits(out2), bits(o_ref))
    ge, gx = torch.zeros_like(ed), torch.empty_like(xd)
    be.encode_backward_dt(gd, xd, ed, od, ge, B, D, C, L, S, H, dydx, gx)
    assert torch.equal(bits(gx), bits(gx_ref)), "input backward must be bit-identical"
    gg, g2 = torch.empty_like(gd), torch.zeros_like(ed)
    be.encode_second_backward_dt(gd, xd, ed, od, B, D, C, L, S, H, dydx, ggd, gg, g2)
    assert torch.equal(bits(gg), bits(gg_ref)), "grad_grad must be bit-identical"
    for name, got, ref in (("grad_embeddings", ge, ge_ref), ("grad2_embeddings", g2, g2_ref)):
        err = float((got.cpu().double() - ref.double()).abs().max())
        scale = float(ref.double().abs().max())
        tol = 1e-12 * scale if dtype == torch.float64 else 2e-2 * scale      # half: every colliding contribution rounds the running sum (2^-11 relative each)
        print(f"PARITY hash {dtype} D={D} C={C} {name}: max |diff| {err:.3e} of {scale:.3e}")
        assert err <= tol and scale > 0, (name, err, scale)
    assert not out[:, 3].any() and not out[:, 9].any() and not ge.isnan().any()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16])
def test_module_dispatches_on_dtype_and_double_passes_gradcheck(dtype):
    """HashEncoder's Function on a double / half call: the reference's three-level structure (forward, backward, second backward) on the *_dt
    kernels; in double the analytic table gradient passes torch's finite-difference gradcheck (what the reference's double instantiation is for)."""
    from holoscene_amd.hashencoder.hashgrid import hash_encode
    emb, offs, S, H = _grid(4, 4, 32, 8, 2, 3, seed=3)
    pls = float(2 ** S)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(40, 3, generator=g) * 0.98 + 0.01).to(DEV).to(dtype)
    e = emb.to(DEV).to(dtype).requires_grad_(True)
    y = hash_encode(x, e, offs.to(DEV), pls, H, False)
    assert y.dtype == dtype and y.shape == (40, 8)
    y.sum().backward()
    assert e.grad is not None and e.grad.dtype == dtype and float(e.grad.abs().sum()) > 0
    if dtype == torch.float64:
        fn = lambda t: hash_encode(x, t, offs.to(DEV), pls, H, False)  # noqa: E731
        assert torch.autograd.gradcheck(fn, (e.detach().clone().requires_grad_(True),), eps=1e-6, atol=1e-7, rtol=1e-5, nondet_tol=1e-12)
        # the double-backward structure: d/d emb of (d y / d x . v)
        xg = x.clone().requires_grad_(True)
        e2 = e.detach().clone().requires_grad_(True)
        yy = hash_encode(xg, e2, offs.to(DEV), pls, H, True)
        gx, = torch.autograd.grad(yy.sum(), xg, create_graph=True)
        (gx * torch.ones_like(gx)).sum().backward()
        assert e2.grad is not None and float(e2.grad.abs().sum()) > 0
