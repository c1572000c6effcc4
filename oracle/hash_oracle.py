"""oracle/hash_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end for ``libhash_oracle.so`` (oracle/hash_oracle.c) plus a CPU
``torch.autograd.Function`` pair that reproduces the double-differentiable
structure of the reference encoder (hashencoder/hashgrid.py:14-101): the first
backward is itself a Function whose backward calls the second-backward kernels
and returns *no* gradient for the inputs (hashgrid.py:101).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C restatement (gcc, a second or two)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libhash_oracle.so")
        srcs = [os.path.join(_HERE, n) for n in ("hash_oracle.c", "hash_oracle_dt.c", "hash_oracle_dt_impl.h")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in srcs):
            build()
        _LIB = ctypes.CDLL(so)
        for name in ("hs_oracle_hash_fwd", "hs_oracle_hash_bwd", "hs_oracle_hash_bwd2", "hs_oracle_level_table"):
            getattr(_LIB, name).restype = ctypes.c_int
    return _LIB


def _p(t):
    assert t.is_contiguous() and t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


# ---- the reference's other scalar types (oracle/hash_oracle_dt.c): torch.float64 / torch.float16 tensors, reference layouts
# (outputs [L, B, C], dy_dx [B, L * D * C], grad [L, B, C])
def _sfx(t):
    return {torch.float64: "_f64", torch.float16: "_f16"}[t.dtype]


def fwd_dt(x, emb, offsets, S, H, calc_dydx):
    B, D = x.shape
    C, L = emb.shape[1], offsets.shape[0] - 1
    out = torch.empty(L, B, C, dtype=x.dtype)
    dydx = torch.empty(B, L * D * C, dtype=x.dtype) if calc_dydx else None
    rc = getattr(lib(), "hs_oracle_dt_fwd" + _sfx(x))(_p(x), _p(emb), _p(offsets), _p(out), B, D, C, L, ctypes.c_float(S), int(H), _p(dydx) if calc_dydx else None)
    assert rc == 0
    return out, dydx


def bwd_dt(grad, x, emb, offsets, S, H, dydx=None, want_gx=False):
    B, D = x.shape
    C, L = emb.shape[1], offsets.shape[0] - 1
    gemb = torch.zeros_like(emb)
    gx = torch.empty_like(x) if want_gx else None
    rc = getattr(lib(), "hs_oracle_dt_bwd" + _sfx(x))(_p(grad), _p(x), _p(offsets), _p(gemb), B, D, C, L, ctypes.c_float(S), int(H),
                                                      _p(dydx) if want_gx else None, _p(gx) if want_gx else None)
    assert rc == 0
    return gemb, gx


def bwd2_dt(grad, x, emb, offsets, S, H, dydx, ggx):
    B, D = x.shape
    C, L = emb.shape[1], offsets.shape[0] - 1
    gg, g2 = torch.empty_like(grad), torch.zeros_like(emb)
    rc = getattr(lib(), "hs_oracle_dt_bwd2" + _sfx(x))(_p(grad), _p(x), _p(offsets), B, D, C, L, ctypes.c_float(S), int(H), _p(dydx), _p(ggx), _p(gg), _p(g2))
    assert rc == 0
    return gg, g2


def level_offsets(num_levels, base_resolution, per_level_scale, log2_hashmap_size, input_dim=3):
    """Entry offsets per level (hashencoder/hashgrid.py:127-138)."""
    cap = 2 ** log2_hashmap_size
    offs, total = [], 0
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        offs.append(total)
        total += min(cap, res ** input_dim)
    offs.append(total)
    return np.asarray(offs, dtype=np.int32)


def per_level_scale_for(base_resolution, desired_resolution, num_levels):
    """hashencoder/hashgrid.py:112-113"""
    return np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))


def level_table(offsets, S, H):
    L = offsets.numel() - 1
    scale = torch.empty(L, dtype=torch.float32)
    res = torch.empty(L, dtype=torch.int32)
    tab = torch.empty(L, dtype=torch.int32)
    lib().hs_oracle_level_table(_p(offsets), ctypes.c_uint32(L), ctypes.c_float(S), ctypes.c_uint32(H), _p(scale), _p(res), _p(tab))
    return scale, res, tab


def fwd(x, emb, offsets, S, H, calc_dydx):
    B, D = x.shape
    C = emb.shape[1]
    L = offsets.numel() - 1
    out = torch.empty(L, B, C, dtype=torch.float32)
    dydx = torch.empty(B, L * D * C, dtype=torch.float32) if calc_dydx else torch.empty(1, dtype=torch.float32)
    rc = lib().hs_oracle_hash_fwd(_p(x), _p(emb), _p(offsets), _p(out), B, D, C, L, ctypes.c_float(S), H, int(calc_dydx), _p(dydx))
    assert rc == 0
    return out, dydx


def bwd(grad, x, emb, offsets, S, H, calc_dydx, dydx):
    B, D = x.shape
    C = emb.shape[1]
    L = offsets.numel() - 1
    grad_emb = torch.zeros_like(emb)
    grad_x = torch.zeros_like(x)
    rc = lib().hs_oracle_hash_bwd(_p(grad), _p(x), _p(emb), _p(offsets), _p(grad_emb), B, D, C, L, ctypes.c_float(S), H,
                                  int(calc_dydx), _p(dydx), _p(grad_x))
    assert rc == 0
    return grad_x, grad_emb


def bwd2(grad, x, emb, offsets, S, H, dydx, ggx):
    B, D = x.shape
    C = emb.shape[1]
    L = offsets.numel() - 1
    grad_grad = torch.zeros_like(grad)
    grad2_emb = torch.zeros_like(emb)
    rc = lib().hs_oracle_hash_bwd2(_p(grad), _p(x), _p(emb), _p(offsets), B, D, C, L, ctypes.c_float(S), H,
                                   _p(dydx), _p(ggx.contiguous()), _p(grad_grad), _p(grad2_emb))
    assert rc == 0
    return grad_grad, grad2_emb


class _Encode(torch.autograd.Function):
    """hashencoder/hashgrid.py:14-68"""

    @staticmethod
    def forward(ctx, x, emb, offsets, S, H, calc_dydx):
        x = x.contiguous()
        emb = emb.contiguous()
        out, dydx = fwd(x, emb, offsets, S, H, calc_dydx)
        B = x.shape[0]
        ctx.save_for_backward(x, emb, offsets, dydx)
        ctx.cfg = (S, H, calc_dydx)
        return out.permute(1, 0, 2).reshape(B, -1)

    @staticmethod
    def backward(ctx, g):
        x, emb, offsets, dydx = ctx.saved_tensors
        S, H, calc = ctx.cfg
        B = x.shape[0]
        L = offsets.numel() - 1
        g = g.view(B, L, -1).permute(1, 0, 2).contiguous()
        gx, gemb = _EncodeBwd.apply(g, x, emb, offsets, S, H, calc, dydx)
        return (gx if calc else None), gemb, None, None, None, None


class _EncodeBwd(torch.autograd.Function):
    """hashencoder/hashgrid.py:71-101"""

    @staticmethod
    def forward(ctx, g, x, emb, offsets, S, H, calc, dydx):
        ctx.save_for_backward(g, x, emb, offsets, dydx)
        ctx.cfg = (S, H)
        return bwd(g, x, emb, offsets, S, H, calc, dydx)

    @staticmethod
    def backward(ctx, ggx, _gg_emb):
        g, x, emb, offsets, dydx = ctx.saved_tensors
        S, H = ctx.cfg
        grad_grad, grad2_emb = bwd2(g, x, emb, offsets, S, H, dydx, ggx)
        return grad_grad, None, grad2_emb, None, None, None, None, None


def hash_encode(x01, emb, offsets, per_level_scale, base_resolution, calc_dydx):
    """x01 in [0,1]^D -> [B, L*C]; S = log2(per_level_scale) (hashgrid.py:31)."""
    S = float(np.log2(per_level_scale))
    return _Encode.apply(x01, emb, offsets, S, int(base_resolution), bool(calc_dydx))


class RefBackendShim:
    """Drop-in for the reference's ``_backend`` pybind module (bindings.cpp:5-9),
    used ONLY by tests/golden/make_golden.py to run the reference's Python layer
    on CPU on top of this oracle."""

    @staticmethod
    def hash_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc, dy_dx):
        rc = lib().hs_oracle_hash_fwd(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), B, D, C, L,
                                      ctypes.c_float(S), H, int(calc), _p(dy_dx))
        assert rc == 0

    @staticmethod
    def hash_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, calc, dy_dx, grad_inputs):
        rc = lib().hs_oracle_hash_bwd(_p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_embeddings), B, D, C, L,
                                      ctypes.c_float(S), H, int(calc), _p(dy_dx), _p(grad_inputs))
        assert rc == 0

    @staticmethod
    def hash_encode_second_backward(grad, inputs, embeddings, offsets, B, D, C, L, S, H, calc, dy_dx, ggx, grad_grad, grad2_emb):
        rc = lib().hs_oracle_hash_bwd2(_p(grad), _p(inputs), _p(embeddings), _p(offsets), B, D, C, L,
                                       ctypes.c_float(S), H, _p(dy_dx), _p(ggx.contiguous()), _p(grad_grad), _p(grad2_emb))
        assert rc == 0
