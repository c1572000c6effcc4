#!/usr/bin/env python3
"""tools/microbench_hash_grids.py -- the points of G per-object hash grids (stock geometry, 16 levels, T = 2^19) in ONE launch
(hsHashLayout::grid_id) against one launch per object: forward with dy_dx and the value+Jacobian scatter."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from holoscene_amd.hashencoder.backend import _backend as be
from holoscene_amd.hashencoder.hashgrid import level_offsets

dev = "cuda"
L, C, base = 16, 2, 16
pls = np.exp2(np.log2(2048 / base) / (L - 1))
offs = torch.from_numpy(level_offsets(3, L, pls, base, 19)).to(dev)
T, S = int(offs[-1]), float(np.log2(pls))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for G, n in [(8, 4096), (8, 32768), (32, 4096), (32, 1024)]:
    B = G * n
    emb = torch.rand(G, T, C, device=dev) * 2e-4 - 1e-4
    x = torch.rand(B, 3, device=dev)
    gid = torch.arange(G, device=dev, dtype=torch.int32).repeat_interleave(n)
    out, dydx = torch.empty(B, L * C, device=dev), torch.empty(L, B, 3 * C, device=dev)
    gf, gj, ge = torch.randn(B, L * C, device=dev), torch.randn(L, B, 3 * C, device=dev), torch.zeros_like(emb)
    xs = [x[k * n:(k + 1) * n] for k in range(G)]
    outs = [torch.empty(n, L * C, device=dev) for _ in range(G)]
    dys = [torch.empty(L, n, 3 * C, device=dev) for _ in range(G)]
    gfs = [gf[k * n:(k + 1) * n].contiguous() for k in range(G)]
    gjs = [gj[:, k * n:(k + 1) * n].contiguous() for k in range(G)]

    def fwd_each():
        for k in range(G):
            be.fwd(xs[k], emb[k], offs, outs[k], n, 3, C, L, S, base, dys[k])

    def bwd_each():
        for k in range(G):
            be.bwd_jac(gfs[k], gjs[k], xs[k], offs, ge[k], n, 3, C, L, S, base)

    t = dict(fwd_each=timed(fwd_each), fwd_one=timed(lambda: be.fwd(x, emb, offs, out, B, 3, C, L, S, base, dydx, grids=(gid, T))),
             bwd_each=timed(bwd_each), bwd_one=timed(lambda: be.bwd_jac(gf, gj, x, offs, ge, B, 3, C, L, S, base, grids=(gid, T))))
    print(f"G={G:3d} x {n:6d} points: forward+dy_dx {t['fwd_each']:8.1f} us per-object launches -> {t['fwd_one']:7.1f} us batched;   "
          f"value+Jacobian scatter {t['bwd_each']:8.1f} -> {t['bwd_one']:7.1f} us")
