#!/usr/bin/env python3
"""tools/microbench_hash.py -- per-level timing of the hash-grid kernels on the GPU box (dev tool, not a test).

Points are laid out like the main render set: R rays x N samples from one camera, so neighbouring
lanes are neighbouring samples of one ray (the locality the kernels see in training)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.hashencoder import backend  # noqa: E402
from holoscene_amd.hashencoder.hashgrid import level_offsets  # noqa: E402

dev = "cuda"
be = backend._backend


def ray_points(R=1024, N=98, seed=0):
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.7, 0.0, 0.0])
    d = torch.nn.functional.normalize(torch.tensor([-1.0, 0, 0]) + (torch.rand(R, 3, generator=g) - 0.5), dim=-1)
    z = torch.sort(torch.rand(R, N, generator=g) * 2.2, -1)[0]
    x = o + z[..., None] * d[:, None]
    return ((x.reshape(-1, 3) + 1) / 2).contiguous()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def main():
    L, base, end, logmap = 16, 16, 2048, 19
    pls = np.exp2(np.log2(end / base) / (L - 1))
    offs = level_offsets(3, L, pls, base, logmap)
    S = float(np.log2(pls))
    x = ray_points().to(dev)
    B = x.shape[0]
    inside = ((x >= 0) & (x <= 1)).all(-1).float().mean().item()
    print(f"B={B} inside={inside:.2f}")
    emb = torch.rand(int(offs[-1]), 2, device=dev) * 2e-4 - 1e-4
    od = torch.from_numpy(offs).to(dev)
    out = torch.empty(B, L * 2, device=dev)
    dydx = torch.empty(L, B, 6, device=dev)
    g = torch.randn(B, L * 2, device=dev)
    gj = torch.randn(L, B, 6, device=dev)
    ge = torch.zeros_like(emb)
    for sched in (0, 1):
        backend.SCHEDULE = sched
        print(f"schedule {sched}: fwd {timeit(lambda: be.fwd(x, emb, od, out, B, 3, 2, L, S, base, None)):.1f} us  "
              f"fwd+dydx {timeit(lambda: be.fwd(x, emb, od, out, B, 3, 2, L, S, base, dydx)):.1f} us  "
              f"scatter {timeit(lambda: be.bwd(g, x, od, ge, B, 3, 2, L, S, base, None, None), 5):.1f} us  "
              f"bwd_jac {timeit(lambda: be.bwd_jac(g, gj, x, od, ge, B, 3, 2, L, S, base), 5):.1f} us")
    # random points (no ray locality)
    xr = torch.rand(B, 3, device=dev)
    print(f"random pts: fwd {timeit(lambda: be.fwd(xr, emb, od, out, B, 3, 2, L, S, base, None)):.1f} us  "
          f"scatter {timeit(lambda: be.bwd(g, xr, od, ge, B, 3, 2, L, S, base, None, None), 5):.1f} us")
    # per level: a one-level encoder with H = that level's resolution
    backend.SCHEDULE = 0
    for l in range(L):
        res = int(np.ceil(base * pls ** l))
        n = int(offs[l + 1] - offs[l])
        o1 = torch.tensor([0, n], dtype=torch.int32, device=dev)
        e1 = torch.zeros(n, 2, device=dev)
        g1 = torch.randn(B, 2, device=dev)
        o_1 = torch.empty(B, 2, device=dev)
        tf = timeit(lambda: be.fwd(x, e1, o1, o_1, B, 3, 2, 1, 0.0, res, None))
        ts = timeit(lambda: be.bwd(g1, x, o1, e1, B, 3, 2, 1, 0.0, res, None, None), 5)
        tr = timeit(lambda: be.bwd(g1, xr, o1, e1, B, 3, 2, 1, 0.0, res, None, None), 5)
        print(f"level {l:2d} res {res:4d} table {n:7d}: fwd {tf:7.1f} us  scatter(ray pts) {ts:8.1f} us  scatter(random pts) {tr:8.1f} us")
    # raw atomic rate: B*16 float atomics to random addresses of a 4 MiB table via index_add_ (torch) for reference
    idx = torch.randint(0, 1 << 20, (B * 16,), device=dev)
    val = torch.randn(B * 16, device=dev)
    tab = torch.zeros(1 << 20, device=dev)
    print(f"torch index_add_ {B * 16} random float atomics: {timeit(lambda: tab.index_add_(0, idx, val), 5):.1f} us")


if __name__ == "__main__":
    main()
