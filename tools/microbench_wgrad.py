#!/usr/bin/env python3
"""tools/microbench_wgrad.py -- hs_wgrad_rows (csrc/wgrad.hip) vs the library's split-M batched GEMM on the appearance / trunk shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.hashencoder import backend as B  # noqa: E402
be = B._backend
dev, bf = "cuda", torch.bfloat16


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
    return s.elapsed_time(e) / n * 1e3


def lib(pairs, S=128):
    return [torch.bmm(g.view(S, g.shape[0] // S, -1).transpose(1, 2), x.view(S, x.shape[0] // S, -1)) for g, x in pairs]


for M, label in ((100352, "appearance (M = 100 352)"), (417792, "trunk (M = 417 792)")):
    t = {w: torch.randn(M, w, device=dev).to(bf) for w in (32, 128, 256)}
    o = torch.randn(M, 256, device=dev).to(bf)
    sets = {"256x256": [(t[256], o)], "2 x 256x256": [(t[256], o), (o, t[256])], "256x128": [(t[256], t[128])], "32x256": [(t[32], o)]}
    if M == 100352:
        sets["the six appearance products"] = [(t[32], o), (t[256], o), (t[256], t[128]), (o, t[256]), (t[256], o), (o, t[128])]
    print(label)
    for name, pairs in sets.items():
        mb = sum((g.numel() + x.numel()) * 2 for g, x in pairs) / 1e6
        a, b = timeit(lambda: be.wgrad_rows(pairs, 128)), timeit(lambda: lib(pairs))
        print(f"  {name:28s} {mb:7.1f} MB of operands: hs_wgrad_rows {a:7.1f} us ({mb / a / 1e3 * 1e3 / 1e3:.2f} TB/s)   library bmm {b:7.1f} us ({mb / b:.2f} TB/s)".replace("TB/s)   library", "TB/s)   library"))
