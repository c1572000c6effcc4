"""hs_gemm_split_nt / _tn against the library's fp32 GEMMs at the fp32 path's shapes (M = 4 rows x 100 k points)."""
import torch
from holoscene_amd.hashencoder.backend import _backend as be

dev = "cuda"
M = 401408


def t(fn, reps=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for N, K in ((256, 256), (256, 80), (32, 256), (256, 337)):
    x, w, g = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.1, torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K
    tl = t(lambda: x @ w.t())
    t3, t2 = t(lambda: be.gemm_split_nt(x, w, None, 3)), t(lambda: be.gemm_split_nt(x, w, None, 2))
    print(f"NT {M}x{N}x{K}: library {tl * 1e6:7.0f} us {fl / tl / 1e12:6.1f} TF/s | 3 planes {t3 * 1e6:7.0f} us {fl / t3 / 1e12:6.1f} | 2 planes {t2 * 1e6:7.0f} us {fl / t2 / 1e12:6.1f}")
    S = 256 if N * K >= 65536 else 1024
    tl = t(lambda: torch.bmm(g.view(S, M // S, -1).transpose(1, 2), x.view(S, M // S, -1)).sum(0))
    t3, t2 = t(lambda: be.gemm_split_tn(g, x, S, 3).sum(0)), t(lambda: be.gemm_split_tn(g, x, S, 2).sum(0))
    print(f"TN {M}x{N}x{K}: library {tl * 1e6:7.0f} us {fl / tl / 1e12:6.1f} TF/s | 3 planes {t3 * 1e6:7.0f} us {fl / t3 / 1e12:6.1f} | 2 planes {t2 * 1e6:7.0f} us {fl / t2 / 1e12:6.1f}")
