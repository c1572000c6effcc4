"""Time k_rr_fwd against k_rr_fwd_value + k_rr_fwd_grad at the Stage-1 batch (131 072 samples, K = 32)."""
import sys
import torch
from holoscene_amd.hashencoder.backend import _backend as be

n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 32
dev, bf = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731
x = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev)
feat, dydx = rn(n, 32, sc=0.1), rn(16, n, 6, sc=0.5)
W0, b0, W1, b1, W2, b2 = rn(256, 71, sc=0.15), rn(256, sc=0.05), rn(256, 256, sc=0.08), rn(256, sc=0.05), rn(K, 256, sc=0.1), rn(K, sc=0.1)
packed = be.sdf_mlp2_pack(W0, b0, W1, b1, W2, b2, K, log2_domain=False)
rr = be.trunk_rr_pack(W0, W1, W2, K)
M = be.tp_rows(n)
tp = lambda: torch.empty(M * 256, device=dev, dtype=bf)  # noqa: E731
H0t, H1t, U0t, V1t, V0t = tp(), tp(), tp(), tp(), tp()
Xp, onehot = torch.zeros(n, 80, device=dev, dtype=bf), torch.zeros(n, 32, device=dev, dtype=bf)
raw, sdf, idx = torch.empty(n, K, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev, dtype=torch.int64)
grad, uxh = torch.empty(n, 3, device=dev), torch.empty(n, 32, device=dev)


def split():
    be.trunk_rr_fwd_value(x, feat, packed, K, H0t, H1t, Xp, raw, sdf, idx, onehot)
    be.trunk_rr_fwd_grad(x, dydx, idx, rr, H0t, H1t, U0t, V1t, V0t, grad, uxh, 0.5)


def fused():
    be.trunk_rr_fwd(x, feat, dydx, packed, rr, K, H0t, H1t, Xp, raw, sdf, idx, onehot, U0t, V1t, V0t, grad, uxh, 0.5)


for name, fn in (("split", split), ("fused", fused), ("split", split), ("fused", fused)):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(name, round(a.elapsed_time(b) / 50 * 1000, 1), "us")
