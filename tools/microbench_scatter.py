"""Scatter kernels (hs_hash_bwd / hs_hash_bwd_jac) with global atomics vs binned records, dense and sparse cotangents, on
ray-ordered points (1 024 rays x 98 samples)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.hashencoder import backend
from holoscene_amd.hashencoder.hashgrid import HashEncoder
be = backend._backend
dev = 'cuda'
torch.manual_seed(0)
enc = HashEncoder(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(dev)
L, C = 16, 2
S_, H_ = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
R, N = 1024, 98
o = torch.rand(R, 1, 3, device=dev) * 0.2 + 0.4
d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
z = torch.sort(torch.rand(R, N, 1, device=dev) * 0.5, 1)[0]
x = (o + z * d).reshape(-1, 3).clamp(0, 1).contiguous()
B = x.shape[0]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
ge = torch.zeros_like(enc.embeddings)
ws = be.scatter_workspace(B, 3, C, L, dev)
for name, keep in (("dense", 1.0), ("sparse 25%", 0.25), ("sparse 5%", 0.05)):
    g = torch.randn(B, L * C, device=dev)
    gj = torch.randn(L, B, 6, device=dev)
    mask = (torch.rand(B, device=dev) < keep).float()
    g = (g * mask[:, None]).contiguous(); gj = (gj * mask[None, :, None]).contiguous()
    ta = timeit(lambda: be.bwd(g, x, enc.offsets, ge, B, 3, C, L, S_, H_, None, None))
    tb = timeit(lambda: be.bwd(g, x, enc.offsets, ge, B, 3, C, L, S_, H_, None, None, ws=ws))
    tc = timeit(lambda: be.bwd_jac(g, gj, x, enc.offsets, ge, B, 3, C, L, S_, H_))
    td = timeit(lambda: be.bwd_jac(g, gj, x, enc.offsets, ge, B, 3, C, L, S_, H_, ws=ws))
    print(f"{name:11s} B={B}: bwd atomic {ta:7.1f} us  binned {tb:7.1f} us | bwd_jac atomic {tc:7.1f} us  binned {td:7.1f} us")
