"""What could weight gradients accumulated INSIDE the colour branch's backward kernel save at most?  

Measured at the benchmarked size (n = 100 352 samples):
  (a) k_appear2_bwd as shipped, and with its four tile-packed cotangent stores removed (HOLOSCENE_A2_ABLATE=nostore, second process): the
      time the kernel spends on the stores a fused kernel would not issue;
  (b) the colour branch's weight-gradient jobs as shipped (cotangent + saved activation from HBM), and with the cotangent operand aliased to the
      activation operand (the second request of a tile is an L2 hit: the pass then streams the ACTIVATIONS only, which a fused kernel must still read);
  (c) a 256-workgroup kernel's own partials: 256 x (4 x 256 x 256) bf16 written once and summed once -- what the fused kernel has to
      flush instead (torch copy / sum at the same byte counts: the stream rate bounds both).
run:  python tools/exp/dw_fusion_bounds.py ; HOLOSCENE_A2_ABLATE=nostore python tools/exp/dw_fusion_bounds.py bwd
"""
import os
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from holoscene_amd.hashencoder.backend import _backend as be

dev, bf = "cuda", torch.bfloat16
g = torch.Generator().manual_seed(3)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731
n = 100352
tiles = (n + 31) // 32


def timed(fn, k=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(k):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3


tp = lambda ks: torch.zeros(tiles * ks * 64 * 8, device=dev, dtype=bf)  # noqa: E731
featc = rn(16, n, 2, sc=0.3)
points, dirs, normals = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev), torch.nn.functional.normalize(rn(n, 3), dim=-1), torch.nn.functional.normalize(rn(n, 3), dim=-1)
wc0, wc1, wr0, wr1, wr2 = rn(256, 32, sc=0.2), rn(256, 256, sc=0.07), rn(256, 337, sc=0.06), rn(256, 256, sc=0.07), rn(3, 256, sc=0.1)
bs = (rn(256, sc=0.1), rn(256, sc=0.1), rn(256, sc=0.1), rn(256, sc=0.1), rn(3, sc=0.1))
P = be.appearance2_pack(wc0, wc1, wr0, wr1, wr2, bs)
XAt, HCt, FVt, R0t, R1t = tp(8), tp(16), tp(16), tp(16), tp(16)
masks = torch.zeros(tiles * 3 * 64 * 4, device=dev, dtype=torch.int32)
rgb = torch.empty(n, 3, device=dev)
be.appearance2_fwd(featc, points, dirs, normals, P, XAt, HCt, FVt, R0t, R1t, masks, rgb)
g_rgb = rn(n, 3)
gy = torch.empty(n, 32, device=dev, dtype=bf)
GR1, GR0, GFV, GHC = tp(16), tp(16), tp(16), tp(16)
d_n, g_fc, gb2 = torch.empty(n, 3, device=dev), torch.empty(16, n, 2, device=dev), torch.zeros(tiles, 4, device=dev)
bwd = lambda: be.appearance2_bwd(g_rgb, rgb, normals, masks, P["streamT"], gy, GR1, GR0, GFV, GHC, d_n, g_fc, gb2)  # noqa: E731
tag = "nostore" if os.environ.get("HOLOSCENE_A2_ABLATE", "").startswith("n") else "shipped"
print(f"BOUNDS k_appear2_bwd [{tag}] {timed(bwd):.1f} us  (n = {n})")
if len(sys.argv) > 1 and sys.argv[1] == "bwd":
    sys.exit(0)
# (b) the colour branch's big weight-gradient jobs: dW_R1 = r1~^T r0, dW_R0f = r0~^T fv, dW_C1 = fv~^T hc (256 x 256, tile-packed both), dW_C0|enc = [hc~ | r0~]^T XA
S = 64
jobs = lambda al: [((256, 256), S, (GR1, GR1 if al else R0t), None), ((256, 256), S, (GR0, GR0 if al else FVt), None),  # noqa: E731
                   ((256, 256), S, (GFV, GFV if al else HCt), None)]
for al in (False, True):
    js = jobs(al)
    us = timed(lambda: be.wgrad_pairs(js, n))
    nbytes = 3 * (1 if al else 2) * tiles * 32 * 512
    print(f"BOUNDS wgrad 3 x (256 x 256) jobs, {'one operand stream (B aliased to A)' if al else 'two operand streams (shipped)'}: {us:.1f} us, {nbytes / 1e6:.0f} MB distinct, {nbytes / us / 1e6:.2f} TB/s")
# (c) the partials a fused kernel would flush: 256 workgroups x 4 layers x 256 x 256, bf16
part = torch.zeros(256, 4, 256, 256, device=dev, dtype=bf)
src = torch.ones(256, 4, 256, 256, device=dev, dtype=bf)
w_us = timed(lambda: part.copy_(src))
out = torch.empty(4, 256, 256, device=dev)
r_us = timed(lambda: torch.sum(part, 0, dtype=torch.float32, out=out))
print(f"BOUNDS fused-kernel partials {part.numel() * 2 / 1e6:.0f} MB: written in >= {w_us:.1f} us (copy: reads as much as it writes), summed in {r_us:.1f} us")
