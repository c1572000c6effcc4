"""k_sdf_mlp (workgroup tile) vs k_sdf_mlp2 (wave tile): time per 131 072-point sweep and agreement with each other and with the
fp32 torch trunk.  Run on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.model import network
from holoscene_amd.model.network import ObjectImplicitNetworkGrid
dev = 'cuda'
torch.manual_seed(0)
K = int(os.environ.get("K", 32))
net = ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=K, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                divide_factor=1.0, sigmoid=10, color_grid_feature=True).to(dev)
with torch.no_grad():
    net.lin0.weight_v[:, 3:].normal_(0, 1e-2)
    net.lin0.bias.normal_(0, 1e-2)
    net.lin1.bias.normal_(0, 1e-2)
    net.encoding.embeddings.uniform_(-0.5, 0.5)
    net.lin2.weight_v[:K] += 0.05 * torch.randn(K, 256, device=dev) * net.lin2.weight_v[:K].abs().mean()
net.set_mlp_precision('bf16')
B = int(os.environ.get("B", 131072))
x = torch.rand(B, 3, device=dev) * 2 - 1
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
from holoscene_amd.hashencoder import backend
be = backend._backend
fl = B * 2 * (71 * 256 + 256 * 256 + 256 * K)
with torch.no_grad():
    x01 = ((x + 1) / 2).contiguous()
    enc = net.encoding
    feat = torch.empty(16, B, 2, device=dev)
    import numpy as np
    be.fwd(x01, enc.embeddings, enc.offsets, feat, B, 3, 2, 16, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), None, level_major=True)
    w0, b0, w1, b1, w2, b2 = net._packed_weights()
    packed = net._packed_weights2() if K <= 32 else None
    o1, o2 = torch.empty(B, 1, device=dev), torch.empty(B, 1, device=dev)
    r1, r2 = torch.empty(B, K, device=dev), torch.empty(B, K, device=dev)
    be.sdf_mlp_fwd(x, feat, w0, b0, w1, b1, w2, b2, K, -1, o1, r1, feat_level_major=True)
    if K <= 32:
        be.sdf_mlp2_fwd(x, feat, packed, K, -1, o2, r2, feat_level_major=True)
    else:
        o2.copy_(o1); r2.copy_(r1)
    torch.cuda.synchronize()
    net.set_mlp_precision('fp32')
    ref = net.get_sdf_raw(x)
    net.set_mlp_precision('bf16')
    sc = float(ref.abs().max())
    print(f"max|tile - fp32| {float((r1 - ref).abs().max()) / sc:.3e}  max|wave - fp32| {float((r2 - ref).abs().max()) / sc:.3e}  "
          f"max|wave - tile| {float((r2 - r1).abs().max()) / sc:.3e}  min: {float((o2 - o1).abs().max()) / sc:.3e}  (relative to max|sdf| = {sc:.3f})")
    if K > 32:
        t1 = timeit(lambda: be.sdf_mlp_fwd(x, feat, w0, b0, w1, b1, w2, b2, K, -1, o1, None, feat_level_major=True))
        print(f"B={B} K={K}: k_sdf_mlp {t1:.1f} us ({fl / t1 / 1e6:.0f} TF unpadded)")
        sys.exit(0)
    # point-major features and a single object / an object subset
    featp = feat.permute(1, 0, 2).reshape(B, 32).contiguous()
    o3 = torch.empty(B, 1, device=dev)
    be.sdf_mlp2_fwd(x, featp, packed, K, 3, o3, None, feat_level_major=False)
    print("select=3 (point-major feats) vs raw column:", float((o3[:, 0] - r2[:, 3]).abs().max()))
    be.sdf_mlp2_fwd(x, feat, packed, K, [1, 4, 7], o3, None, feat_level_major=True)
    print("subset [1,4,7] vs raw columns:", float((o3[:, 0] - r2[:, [1, 4, 7]].min(-1)[0]).abs().max()))
    t1 = timeit(lambda: be.sdf_mlp_fwd(x, feat, w0, b0, w1, b1, w2, b2, K, -1, o1, None, feat_level_major=True))
    t2 = timeit(lambda: be.sdf_mlp2_fwd(x, feat, packed, K, -1, o2, None, feat_level_major=True))
    print(f"B={B} K={K}: k_sdf_mlp {t1:.1f} us ({fl / t1 / 1e6:.0f} TF unpadded)   k_sdf_mlp2 {t2:.1f} us ({fl / t2 / 1e6:.0f} TF unpadded)")
    tp = timeit(lambda: (net.invalidate_packed_weights(), net._packed_weights2()))
    print(f"pack2 {tp:.1f} us")
    for Bs in (4096, 1000, 33):
        xs = x[:Bs].contiguous(); fs = feat[:, :Bs].contiguous(); oo = torch.empty(Bs, 1, device=dev); rr = torch.empty(Bs, K, device=dev)
        be.sdf_mlp2_fwd(xs, fs, packed, K, -1, oo, rr, feat_level_major=True)
        print(f"B={Bs}: max|wave - tile| {float((rr - r1[:Bs]).abs().max()) / sc:.3e}  {timeit(lambda: be.sdf_mlp2_fwd(xs, fs, packed, K, -1, oo, None, feat_level_major=True)):.1f} us")
    # ---- fixed cost vs per-tile cost of the wave-tile kernel
    closed = (torch.zeros(1, device=dev), torch.ones(1, device=dev))      # gate: run only if a > b
    print(f"gate closed (launch only): {timeit(lambda: be.sdf_mlp2_fwd(x, feat, packed, K, -1, o2, None, gate=closed, feat_level_major=True)):.1f} us")
    for Bs in (32 * 8 * 256, 2 * 32 * 8 * 256, 3 * 32 * 8 * 256, 4 * 32 * 8 * 256):
        xs = torch.rand(Bs, 3, device=dev) * 2 - 1
        fs = torch.rand(16, Bs, 2, device=dev) - 0.5
        oo = torch.empty(Bs, 1, device=dev)
        print(f"B={Bs} ({Bs // (32 * 8 * 256)} tiles per wave): {timeit(lambda: be.sdf_mlp2_fwd(xs, fs, packed, K, -1, oo, None, feat_level_major=True)):.1f} us")
