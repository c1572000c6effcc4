"""Do independent branches of a captured HIP graph run concurrently on this platform?  hash gather (memory-bound) vs fused
SDF MLP (matrix-core-bound) on disjoint data: serial graph vs forked graph vs eager two-stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.model.network import ObjectImplicitNetworkGrid
from holoscene_amd.hashencoder import backend
import numpy as np
dev = 'cuda'
torch.manual_seed(0)
net = ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=32, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                divide_factor=1.0, sigmoid=10, color_grid_feature=True).to(dev)
net.set_mlp_precision('bf16')
be = backend._backend
B = 131072
enc = net.encoding
L, C = enc.num_levels, enc.level_dim
S_, H_ = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
with torch.no_grad():
    xa, xb = torch.rand(B, 3, device=dev), torch.rand(B, 3, device=dev)
    fa, fb = torch.empty(B, L * C, device=dev), torch.randn(B, L * C, device=dev) * 1e-3
    out = torch.empty(B, 1, device=dev)
    w = net._packed_weights()
    def hash_():
        be.fwd(xa, enc.embeddings, enc.offsets, fa, B, 3, C, L, S_, H_, None)
    def mlp_():
        be.sdf_mlp_fwd(xb, fb, *w, 32, -1, out, None)
    def timeit(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    print(f"hash alone {timeit(hash_):.1f} us, mlp alone {timeit(mlp_):.1f} us")
    side = torch.cuda.Stream()
    def serial():
        for _ in range(4): hash_(); mlp_()
    def forked():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(4): hash_()
        for _ in range(4): mlp_()
        cur.wait_stream(side)
    print(f"eager: 4x(hash, mlp) serial {timeit(serial):.1f} us, two streams {timeit(forked):.1f} us")
    for name, body in (("serial", serial), ("forked", forked)):
        body(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        print(f"graph {name}: {timeit(g.replay):.1f} us")
