import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.model.network import ObjectImplicitNetworkGrid
dev = 'cuda'
torch.manual_seed(0)
net = ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=32, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                divide_factor=1.0, sigmoid=10, color_grid_feature=True).to(dev)
net.set_mlp_precision('bf16')
B = 131072
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
with torch.no_grad():
    feat = net.encoding(x)
    w0, b0, w1, b1, w2, b2 = net._packed_weights()
    out = torch.empty(B, 1, device=dev)
    t = timeit(lambda: be.sdf_mlp_fwd(x, feat, w0, b0, w1, b1, w2, b2, 32, -1, out, None))
    fl = B * 2 * (96 * 256 + 256 * 256 + 256 * 32)
    print(f"k_sdf_mlp B={B}: {t:.1f} us  {fl / t / 1e6:.1f} TFLOP/s (padded flops)  whole get_sdf_vals {timeit(lambda: net.get_sdf_vals(x)):.1f} us")
    net.set_mlp_precision('fp32')
    print(f"torch fp32 get_sdf_vals {timeit(lambda: net.get_sdf_vals(x)):.1f} us")
