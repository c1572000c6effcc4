"""GPU time of the fp32 fused SDF sweep (csrc/sdf_mlp32.hip) against the library-GEMM form, 131 072 points, K = 32: 20 sweeps in one graph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from holoscene_amd.model import network as N
torch.manual_seed(0)
net = N.ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=32, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6, divide_factor=1.0,
                                  sigmoid=10, color_grid_feature=True, num_levels=16, logmap=19, end_size=2048).cuda()
net.set_mlp_precision("fp32")
B = 131072
x = torch.rand(B, 3, device="cuda") * 2 - 1
x01 = ((x + 1) / 2).contiguous()
def t(fn, n=20):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(n): fn()
        torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
print("fused fp32 sweep (gather + k_sdf_mlp32): %.1f us" % t(lambda: net.sdf_at_points(x, x01, 1024, 128)))
N.ops.FP32_SDF = "gemm"
print("library GEMM form (get_sdf_vals):       %.1f us" % t(lambda: net.get_sdf_vals(x)))
N.ops.FP32_SDF = "mfma"
net.set_mlp_precision("bf16")
print("bf16 sweep (gather + k_sdf_mlp2):        %.1f us" % t(lambda: net.sdf_at_points(x, x01, 1024, 128)))
