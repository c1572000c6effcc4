"""Throughput of the dense SDF-volume sweep (reference: utils/plots.py:154-205, 100 000-point chunks with a
host copy each) through the fused kernels, K = 32, stock grid."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.model.network import ObjectImplicitNetworkGrid
from holoscene_amd.utils.sdf_grid import evaluate_sdf_volume
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    net = ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=32, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                    divide_factor=1.0, sigmoid=10, color_grid_feature=True).cuda()
for prec in ("bf16", "fp32"):
    net.set_mlp_precision(prec)
    for res, kind in ((256, "min"), (512, "min"), (512, "raw")) if prec == "bf16" else ((256, "min"),):
        evaluate_sdf_volume(net, 64, kind=kind)
        torch.cuda.synchronize()
        v = evaluate_sdf_volume(net, res, kind=kind)      # first call pays hipMalloc of the result (17 GB for 512^3 raw)
        del v
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = evaluate_sdf_volume(net, res, kind=kind)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{prec} {res}^3 {kind}: {dt*1e3:.1f} ms  {res**3/dt/1e9:.2f} G points/s")
        del v
