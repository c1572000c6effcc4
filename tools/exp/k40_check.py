"""K = 40 (32 < K < 64: padded output tile) on the bf16 value+Jacobian tile kernels against the fp32 path: per-object SDFs, minimum, gradient of the minimum."""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from holoscene_amd.model.network import HoloSceneNetwork
from holoscene_amd.training.trainer import stock_conf, benchmark_model_state

for K in (32, 40, 48, 64):
    torch.manual_seed(0)
    m = HoloSceneNetwork(stock_conf(num_rays=64, S=32, d_out=K, mlp_precision="bf16", logmap=15, end_size=512).get_config("model")).cuda().train()
    benchmark_model_state(m, 0.05)
    with torch.no_grad():
        l2 = m.implicit_network.lin2
        l2.weight_v += 0.05 * torch.randn_like(l2.weight_v) * l2.weight_v.abs().mean()
        l2.bias += 0.1 * torch.randn_like(l2.bias)
    net = m.implicit_network
    x = (torch.rand(3000, 3, device="cuda") * 1.6 - 0.8)
    res = {}
    for prec in ("bf16", "fp32"):
        net.set_mlp_precision(prec)
        y, J = net.sdf_and_jacobian(x)
        y, J = y[:, :K], J[:, :K]
        sdf, idx = y.min(-1, keepdim=True)
        g = torch.gather(J, 1, idx.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
        res[prec] = (y.detach(), sdf.detach(), idx, g.detach(), J.detach())
    a, b = res["bf16"], res["fp32"]
    same = (a[2] == b[2]).float().mean()
    rel = lambda p, q: float((p - q).norm() / q.norm())
    print(f"K={K}: y relL2 {rel(a[0], b[0]):.3e}  min relL2 {rel(a[1], b[1]):.3e}  argmin agree {float(same):.4f}  grad(min) relL2 on agreeing {rel(a[3][(a[2] == b[2]).view(-1)], b[3][(a[2] == b[2]).view(-1)]):.3e}  "
          f"J relL2 {rel(a[4], b[4]):.3e}  J per column max relL2 {max(rel(a[4][:, k], b[4][:, k]) for k in range(K)):.3e}")
