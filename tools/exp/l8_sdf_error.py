"""bf16 fused SDF sweep against the fp32 evaluation of the same network at random points: the eight-level fixtures beside a sixteen-level one."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from helpers import load
from model_helpers import build_model
torch.manual_seed(0)
x = (torch.rand(20000, 3, device="cuda") * 2 - 1) * 0.9
for name in ("stock_k21", "stock_l8_k5", "stock_l8_k3_bg"):
    rec = load(name)
    m32 = build_model(rec, "cuda").eval()
    mb = build_model(rec, "cuda").eval()
    mb.implicit_network.set_mlp_precision("bf16")
    with torch.no_grad():
        a = m32.implicit_network.get_sdf_vals(x)
        b = mb.implicit_network.get_sdf_vals(x)
        # roughness: |sdf(x + 1e-3 e) - sdf(x)| / 1e-3
        d = torch.zeros_like(x); d[:, 0] = 1e-3
        g = (m32.implicit_network.get_sdf_vals(x + d) - a).abs() / 1e-3
        enc = m32.implicit_network.encoding
        f = enc(x)
    print(f"{name}: sdf rms {float(a.pow(2).mean().sqrt()):.3f}  bf16 error rms {float((a - b).pow(2).mean().sqrt()):.2e} max {float((a - b).abs().max()):.2e}  "
          f"|d sdf / dx| median {float(g.median()):.2f} p99 {float(g.quantile(0.99)):.1f}  feature rms {float(f.pow(2).mean().sqrt()):.2e}  L {enc.num_levels}")
