#!/usr/bin/env python3
"""tools/bench_stage2_entry.py -- time the Stage-2/3 entry points on the stock-shaped model in bf16 mode:
1 024 world-space rays against an object subset, forward only (no_grad, eval) and forward + backward (train)."""
import sys, time
import torch
sys.path.insert(0, ".")
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf

dev = "cuda"
tr = Stage1Trainer(stock_conf(num_rays=1024, S=128, d_out=32, beta=0.01, mlp_precision="bf16"), device=dev, optimizer="flat", graph=False)
model = tr.model
benchmark_model_state(model, 0.01)
R = 1024
g = torch.Generator().manual_seed(0)
o = (torch.randn(R, 3, generator=g) * 0.05 + torch.tensor([0.7, 0.0, 0.0])).to(dev)
d = (-o + torch.randn(R, 3, generator=g).to(dev) * 0.3)
pose = torch.eye(4, device=dev)[None]
objs, subset = [3, 7, 9], [0, 3, 7, 9]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def train_call(f):
    def run():
        model.zero_grad(set_to_none=True)
        out = f()
        (out["rgb_values"].sum() + out["depth_values"].sum() + out["normal_map"].sum()).backward()
    return run


cases = {
    "forward_multi_obj_rays": lambda: model.forward_multi_obj_rays(o, d, pose, objs),
    "forward_only_multi_obj_rays": lambda: model.forward_only_multi_obj_rays(o, d, pose, objs),
    "forward_multi_obj_rays_subset_all_sdf": lambda: model.forward_multi_obj_rays_subset_all_sdf(o, d, pose, objs, subset),
    "forward_multi_obj_rays_subset_all_sdf_near_far": lambda: model.forward_multi_obj_rays_subset_all_sdf_near_far(o, d, pose, objs, subset, 0.05, 2.0),
    "get_colors_from_point_rays_obj": lambda: {"rgb_values": model.get_colors_from_point_rays_obj(o, d, 3), "depth_values": torch.zeros(1, device=dev, requires_grad=True), "normal_map": torch.zeros(1, device=dev)},
}
for name, f in cases.items():
    model.eval()
    with torch.no_grad():
        t_eval = timeit(f)
    model.train()
    t_train = timeit(train_call(f))
    print(f"{name:52s} eval fwd {t_eval:7.2f} ms   train fwd+bwd {t_train:7.2f} ms   (1024 rays, sampler rounds {model.ray_sampler.last_rounds})")
