#!/usr/bin/env python3
"""tools/exp/draw_ahead_long.py -- the draw-ahead path across ring refills: 1 300 iterations (frozen parameters, identically seeded scenes) with the batch drawn one
iteration ahead inside the graph vs drawn eagerly in front of every replay; the objectives must agree at every iteration (ring of 512 batches: two refills)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from holoscene_amd.training.synthetic import SyntheticScene
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf

dev = torch.device("cuda", 0)
N = 1300
res = {}
for mode in (True, False):
    torch.manual_seed(3)
    tr = Stage1Trainer(stock_conf(num_rays=256, S=32, d_out=4, num_levels=16, end_size=512, logmap=15, beta=0.05, mlp_precision="bf16"), device=dev,
                       optimizer="flat", graph=True, freeze_parameters=True, draw_in_graph=mode)
    benchmark_model_state(tr.model, 0.05)
    scene = SyntheticScene(256, 4, img_res=(64, 64), num_frames=5, seed=9, device=dev)
    out = []
    for i in range(N):
        _, lo = tr.train_step_resident(scene)
        out.append(lo["loss"].detach().clone())
    res[mode] = torch.stack(out).cpu()
    print(mode, "allocated MB", torch.cuda.memory_allocated() // 2 ** 20, flush=True)
d = (res[True] - res[False]).abs() / res[False].abs()
print("max relative objective difference over", N, "iterations:", float(d.max()), "at", int(d.argmax()), "| distinct objectives:", len(set(res[False].tolist())))
assert float(d.max()) <= 2e-4
