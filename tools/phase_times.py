#!/usr/bin/env python3
"""tools/phase_times.py -- wall time per phase of one Stage-1 iteration on the GPU (dev tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.training.synthetic import SyntheticScene  # noqa: E402
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf  # noqa: E402

beta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
dev = "cuda"
tr = Stage1Trainer(stock_conf(beta=beta), device=dev)
benchmark_model_state(tr.model, beta)
scene = SyntheticScene(1024, 32, device=dev)
model = tr.model
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


orig_z = model.ray_sampler.get_z_vals
orig_out = model.implicit_network.get_outputs
orig_jac = model.implicit_network.sdf_and_jacobian
orig_render = model.rendering_network.forward


def wrap(fn, name):
    def f(*a, **k):
        mark("pre:" + name)
        r = fn(*a, **k)
        mark(name)
        return r
    return f


model.ray_sampler.get_z_vals = wrap(orig_z, "sampler")
model.implicit_network.get_outputs = wrap(orig_out, "get_outputs")
model.rendering_network.forward = wrap(orig_render, "render_mlp")

for it in range(8):
    idx, mi, gt = scene.next_batch()
    marks.clear()
    mark("start")
    model.train()
    tr.optimizer.zero_grad(set_to_none=True)
    out = model(mi, idx, iter_step=tr.iter_step if it != 7 else 3)
    mark("forward_end")
    out["iter_step"] = 3
    lo = tr.loss(out, gt)
    mark("loss")
    lo["loss"].backward()
    mark("backward")
    tr.optimizer.step()
    tr.scheduler.step()
    mark("adam")
    tr.iter_step += 1
    torch.cuda.synchronize()
    if it >= 6:
        print(f"--- iteration {it} (rounds {model.ray_sampler.last_rounds}, bg={'bg_depth_values' in out})")
        prev = marks[0][1]
        for name, e in marks[1:]:
            print(f"{name:20s} {prev.elapsed_time(e):8.3f} ms")
            prev = e
        print(f"{'TOTAL':20s} {marks[0][1].elapsed_time(marks[-1][1]):8.3f} ms")
