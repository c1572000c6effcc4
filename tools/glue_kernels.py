#!/usr/bin/env python3
"""tools/glue_kernels.py -- which lines of this package launch the ATen (non-HIP-library) kernels of one training iteration.
Runs the benchmarked trainer eagerly under torch.profiler with Python stacks and lists every aten op that launched a device kernel
with the innermost holoscene_amd frame."""
import sys
import torch
sys.path.insert(0, ".")
from torch.profiler import profile, ProfilerActivity
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf

from holoscene_amd.training.synthetic import SyntheticScene
tr = Stage1Trainer(stock_conf(num_rays=1024, S=128, d_out=32, beta=0.001, mlp_precision="bf16", learning_rate=5e-10), device="cuda", optimizer="flat",
                   graph=True)
benchmark_model_state(tr.model, 0.001)
scene = SyntheticScene(1024, 32, seed=1234, device="cuda")
for _ in range(4):
    tr.train_step_resident(scene)
key = next(k for k in tr._graphs if k[0] == "full" and not k[1])
st = tr._graphs[key]["static"]
run = lambda: tr._full_body(st, False, key[2])  # noqa: E731
run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    run()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.device_type.name == "CPU":
        kern = list(getattr(ev, "kernels", []))
        if ev.name.startswith("aten::") and kern:
            rows.append((kern[0].time_range.start if hasattr(kern[0], "time_range") else ev.time_range.start, f"{ev.name:24s} {sum(k.duration for k in kern):6.1f} us  shapes {ev.input_shapes}"))
    elif ev.name.startswith("k_") or "hs_" in ev.name:
        rows.append((ev.time_range.start, f"    [{ev.name[:50]}]"))
rows.sort()
for _, line in rows:
    print(line)
