"""Which ATen ops with device kernels run per iteration OUTSIDE the replayed graph (train_step_resident on the graph path)."""
import sys
import torch
sys.path.insert(0, ".")
from torch.profiler import profile, ProfilerActivity
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
from holoscene_amd.training.synthetic import SyntheticScene
tr = Stage1Trainer(stock_conf(num_rays=1024, S=128, d_out=32, beta=0.001, mlp_precision="bf16", learning_rate=5e-10), device="cuda", optimizer="flat", graph=True)
benchmark_model_state(tr.model, 0.001)
scene = SyntheticScene(1024, 32, seed=1234, device="cuda")
for _ in range(12):
    tr.train_step_resident(scene)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.train_step_resident(scene)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.device_type.name == "CPU" and ev.name.startswith("aten::") and list(getattr(ev, "kernels", [])):
        print(ev.name, ev.input_shapes, [k.name[:50] for k in ev.kernels], (ev.stack or ["?"])[:6])
