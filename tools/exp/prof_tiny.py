import os, sys, torch
sys.path.insert(0, '/root/repo')
from holoscene_amd.training.synthetic import SyntheticScene
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
tr = Stage1Trainer(stock_conf(beta=0.001, mlp_precision='bf16'), device='cuda', optimizer='flat', graph=False)
benchmark_model_state(tr.model, 0.001)
scene = SyntheticScene(1024, 32, num_frames=8, ring=64, device='cuda')
for _ in range(3): tr.train_step_resident(scene)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step_resident(scene)
    torch.cuda.synchronize()
names = ("aten::fill_", "aten::zero_", "aten::abs", "aten::add", "aten::copy_", "aten::sign", "aten::mul", "aten::sum", "aten::cat", "aten::index", "aten::gather", "aten::scatter", "aten::_foreach", "aten::floor", "aten::arange", "aten::uniform_", "aten::rand", "aten::sub", "aten::div", "aten::neg", "aten::where", "aten::clamp", "aten::index_select", "aten::scatter_add", "aten::_to_copy")
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and any(e.name.startswith(n) for n in names)]
seen = set()
for e in evs:
    st = [s for s in (e.stack or []) if 'holoscene_amd' in s]
    key = (e.name, st[0] if st else '')
    if key in seen: continue
    seen.add(key)
    print(f"{e.name:28s} cuda {e.device_time_total if hasattr(e,'device_time_total') else 0:8.1f}us  {st[0] if st else '?'}")
