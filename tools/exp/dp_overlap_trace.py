"""Development check: a few iterations of the stock-size trainer with the overlapped exchange (one-rank RCCL group), to be run under
rocprofv3 --kernel-trace; tools/exp/dp_overlap_trace_report.py then lists where the early segment's kernels ran."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from holoscene_amd.training.synthetic import SyntheticScene  # noqa: E402
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "overlap"
conf = stock_conf(num_rays=1024, S=128, d_out=32, beta=0.001, mlp_precision="bf16", learning_rate=5e-10)
tr = Stage1Trainer(conf, device=dev, optimizer="flat", graph=True, data_parallel=True, exchange=mode)
benchmark_model_state(tr.model, 0.001)
scene = SyntheticScene(1024, 32, seed=1234, device=dev)
for _ in range(12):
    tr.train_step_resident(scene)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    tr.train_step_resident(scene)
    ev[i + 1].record()
torch.cuda.synchronize()
t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40))
print("mode", mode, "overlap", tr._overlap, "median ms", round(t[20], 3), "mean ms", round(sum(t) / 40, 3), flush=True)
dist.destroy_process_group()
