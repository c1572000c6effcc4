"""Development check: run-to-run differences of the graph trainer vs differences between exchange forms (one-rank RCCL group)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from holoscene_amd.training.synthetic import SyntheticScene  # noqa: E402
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf  # noqa: E402


def run(dp, exchange, steps):
    conf = stock_conf(num_rays=256, S=32, d_out=4, num_levels=16, end_size=512, logmap=15, beta=0.05, mlp_precision="bf16")
    tr = Stage1Trainer(conf, device=dev, optimizer="flat", graph=True, data_parallel=dp, exchange=exchange)
    benchmark_model_state(tr.model, 0.05)
    start = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
    scene = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, ring=4, device=dev)
    torch.manual_seed(77)
    for _ in range(steps):
        tr.train_step_resident(scene)
    torch.cuda.synchronize()
    return {n: (p.detach() - start[n]).cpu().numpy() for n, p in tr.model.named_parameters()}


for steps in (1, 3, 12):
    a, a2 = run(False, None, steps), run(False, None, steps)
    b, c = run(True, "overlap", steps), run(True, "serial", steps)
    for name in ("implicit_network.encoding.embeddings", "implicit_network.color_encoding.embeddings", "implicit_network.lin1.weight_v",
                 "rendering_network.lin1.weight_v"):
        w = a[name]
        row = [f"{float(np.mean(np.abs(x[name] - w))):.3e}" for x in (a2, b, c)]
        print(steps, name, "mean|upd|", f"{float(np.mean(np.abs(w))):.3e}", "rerun/overlap/serial mean err", row, flush=True)
dist.destroy_process_group()
