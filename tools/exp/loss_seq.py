"""Loss sequence of tests/test_model_gpu.py::_full_graph_trainer's scenario (random synthetic scene, stock learning rates) in fp32 and bf16."""
import sys
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from holoscene_amd.training.synthetic import SyntheticScene
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf

for prec, graph in (("fp32", False), ("bf16", True), ("bf16", False)):
    torch.manual_seed(0)
    tr = Stage1Trainer(stock_conf(num_rays=256, S=32, d_out=4, num_levels=16, end_size=512, logmap=15, beta=0.05, mlp_precision=prec), device="cuda",
                       optimizer="flat", graph=graph)
    benchmark_model_state(tr.model, 0.05)
    scene = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, ring=4, device="cuda")
    torch.cuda.manual_seed(7)
    tr.model.rng_state("cuda")[1] = 7
    ls = []
    for it in range(60):
        _, lo = tr.train_step(*scene.next_batch())
        ls.append(float(lo["loss"]))
    print(prec, graph, " ".join(f"{v:.2f}" for v in ls))
