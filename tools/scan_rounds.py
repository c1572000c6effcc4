import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.training.synthetic import SyntheticScene
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
import io, contextlib
scene = SyntheticScene(1024, 32, device='cuda')
for beta in (0.3, 0.1, 0.03, 0.01, 0.003, 0.001):
    with contextlib.redirect_stdout(io.StringIO()):
        tr = Stage1Trainer(stock_conf(beta=beta, mlp_precision='bf16'), device='cuda', optimizer='flat')
    benchmark_model_state(tr.model, beta)
    rs = []
    for i in range(12):
        idx, mi, gt = scene.next_batch()
        tr.train_step(idx, mi, gt)
        rs.append(tr.model.ray_sampler.last_rounds)
    print(beta, rs, float(tr.model.density.get_beta()))
