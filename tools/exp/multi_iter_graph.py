#!/usr/bin/env python3
"""tools/exp/multi_iter_graph.py -- what would U iterations per graph launch buy?  The regular iteration's body (Stage1Trainer._full_body) captured U times
into ONE graph (same static batch for all of them: a timing probe, not a training loop -- the batch draw stays outside) and replayed; ms per iteration for
U = 1, 2, 5.  The difference to U = 1 is the share of the ~14 us between two graph launches that amortises."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from holoscene_amd.training.synthetic import SyntheticScene
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf

dev = torch.device("cuda", 0)
res = {}
for U in (1, 2, 5, 1, 2, 5):
    conf = stock_conf(num_rays=1024, S=128, d_out=32, beta=0.001, mlp_precision="bf16", learning_rate=5.0e-10)
    tr = Stage1Trainer(conf, device=dev, seed=42, optimizer="flat", graph=True)
    benchmark_model_state(tr.model, 0.001)
    scene = SyntheticScene(1024, 32, seed=1234, device=dev)
    for _ in range(12):      # past iteration 0 (background patch) and the captures of both variants
        tr.train_step_resident(scene)
    tr.iter_step = 1         # a regular iteration
    key = ("full", False, False)
    st = tr._graphs[key]["static"]
    tr.model.train()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(U):
            out, lo = tr._full_body(st, False, False)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    n = 200 // U
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (n * U) * 1e3
    res.setdefault(U, []).append(ms)
    print(f"U = {U}: {ms:.4f} ms per iteration (no batch draw)", flush=True)
    del g, tr, scene
    torch.cuda.empty_cache()
print({u: [round(v, 4) for v in vs] for u, vs in res.items()})
