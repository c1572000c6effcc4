"""The convergence scenario of tests/test_convergence_gpu.py with d_out = 40 (the K = 33..64 path: 128-point tile kernels): does bf16 end where fp32
does -- with W2 as two planes (default) and as one (HOLOSCENE_W2_PLANES=1)?   usage: python tools/exp/conv_k40.py bf16|fp32"""
import sys
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_convergence_gpu as T  # noqa: E402
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf  # noqa: E402

prec = sys.argv[1]
make = T._teacher_scene()
tail = lambda h, k: float(torch.stack(h[k][-T.TAIL:]).float().mean())  # noqa: E731
for scene, seed0 in ((77, 9000), (31, 5000)):
    conf = stock_conf(num_rays=256, S=32, d_out=40, num_levels=16, end_size=512, logmap=15, beta=0.05, mlp_precision=prec, use_bg_reg=True, learning_rate=1.0e-4)
    tr = Stage1Trainer(conf, device="cuda", optimizer="flat", graph=(prec == "bf16"), seed=42)
    benchmark_model_state(tr.model, 0.05)
    sc = make(scene)
    h = {"loss": [], "eikonal_loss": [], "normal_l1": [], "rgb_loss": []}
    for i in range(T.STEPS):
        torch.manual_seed(seed0 + i)
        _, lo = tr.train_step(*sc.next_batch())
        for k in h:
            h[k].append(lo[k].detach().clone())
    torch.cuda.synchronize()
    print(f"K40 scene {scene} {prec}: " + " ".join(f"{k} {tail(h, k):.5f}" for k in h), flush=True)
