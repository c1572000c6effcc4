import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.training.synthetic import SyntheticScene
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
tr = Stage1Trainer(stock_conf(beta=0.1, mlp_precision=prec), device='cuda', optimizer='flat', graph=True)
benchmark_model_state(tr.model, 0.1)
scene = SyntheticScene(1024, 32, num_frames=8, ring=64, device='cuda')
t0 = time.time()
for it in range(n):
    out, lo = tr.train_step_resident(scene)
    if it % 100 == 0 or it == n - 1:
        torch.cuda.synchronize()
        st = tr.flat.read_state()
        print(f"it {it:4d} loss {float(lo['loss']):9.4f} rgb {float(lo['rgb_loss']):.4f} eik {float(lo['eikonal_loss']):.4f} sem {float(lo['semantic_loss']):.4f} "
              f"beta {float(tr.model.density.get_beta()):.5f} rounds {tr.model.ray_sampler.last_rounds} lr_grid {st.lr[0]:.6f} "
              f"finite {bool(torch.isfinite(tr.flat.flat_p).all())} {time.time() - t0:.1f}s")
