"""Host-side time of each phase of the graph-replayed training step (no extra syncs) + wall time per step."""
import os, sys, time, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.training.synthetic import SyntheticScene
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
import io, contextlib
beta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
with contextlib.redirect_stdout(io.StringIO()):
    tr = Stage1Trainer(stock_conf(beta=beta, mlp_precision='bf16', learning_rate=5e-10), device='cuda', optimizer='flat', graph=True)
benchmark_model_state(tr.model, beta)
scene = SyntheticScene(1024, 32, device='cuda')
for _ in range(25):
    idx, mi, gt = scene.next_batch(); tr.train_step(idx, mi, gt)
torch.cuda.synchronize()
acc = collections.Counter()
model = tr.model
N = 50
T0 = time.perf_counter()
for it in range(N):
    t = [time.perf_counter()]
    idx, mi, gt = scene.next_batch(); t.append(time.perf_counter())
    with torch.no_grad():
        rays = model.prepare_rays(mi); t.append(time.perf_counter())
        z_vals, z_eik = model.sample(rays); t.append(time.perf_counter())
        with_bg = model.wants_background(tr.iter_step)
        bg = model.prepare_background(mi) if with_bg else None; t.append(time.perf_counter())
    key = (with_bg, False)
    entry = tr._graphs[key]
    st = entry["static"]
    tr._copy_into(st["rays"], rays); st["z_vals"].copy_(z_vals); st["z_eik"].copy_(z_eik); tr._copy_into(st["gt"], gt)
    if with_bg: tr._copy_into(st["bg"], bg)
    t.append(time.perf_counter())
    entry["graph"].replay(); t.append(time.perf_counter())
    tr.iter_step += 1
    for n, a, b in zip(("next_batch", "prepare_rays", "sample", "background", "copies", "replay"), t[:-1], t[1:]):
        acc[n] += (b - a) * 1e3
torch.cuda.synchronize()
wall = (time.perf_counter() - T0) * 1e3 / N
print(f"rounds {model.ray_sampler.last_rounds} wall {wall:.3f} ms/step; host ms/step:", {k: round(v / N, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()) / N, 3))
