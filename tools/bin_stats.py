"""Record counts per (level, bin) of the binned scatter in the benchmark state."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.training.synthetic import SyntheticScene
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
from holoscene_amd.hashencoder import backend as be
import io, contextlib
beta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
with contextlib.redirect_stdout(io.StringIO()):
    tr = Stage1Trainer(stock_conf(beta=beta, mlp_precision='bf16', learning_rate=5e-10), device='cuda', optimizer='flat', graph=False)
benchmark_model_state(tr.model, beta)
scene = SyntheticScene(1024, 32, device='cuda')
B_ = be._HipBackend
orig = B_.__dict__['bwd_jac'].__func__
seen = []
def jac(cls, g_feat, g_dydx, inputs, offsets, ge, B, D, C, L, S, H, ws=None, level_major=False):
    r = orig(cls, g_feat, g_dydx, inputs, offsets, ge, B, D, C, L, S, H, ws=ws, level_major=level_major)
    if ws is not None:
        torch.cuda.synchronize()
        cnt = ws[0][:32 * 128 * 4].view(torch.int32).view(32, 128)[:L].clone()
        nz = float(((g_feat != 0).any(-1).any(0) if level_major else (g_feat != 0).any(1)).float().mean())
        seen.append((B, ws[1], int(cnt.sum()), int(cnt.max()), [int(v) for v in cnt.sum(1)], nz))
    return r
setattr(B_, 'bwd_jac', classmethod(jac))
for it in range(4):
    idx, mi, gt = scene.next_batch()
    seen.clear()
    tr.train_step(idx, mi, gt)
for s in seen:
    print("B", s[0], "cap", s[1], "records", s[2], "max/bin", s[3], "nonzero points", round(s[5], 3), "per level", s[4])
