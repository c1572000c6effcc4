"""How many points reach the hash scatter with an exactly-zero cotangent?  (decides whether a zero-skip in the scatter pays)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.training.synthetic import SyntheticScene
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
from holoscene_amd.hashencoder import backend as be
beta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
tr = Stage1Trainer(stock_conf(beta=beta, mlp_precision='bf16'), device='cuda', optimizer='flat', graph=False)
benchmark_model_state(tr.model, beta)
scene = SyntheticScene(1024, 32, num_frames=8, ring=64, device='cuda')
B_ = be._backend
orig_jac, orig_bwd = B_.bwd_jac, B_.bwd
log = []
def rows(t, B, L, per, level_major):
    """[B, L, per] view of a cotangent image stored point-major [B, L*per] or level-major [L, B, per]."""
    return t.view(L, B, per).transpose(0, 1) if level_major else t.view(B, L, per)
def jac(g_feat, g_dydx, x01, offsets, target, B, D, C, L, S, H, **kw):
    lm = bool(kw.get("level_major"))
    zf = (rows(g_feat, B, L, C, lm) == 0).all(2) if g_feat is not None else torch.ones(B, L, dtype=torch.bool, device=x01.device)
    zj = (g_dydx.view(L, B, -1) == 0).all(2).t() if g_dydx is not None else torch.ones(B, L, dtype=torch.bool, device=x01.device)
    both = zf & zj
    log.append(("jac", B, float(both.all(1).float().mean()), float(both.float().mean())))
    return orig_jac(g_feat, g_dydx, x01, offsets, target, B, D, C, L, S, H, **kw)
def bwd(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, **kw):
    if grad_embeddings is not None:
        z = rows(grad, B, L, C, bool(kw.get("level_major"))) == 0
        log.append(("bwd", B, float(z.all(2).all(1).float().mean()), float(z.all(2).float().mean())))
    return orig_bwd(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, **kw)
B_.bwd_jac, B_.bwd = staticmethod(jac), staticmethod(bwd)
for it in range(40):
    idx, mi, gt = scene.next_batch()
    log.clear()
    out, lo = tr.train_step(idx, mi, gt)
    if it in (0, 5, 39):
        # (kernel, points, fraction of points whose whole cotangent is exactly zero, fraction of (point, level) cells that are)
        print("iter", it, "loss", float(lo["loss"]), [(k, b, round(z, 4), round(l, 4)) for k, b, z, l in log])
