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
def jac(g_feat, g_dydx, x01, offsets, target, B, D, C, L, S, H):
    zf = (g_feat == 0).all(1) if g_feat is not None else torch.ones(B, dtype=torch.bool, device=x01.device)
    zj = (g_dydx == 0).all(2).all(0) if g_dydx is not None else torch.ones(B, dtype=torch.bool, device=x01.device)
    lvl = ((g_feat.view(B, L, C) == 0).all(2) & (g_dydx.view(L, B, -1) == 0).all(2).t()).float().mean().item() if g_feat is not None and g_dydx is not None else -1
    log.append(("jac", B, float((zf & zj).float().mean()), lvl))
    return orig_jac(g_feat, g_dydx, x01, offsets, target, B, D, C, L, S, H)
def bwd(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs):
    if grad_embeddings is not None:
        log.append(("bwd", B, float((grad == 0).all(1).float().mean()), float((grad.view(B, L, C) == 0).all(2).float().mean())))
    return orig_bwd(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs)
B_.bwd_jac, B_.bwd = staticmethod(jac), staticmethod(bwd)
for it in range(40):
    idx, mi, gt = scene.next_batch()
    log.clear()
    out, lo = tr.train_step(idx, mi, gt)
    if it in (0, 5, 39):
        print("iter", it, "loss", float(lo["loss"]), [(k, b, round(z, 4), round(l, 4)) for k, b, z, l in log])
