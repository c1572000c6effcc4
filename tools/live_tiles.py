"""Which 32-sample tiles of the rendered samples reach the backward kernels with a NON-zero cotangent?

Decides what a tile-level skip in the backward kernels (appearance backward, rr backward pair, weight gradients, table scatters) can save:
compositing weights vanish exactly behind the first surface (transmittance underflows) and, at small beta, far in front of it (the
Laplace density underflows), so whole runs of consecutive samples of a ray carry exactly-zero cotangents.

    python tools/live_tiles.py [beta] [lr_scale] [steps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from holoscene_amd.training.synthetic import SyntheticScene  # noqa: E402
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf  # noqa: E402
from holoscene_amd.hashencoder import backend as be  # noqa: E402

beta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
lr_scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
tr = Stage1Trainer(stock_conf(beta=beta, mlp_precision="bf16", learning_rate=5.0e-4 * lr_scale), device="cuda", optimizer="flat", graph=False)
benchmark_model_state(tr.model, beta)
scene = SyntheticScene(1024, 32, num_frames=8, ring=64, device="cuda")
B_ = be._backend
log = {}


def tiles(mask_rows):
    n = mask_rows.shape[0]
    pad = (-n) % 32
    if pad:
        mask_rows = torch.cat([mask_rows, torch.zeros(pad, dtype=torch.bool, device=mask_rows.device)])
    return mask_rows.view(-1, 32).any(1)


o_app, o_gy, o_bg = B_.appearance2_bwd, B_.trunk_rr_gy, B_.trunk_rr_bwd_grad


def app(g_rgb, *a, **k):
    live = (g_rgb != 0).any(1)
    log["appearance"] = (float(live.float().mean()), float(tiles(live).float().mean()))
    log["_app_rows"] = live
    return o_app(g_rgb, *a, **k)


def gy(g_raw, g_sdf, idx, K, out, part, *a, **k):
    live = (g_raw != 0).any(1) if g_raw is not None else None
    if g_sdf is not None:
        l2 = g_sdf.view(-1) != 0
        live = l2 if live is None else (live | l2)
    log["_gy_rows"] = live
    return o_gy(g_raw, g_sdf, idx, K, out, part, *a, **k)


def bg(x, dydx, gg, *a, **k):
    live = (gg != 0).any(1)
    rows = log.get("_gy_rows")
    both = live if rows is None else (live | rows)
    log["trunk value cot"] = (float(rows.float().mean()), float(tiles(rows).float().mean())) if rows is not None else None
    log["trunk grad cot"] = (float(live.float().mean()), float(tiles(live).float().mean()))
    log["trunk either"] = (float(both.float().mean()), float(tiles(both).float().mean()))
    return o_bg(x, dydx, gg, *a, **k)


B_.appearance2_bwd, B_.trunk_rr_gy, B_.trunk_rr_bwd_grad = staticmethod(app), staticmethod(gy), staticmethod(bg)
for it in range(steps):
    idx, mi, gt = scene.next_batch()
    log.clear()
    out, lo = tr.train_step(idx, mi, gt)
    if it in (0, 1, steps // 2, steps - 1):
        # (fraction of live samples, fraction of live 32-sample tiles)
        print("iter", it, "loss %.4f" % float(lo["loss"]), {k: tuple(round(x, 4) for x in v) for k, v in log.items() if not k.startswith("_") and v})
