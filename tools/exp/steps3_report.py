import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, numpy as np
from helpers import load, section
from model_helpers import build_model, run_three_steps
rec = load("steps3_k3")
for flat in (False, True):
    model = build_model(rec, "cuda").train()
    before = {k: p.detach().cpu().clone() for k, p in model.named_parameters()}
    losses, lrs, snaps = run_three_steps(model, rec, "cuda", flat=flat)
    rep = {}
    for n in (1, 3):
        for k, v in section(rec, f"adam{n}.").items():
            d_ref, d_got = v - before[k], snaps[n][k] - before[k]
            if float(d_ref.norm()) > 0: rep[(n, k)] = float((d_got - d_ref).norm() / d_ref.norm())
    print(flat, [round(abs(l - float(rec[f"s{i}.loss"])) / float(rec[f"s{i}.loss"]), 6) for i, l in enumerate(losses)], sorted(rep.items(), key=lambda kv: -kv[1])[:4])
