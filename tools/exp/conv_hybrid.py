"""Which bf16 stage moves the trained Eikonal / normal balance?  

tests/test_convergence_gpu.py's two scenes, 300 iterations each, eager trainers: bf16 everywhere, fp32 everywhere, and bf16 with ONE stage
at a time (or a chosen set) in fp32 (HoloSceneNetwork.fp32_stages).  Prints the trailing means of the loss terms.
usage: python tools/exp/conv_hybrid.py [stage-set ...]     (a stage set = comma list of sampler,trunk,eikonal,colour; "" = pure bf16; "fp32" = fp32 run)
"""
import os
import sys
import torch
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import test_convergence_gpu as T  # noqa: E402
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state  # noqa: E402


def fit(stages, scene, seed0, graph=False):
    prec = "fp32" if stages == "fp32" else "bf16"
    tr = Stage1Trainer(T._conf(prec), device=T.DEV, optimizer="flat", graph=graph, seed=42)
    if prec == "bf16":
        tr.model.fp32_stages = frozenset(t for t in stages.split(",") if t)
    benchmark_model_state(tr.model, 0.05)
    hist = {"loss": [], "rgb_loss": [], "eikonal_loss": [], "depth_loss": [], "normal_l1": [], "normal_cos": [], "semantic_loss": []}
    for i in range(T.STEPS):
        torch.manual_seed(seed0 + i)
        _, lo = tr.train_step(*scene.next_batch())
        for k in hist:
            if k in lo:
                hist[k].append(lo[k].detach().clone())
    torch.cuda.synchronize()
    return {k: torch.stack(v).float().cpu() for k, v in hist.items() if v}


sets = [a for a in sys.argv[1:] if a != "-"] if len(sys.argv) > 1 else ["", "fp32", "sampler", "trunk", "eikonal", "colour"]
make = T._teacher_scene()
tail = lambda h, k: float(h[k][-T.TAIL:].mean())  # noqa: E731
for scene, seed0 in ((77, 9000), (31, 5000)):
    for st in sets:
        h = fit(st, make(scene), seed0)
        print(f"HYBRID scene {scene} fp32-stages [{st or 'none (bf16)'}]: " + " ".join(f"{k} {tail(h, k):.5f}" for k in h), flush=True)


# ---- second experiment: WHICH rounding inside the bf16 trunk of the rendered samples moves the result?  An all-fp32 model whose trunk call on
# the rendered samples (the call with n_main rows) rounds ONE thing to bf16 (straight-through: the backward pass sees the identity):
#   xraw / pe / feat: the raw coordinates / positional encodings / hash features of the VALUE row;  tan: the three tangent rows of the input;
#   w: the weight matrices (w0 / w1 / w2: one of them);  act: the activations handed from layer to layer (value and tangent rows)
class _LinFB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, round_fwd):
        wr = w.detach().to(torch.bfloat16).float()
        ctx.save_for_backward(x, w.detach() if round_fwd else wr)
        return x @ (wr if round_fwd else w.detach()).t()

    @staticmethod
    def backward(ctx, g):
        x, wb = ctx.saved_tensors
        return g @ wb, g.t() @ x, None


def emulate(rounds, scene, seed0):
    import numpy as np
    from holoscene_amd.model import network as N
    tr = Stage1Trainer(T._conf("bf16"), device=T.DEV, optimizer="flat", graph=False, seed=42)
    tr.model.fp32_stages = frozenset(("sampler", "trunk", "eikonal", "colour"))
    benchmark_model_state(tr.model, 0.05)
    net = tr.model.implicit_network
    orig = net.sdf_and_jacobian
    n_main = 256 * (32 // 2 + 32 // 4 + 2)
    ste = lambda t: t + (t.detach().to(torch.bfloat16).float() - t.detach())  # noqa: E731

    def patched(x):
        if x.shape[0] != n_main or net.mlp_bf16:
            return orig(x)
        x = x.detach()
        enc = net.encoding
        inp = N._trunk_input.apply(x, enc.embeddings, enc.offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                                   net.embedder.multires, float(net.divide_factor), torch.float32)       # [B, 4, 71]
        v, t = inp[:, :1], inp[:, 1:]
        cols = []
        for name, sl in (("xraw", slice(0, 3)), ("pe", slice(3, 39)), ("feat", slice(39, 71))):
            cols.append(ste(v[:, :, sl]) if name in rounds else v[:, :, sl])
        v = torch.cat(cols, 2)
        if "tan" in rounds:
            t = ste(t)
        h = torch.cat([v, t], 1)
        lins = net._lins()
        for l, lin in enumerate(lins):
            w = ste(lin.weight) if ("w" in rounds or f"w{l}" in rounds) else lin.weight
            if f"w{l}f" in rounds or f"w{l}b" in rounds:      # rounded weights in the forward product only / in the input-gradient product only
                out = _LinFB.apply(h.reshape(-1, h.shape[-1]), lin.weight, f"w{l}f" in rounds).view(*h.shape[:-1], -1)
            else:
                out = N.linear_rows(h, w, None, False)
            if l < len(lins) - 1:
                h = N.softplus_tangent(out, lin.bias)
                if "act" in rounds:
                    h = ste(h)
            else:
                return N._split_value_jacobian.apply(out, lin.bias)
    net.sdf_and_jacobian = patched
    hist = {"loss": [], "eikonal_loss": [], "normal_l1": [], "rgb_loss": []}
    for i in range(T.STEPS):
        torch.manual_seed(seed0 + i)
        _, lo = tr.train_step(*scene.next_batch())
        for k in hist:
            hist[k].append(lo[k].detach().clone())
    torch.cuda.synchronize()
    return {k: torch.stack(v).float().cpu() for k, v in hist.items()}


if os.environ.get("HS_EMULATE"):
    for rounds in os.environ["HS_EMULATE"].split(";"):
        h = emulate(frozenset(t for t in rounds.split(",") if t), make(77), 9000)
        print(f"EMULATE scene 77 fp32 model, rendered-sample trunk rounds [{rounds or 'nothing'}] to bf16: " + " ".join(f"{k} {tail(h, k):.5f}" for k in h), flush=True)
