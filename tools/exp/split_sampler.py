"""Feasibility: the sampler's chain (gather -> SDF sweep -> update, five rounds) on two halves of the rays, forked onto two streams inside
one captured graph, against the single chain.  python tools/exp/split_sampler.py [parts]"""
import sys
import torch

sys.path.insert(0, ".")
from holoscene_amd.model import network as _net                      # noqa: E402
from holoscene_amd.training.synthetic import SyntheticScene          # noqa: E402
from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf  # noqa: E402

DEV = "cuda"
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tr = Stage1Trainer(stock_conf(mlp_precision="bf16", beta=0.02), device=DEV, optimizer="flat", graph=True)
benchmark_model_state(tr.model, 0.02)
model = tr.model.train()
scene = SyntheticScene(1024, 32, device=DEV)
_, inp, gt = scene.next_batch()
R = inp["uv"].shape[1]
sizes = model.uniform_sizes(R)
streams = [torch.cuda.Stream() for _ in range(parts)]


def cut(v, a, b):
    if torch.is_tensor(v):
        for d, n in enumerate(v.shape):
            if n == R:
                return v.narrow(d, a, b - a).contiguous()
        return v
    if isinstance(v, dict):
        return {k: cut(x, a, b) for k, x in v.items()}
    if isinstance(v, (tuple, list)):
        return type(v)(cut(x, a, b) for x in v)
    return v


def body(split):
    with _net.iteration_prologue(model, None, sizes) as drawn:
        with torch.no_grad():
            rays = model.prepare_rays(inp, drawn)
            if not split:
                return model.sample(dict(rays), drawn)
            cur = torch.cuda.current_stream()
            outs = []
            for i, s in enumerate(streams):
                a, b = i * R // parts, (i + 1) * R // parts
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs.append(model.sample(cut(dict(rays), a, b), cut(drawn, a, b)))
            for s in streams:
                cur.wait_stream(s)
            return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])


res = {}
for split in (False, True):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            model.rng_state(DEV)[1] = 7
            out = body(split)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    _net._be._backend.scatter_workspaces_idle()
    g = torch.cuda.CUDAGraph()
    model.rng_state(DEV)[1] = 7
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = body(split)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 40 * 1000)
    model.rng_state(DEV)[1] = 7
    g.replay()
    torch.cuda.synchronize()
    res[split] = (sorted(ts)[2], out[0].clone(), out[1].clone(), int(model.ray_sampler._rounds) if not torch.is_tensor(model.ray_sampler._rounds) else int(model.ray_sampler._rounds.item()))
    print(f"split={split} parts={parts}: prologue + rays + sampler {res[split][0]:.1f} us per replay, rounds {res[split][3]}")
z0, z1 = res[False][1], res[True][1]
print("z_vals equal:", bool(torch.equal(z0, z1)), "max |dz|", float((z0 - z1).abs().max()), "z_eik max |d|", float((res[False][2] - res[True][2]).abs().max()))
