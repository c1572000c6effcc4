"""GPU: the Stage-1 path through libholoscene_hip.so against the reference-generated golden fixtures
and the CPU oracle (same checks as tests/test_model_cpu.py, plus full-size properties)."""
import numpy as np
import pytest
import torch

from helpers import load, rand_dict, section
from model_helpers import build_model, build_loss, close, z_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dev(d):
    return {k: (v.to(DEV) if torch.is_tensor(v) else ({kk: vv.to(DEV) for kk, vv in v.items()} if isinstance(v, dict) else v))
            for k, v in d.items()}


@pytest.mark.parametrize("impl", ["hip", "torch"])
@pytest.mark.parametrize("name", [f"sampler_{i}" for i in range(5)] + ["sampler_eval"])
def test_sampler(name, impl, monkeypatch):
    from holoscene_amd.model import ray_sampler
    monkeypatch.setattr(ray_sampler, "SAMPLER_IMPL", impl)
    rec = load(name)
    model = build_model(rec, DEV)
    model.train(bool(rec["meta.train"]))
    ins = _dev(section(rec, "in."))
    z, z_eik = model.ray_sampler.get_z_vals(ins["ray_dirs"], ins["cam_loc"], model, rng=_dev(rand_dict(rec)))
    assert model.ray_sampler.last_rounds == int(rec["meta.rounds"])
    z_close(z, torch.from_numpy(rec["out.z_vals"]), frac_loose=0.05)
    assert z_eik.shape == rec["out.z_samples_eik"].shape
    eik_idx = torch.from_numpy(rec["rand.eik_idx"]).to(DEV)
    assert torch.equal(z_eik, torch.gather(z, 1, eik_idx[:, None]))


@pytest.mark.parametrize("impl,control", [("hip", "device"), ("hip", "host"), ("torch", "host")])
def test_sampler_inverse_sphere_branch(impl, control, monkeypatch):
    """inverse_sphere_bg=True (ray_sampler.py:127-128, 262-265, 282-285) through the fused kernels -- the per-ray far bound reaches the
    tail / final kernels as `far_rays` -- against the reference's own call on its draws (tests/golden/sampler_inv.npz)."""
    from holoscene_amd.model import ray_sampler
    monkeypatch.setattr(ray_sampler, "SAMPLER_IMPL", impl)
    monkeypatch.setattr(ray_sampler, "CONTROL", control)
    rec = load("sampler_inv")
    model = build_model(rec, DEV).train()
    ins = _dev(section(rec, "in."))
    (z, z_inv), z_eik = model.ray_sampler.get_z_vals(ins["ray_dirs"], ins["cam_loc"], model, rng=_dev(rand_dict(rec)))
    assert model.ray_sampler.last_rounds == int(rec["meta.rounds"])
    z_close(z, torch.from_numpy(rec["out.z_vals"]), frac_loose=0.05)
    close(z_inv, rec["out.z_vals_inverse_sphere"], 1e-6, 1e-7, "z_vals_inverse_sphere")
    assert torch.equal(z_eik, torch.gather(z, 1, torch.from_numpy(rec["rand.eik_idx"]).to(DEV)[:, None]))
    from holoscene_amd.utils.rend_util import get_sphere_intersections
    far = get_sphere_intersections(ins["cam_loc"], ins["ray_dirs"], r=1.0)[:, 1:]
    assert bool(((z - far).abs().min(dim=1)[0] == 0).all()) and float(z.max()) < 2.0


@pytest.mark.parametrize("name", ["iter_k3_bg", "iter_k5"])
def test_iteration(name):
    rec = load(name)
    model = build_model(rec, DEV).train()
    ins, gt = _dev(section(rec, "in.")), _dev(section(rec, "gt."))
    out = model(ins, torch.tensor([0]), iter_step=int(rec["meta.iter_step"]), rng=_dev(rand_dict(rec)))
    ref = section(rec, "out.")
    z_close(out["z_vals"], ref["z_vals"], frac_loose=0.05)
    same = ((out["z_vals"].cpu() - ref["z_vals"]).abs() < 1e-4).all(dim=1)  # rays whose samples did not slide
    assert same.float().mean() > 0.8
    for k in ("rgb_values", "depth_values", "normal_map", "object_opacity", "semantic_values"):
        close(out[k].cpu()[same], ref[k][same], 2e-3, 5e-4, k)
    for k in ("sample_sdf", "sample_minsdf"):
        if k in ref:
            close(out[k], ref[k], 1e-3, 1e-4, k)
    out["iter_step"] = int(rec["meta.iter_step"])
    lo = build_loss()(out, gt, call_reg=bool(rec["meta.call_reg"]))
    for k, v in section(rec, "loss.").items():
        close(lo[k], v, 5e-3, 1e-4, "loss." + k)
    lo["loss"].backward()
    params = dict(model.named_parameters())
    for k, v in section(rec, "grad.").items():
        g = params[k].grad
        assert g is not None and torch.isfinite(g).all(), k
        rel = float((g.cpu() - v).norm() / (v.norm() + 1e-12))
        assert rel < 2e-2, (k, rel)


@pytest.mark.parametrize("path", ["bf16_graph", "fp32_torch"])
def test_full_size_iteration_properties(path):
    """BASELINE config 2 (1024 rays x 128 samples, K=32, 16-level grid): structural invariants -- on the BENCHMARKED path (bf16 MLP
    operands, flat buffers, the whole-iteration HIP graph: what bench.py replays) and on the fp32 / torch.optim path."""
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    if path == "bf16_graph":
        tr = Stage1Trainer(stock_conf(beta=0.001, mlp_precision="bf16"), device=DEV, optimizer="flat", graph=True)
    else:
        tr = Stage1Trainer(stock_conf(beta=0.001), device=DEV, optimizer="torch")
    benchmark_model_state(tr.model, 0.001)
    scene = SyntheticScene(1024, 32, device=DEV)
    idx, mi, gt = scene.next_batch()
    before = tr.model.implicit_network.encoding.embeddings.detach().clone()
    out, lo = tr.train_step(idx, mi, gt)
    torch.cuda.synchronize()
    if path == "bf16_graph":
        assert ("full", True, False) in tr._graphs, "the step must have gone through the whole-iteration graph"
    z = out["z_vals"]
    assert z.shape == (1024, 98)
    assert bool((z[:, 1:] >= z[:, :-1]).all()), "depths must be sorted"
    assert float(z.min()) >= 0.0 and float(z.max()) <= 3.5 + 1e-6
    w = out["weights"]
    assert bool((w >= -1e-6).all()) and bool((w.sum(-1) <= 1 + (1e-3 if path == "bf16_graph" else 1e-4)).all()), "compositing weights form a sub-probability"
    assert out["grad_theta"].shape[0] == (32 + 1) * 4 * 1024 // 2
    assert 1 <= int(tr.model.ray_sampler.last_rounds) <= 5
    assert torch.isfinite(lo["loss"]) and "bg_depth_values" in out  # iteration 0 renders the background patch
    after = tr.model.implicit_network.encoding.embeddings.detach()
    assert float((after - before).abs().max()) > 0, "Adam must have moved the geometry grid"


@pytest.mark.parametrize("rows,W,B", [(4, 256, 1000), (4, 64, 333), (1, 256, 65), (3, 128, 64)])
def test_softplus_tangent_kernels(rows, W, B):
    """Fused trunk stage vs plain torch ops (fp32 reference of the same op), forward and backward."""
    from holoscene_amd.model.network import softplus_tangent
    g = torch.Generator().manual_seed(rows * 100 + W)
    A = (torch.randn(B, rows, W, generator=g) * 0.05).to(DEV).requires_grad_(True)
    A.data[0, 0, :8] = torch.tensor([0.3, -0.3, 0.2001, 0.1999, 0.0, 1e-4, -1e-4, 5.0])  # both softplus branches
    bias = (torch.randn(W, generator=g) * 0.01).to(DEV).requires_grad_(True)
    G = torch.randn(B, rows, W, generator=g).to(DEV)
    out = softplus_tangent(A, bias)
    v = A[:, 0] + bias
    ref = torch.cat([torch.nn.functional.softplus(v, beta=100).unsqueeze(1), torch.sigmoid(100 * v).unsqueeze(1) * A[:, 1:]], 1)
    close(out, ref, 1e-5, 1e-6, "fwd")
    gA, gb = torch.autograd.grad(out, (A, bias), G)
    rA, rb = torch.autograd.grad(ref, (A, bias), G)
    close(gA, rA, 1e-4, 1e-5, "gA")
    close(gb, rb, 1e-4, 1e-4 * float(rb.abs().max()), "gbias")


def test_bf16_mlp_mode_within_stated_tolerance():
    """bf16 matrix-core mode vs the fp32 parity mode on identical parameters, rays, draws AND sample depths
    (sample placement is an ill-conditioned function of the SDF, so the fp32 depths are injected into the bf16
    run; the bf16 sampler itself is checked for well-formedness).
    Stated tolerances: SDF atol 5e-3, rendered RGB/depth atol 2e-2, Eikonal loss rtol 5e-2."""
    rec = load("iter_k5")
    ins, gt = _dev(section(rec, "in.")), _dev(section(rec, "gt."))
    outs, losses, grads = {}, {}, {}
    fixed = {}
    for prec in ("fp32", "bf16"):
        model = build_model(rec, DEV).train()
        model.implicit_network.set_mlp_precision(prec)
        model.rendering_network.set_mlp_precision(prec)
        if prec == "bf16":
            own_z, _ = model.ray_sampler.get_z_vals(fixed["dirs"], fixed["loc"], model, rng=_dev(rand_dict(rec)))
            assert bool((own_z[:, 1:] >= own_z[:, :-1]).all()) and float(own_z.min()) >= 0 and float(own_z.max()) <= 3.5 + 1e-6
            model.ray_sampler.get_z_vals = lambda *a, **k: fixed["z"]
        else:
            orig = model.ray_sampler.get_z_vals

            def capture(d, o, m, **k):
                z = orig(d, o, m, **k)
                fixed.update(z=z, dirs=d, loc=o)
                return z
            model.ray_sampler.get_z_vals = capture
        out = model(ins, torch.tensor([0]), iter_step=3, rng=_dev(rand_dict(rec)))
        out["iter_step"] = 3
        lo = build_loss()(out, gt)
        lo["loss"].backward()
        outs[prec], losses[prec] = out, lo
        grads[prec] = {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
    a, b = outs["fp32"], outs["bf16"]
    assert torch.equal(a["z_vals"], b["z_vals"])
    report = {k: float((a[k] - b[k]).abs().max()) for k in ("sdf", "rgb_values", "depth_values", "normal_map", "object_opacity")}
    report["eik_rel"] = abs(float(losses["bf16"]["eikonal_loss"]) / float(losses["fp32"]["eikonal_loss"]) - 1)
    report["loss_rel"] = abs(float(losses["bf16"]["loss"]) / float(losses["fp32"]["loss"]) - 1)
    report["sdf_scale"] = float(a["sdf"].abs().max())
    print("bf16-vs-fp32:", report)
    # measured on MI355X (this fixture): sdf 1.4e-2 on |sdf| <= 2.74 (= bf16 epsilon, relative 5e-3), rgb 2.7e-4, depth 1.4e-3,
    # normal 4.7e-3, opacity 1.8e-3, Eikonal loss 2e-4 relative.  Bounds below leave ~4x head room.
    assert report["sdf"] < 2e-2 * max(1.0, report["sdf_scale"])
    assert report["rgb_values"] < 2e-3 and report["depth_values"] < 1e-2 and report["normal_map"] < 2e-2 and report["object_opacity"] < 1e-2
    assert report["eik_rel"] < 5e-3
    for k in ("rgb_loss", "depth_loss", "normal_l1", "normal_cos", "smooth_loss"):
        va, vb = float(losses["fp32"][k]), float(losses["bf16"][k])
        assert abs(va - vb) <= 2e-2 * abs(va) + 1e-4, (k, va, vb)
    # the opacity BCE clips probabilities at 1e-4 (loss.py:489): where a ray's opacity for its target object is ~0 the
    # loss is log of a number of the size of the bf16 rounding noise, so only its order of magnitude is comparable
    assert abs(float(losses["bf16"]["semantic_loss"]) / float(losses["fp32"]["semantic_loss"]) - 1) < 0.3
    for k in ("implicit_network.lin1.weight_v", "rendering_network.lin1.weight_v", "implicit_network.encoding.embeddings"):
        ga, gb = grads["fp32"][k], grads["bf16"][k]
        cos = float((ga * gb).sum() / (ga.norm() * gb.norm() + 1e-20))
        print("grad cosine", k, cos)
        assert cos > 0.9, (k, cos)


def test_fused_flat_adam_matches_torch_adam():
    """csrc/optim.hip vs torch.optim.Adam + ExponentialLR with the reference's groups, 3 steps, ragged sizes."""
    from holoscene_amd.training.flat import FlatAdam
    from holoscene_amd.training.optim import build_optimizer, build_scheduler
    rec = load("iter_k5")
    ref_model, model = build_model(rec, DEV), build_model(rec, DEV)
    opt = build_optimizer(ref_model, lr=5e-4, lr_factor_for_grid=20.0)
    sched = build_scheduler(opt, 0.1, 1000)
    flat = FlatAdam(model, 5e-4, 20.0, 0.1, 1000)
    names = [n for n, _ in model.named_parameters()]
    pm, pr = dict(model.named_parameters()), dict(ref_model.named_parameters())
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        flat.zero_grad()
        for n in names:
            grad = (torch.randn(pm[n].shape, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g)))).to(DEV)
            if pm[n].grad is None:      # small tensors are detached by zero_grad and handed over like autograd does
                pm[n].grad = grad.clone()
            else:                       # hash tables keep their flat view attached
                pm[n].grad.copy_(grad)
            pr[n].grad = grad.clone()
        flat.gather_grads()
        flat.step()
        opt.step()
        sched.step()
        for n in names:
            close(pm[n], pr[n], 2e-6, 3e-8, f"step{step}.{n}")  # atol ~ fp32 round-off of a 1e-2 update
    st = flat.read_state()
    assert st.step == 3
    assert abs(st.lr[0] - opt.param_groups[0]["lr"] / (0.1 ** (1 / 1000))) < 1e-9  # state holds the lr used by the last step
    # parameters are still the module's tensors (views of the flat buffer), state-dict names unchanged
    assert sorted(model.state_dict().keys()) == sorted(ref_model.state_dict().keys())


def test_graph_replay_matches_eager_execution(monkeypatch):
    """The captured HIP graph (render + loss + backward) must reproduce the eager execution of the same body on the
    same static inputs and the same generator state: outputs equal, gradients equal up to float-atomic ordering."""
    from holoscene_amd.model import ray_sampler
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    monkeypatch.setattr(ray_sampler, "CONTROL", "host")     # the render-only graph behind a host-controlled sampler (with device control this
                                                            # conf takes the whole-iteration graph: the fp32 sweeps accept its 8-level grid)
    tr = Stage1Trainer(stock_conf(num_rays=256, S=32, d_out=4, num_levels=8, end_size=256, logmap=14, beta=0.05), device=DEV,
                       optimizer="flat", graph=True, freeze_parameters=True)
    benchmark_model_state(tr.model, 0.05)
    scene = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, ring=4, device=DEV)
    for key_iter in (0, 3):  # iteration 0 renders the background patch, iteration 3 does not
        tr.iter_step = key_iter
        idx, mi, gt = scene.next_batch()
        tr.train_step(idx, mi, gt)
        entry = tr._graphs[(key_iter == 0, False)]
        torch.cuda.manual_seed(7)
        entry["graph"].replay()
        torch.cuda.synchronize()
        g_graph = tr.flat.flat_g.clone()
        out_graph = {k: v.clone() for k, v in entry["out"].items() if torch.is_tensor(v)}
        loss_graph = float(entry["loss"]["loss"])
        torch.cuda.manual_seed(7)
        out_eager, loss_eager = tr._graph_body(entry["static"], key_iter == 0, False)
        g_eager = tr.flat.flat_g.clone()
        assert abs(loss_graph - float(loss_eager["loss"])) <= 1e-5 * abs(loss_graph)
        for k in ("rgb_values", "depth_values", "normal_map", "grad_theta", "sample_sdf"):
            close(out_graph[k], out_eager[k], 1e-5, 1e-6, k)
        assert float(g_graph.abs().max()) > 0
        rel = float((g_graph - g_eager).norm() / g_eager.norm())
        assert rel < 1e-4, rel


def test_graph_training_reduces_loss():
    """A few dozen graph-replayed iterations with the fused Adam actually train (loss goes down, nothing blows up)."""
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    tr = Stage1Trainer(stock_conf(num_rays=256, S=32, d_out=4, num_levels=8, end_size=256, logmap=14, beta=0.05), device=DEV,
                       optimizer="flat", graph=True)
    benchmark_model_state(tr.model, 0.05)
    scene = SyntheticScene(256, 4, img_res=(64, 64), num_frames=1, ring=1, device=DEV, redraw=False)   # overfit ONE batch
    losses = []
    for _ in range(40):
        idx, mi, gt = scene.next_batch()
        _, lo = tr.train_step(idx, mi, gt)
        losses.append(float(lo["loss"]))
    assert all(l == l and abs(l) < 1e6 for l in losses)
    assert sum(losses[-5:]) / 5 < sum(losses[:5]) / 5
    st = tr.flat.read_state()
    assert st.step >= 40 and 0 < st.lr[1] <= 5e-4


@pytest.mark.parametrize("R,N,K", [(37, 98, 32), (5, 50, 2), (3, 146, 21), (4, 14, 5)])
def test_fused_composite_matches_torch_formulation(R, N, K):
    """csrc/composite.hip vs the whole-tensor torch formulation of volume_rendering / occlusion_opacity / weighted sums
    (fp32 reference of the same op): every output and every input gradient, incl. d/d beta."""
    from holoscene_amd.model.network import _composite
    from holoscene_amd.model.density import laplace_density
    g = torch.Generator().manual_seed(R * 1000 + N)
    z = torch.sort(torch.rand(R, N, generator=g) * 3.0, -1)[0].to(DEV)
    z[:, 0] = 0.0
    sdf = (torch.randn(R * N, 1, generator=g) * 0.2).to(DEV).requires_grad_(True)
    raw = (torch.randn(R * N, K, generator=g) * 0.2).to(DEV).requires_grad_(True)
    rgb = torch.rand(R * N, 3, generator=g).to(DEV).requires_grad_(True)
    grd = torch.randn(R * N, 3, generator=g).to(DEV).requires_grad_(True)
    beta = torch.tensor(0.07, device=DEV, requires_grad=True)
    ds = (torch.rand(R, 1, generator=g) + 0.5).to(DEV)
    sem_scale = 10.0

    def reference():
        sig = laplace_density(sdf, beta).reshape(R, N)
        dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10, device=DEV)], -1)
        fe = dists * sig
        T = torch.exp(-torch.cumsum(torch.cat([torch.zeros(R, 1, device=DEV), fe[:, :-1]], -1), -1))
        w = (1 - torch.exp(-fe)) * T
        osig = laplace_density(raw, beta).transpose(0, 1).reshape(K, R, N)
        opac = ((1 - torch.exp(-dists * osig)) * T).sum(-1).transpose(0, 1)
        sem = (sem_scale * torch.sigmoid(-sem_scale * raw)).reshape(R, N, K)
        n = (grd / (grd.norm(2, -1, keepdim=True) + 1e-6)).reshape(R, N, 3)
        return (w, (w[..., None] * rgb.reshape(R, N, 3)).sum(1), ds * ((w * z).sum(1, keepdim=True) / (w.sum(1, keepdim=True) + 1e-8)),
                (w[..., None] * n).sum(1), (w[..., None] * sem).sum(1), opac)

    ref = reference()
    w, _, rgbv, depth, nmap, semv, opac = _composite.apply(z, sdf, raw, rgb, grd, beta, ds, sem_scale)
    out = (w, rgbv, depth, nmap, semv, opac)
    names = ("weights", "rgb_values", "depth_values", "normal_map", "semantic_values", "object_opacity")
    for a, b, n_ in zip(out, ref, names):
        close(a, b, 2e-4, 2e-6, n_)
    cot = [torch.randn(t.shape, generator=g).to(DEV) for t in ref]
    ins = (sdf, raw, rgb, grd, beta)
    g_ref = torch.autograd.grad(ref, ins, cot)
    g_out = torch.autograd.grad(out, ins, cot)
    for a, b, n_ in zip(g_out, g_ref, ("d_sdf", "d_raw", "d_rgb", "d_g", "d_beta")):
        scale = float(b.abs().max())
        close(a, b, 2e-3, 2e-5 * max(scale, 1e-6), n_)


@pytest.mark.parametrize("d_out,B", [(32, 131072), (21, 1000), (2, 129), (40, 4096), (64, 70001), (33, 333)])
def test_fused_mfma_sdf_kernel_vs_torch(d_out, B):
    """csrc/sdf_mlp.hip (bf16 MFMA, fp32 accumulate) vs the plain fp32 PyTorch SDF trunk on the same weights.
    Tolerance: bf16 operand rounding (2^-8 relative per product, 256-term sums) -> 1e-2 of the value scale."""
    from holoscene_amd.model.network import ObjectImplicitNetworkGrid
    torch.manual_seed(d_out)
    net = ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=d_out, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                    divide_factor=1.0, sigmoid=10, color_grid_feature=True, num_levels=16, logmap=15, end_size=512).to(DEV)
    with torch.no_grad():
        net.lin0.weight_v[:, 3:].normal_(0, 1e-2)
        net.encoding.embeddings.uniform_(-0.5, 0.5)
    x = (torch.rand(B, 3, device=DEV) * 2.4 - 1.2)
    with torch.no_grad():
        net.set_mlp_precision("fp32")
        ref_raw = net.get_sdf_raw(x)
        ref_min = net.get_sdf_vals(x)
        net.set_mlp_precision("bf16")
        assert net._fused_sdf_supported(x)
        got_min = net.get_sdf_vals(x)
        got_raw = net.get_sdf_raw(x)
        got_sel = net.get_object_sdf_vals(x, d_out - 1)
    scale = float(ref_raw.abs().max())
    assert got_raw.shape == ref_raw.shape and got_min.shape == ref_min.shape
    assert (got_raw - ref_raw).abs().max() < 1e-2 * scale, float((got_raw - ref_raw).abs().max())
    assert (got_min - ref_min).abs().max() < 1e-2 * scale
    assert torch.equal(got_min, got_raw.min(-1, keepdim=True)[0]), "min output must be the min of the raw outputs of the same launch"
    assert torch.equal(got_sel, got_raw[:, d_out - 1])


@pytest.mark.parametrize("d_out,B", [(64, 131072), (40, 1000), (33, 65)])
def test_wide_sdf_sweep_vs_workgroup_tile_kernel(d_out, B, monkeypatch):
    """33..64 objects: the wave-tile kernel with the last layer's second tile (hs_sdf_mlp2_fwd_wide) against the workgroup-tile kernel it replaces
    for the sampler sweeps (csrc/sdf_mlp.hip) on the same weights -- minimum over all objects, one object on either side of the tile boundary, a
    subset spanning it, the raw outputs; through the ray entry point (bf16-word features, gated) too.  Both round operands to bf16 and
    accumulate in fp32: agreement to the order of summation."""
    from holoscene_amd.hashencoder import backend as be_mod
    from holoscene_amd.model import network as N
    torch.manual_seed(d_out)
    net = N.ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=d_out, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                      divide_factor=1.0, sigmoid=10, color_grid_feature=True, num_levels=16, logmap=15, end_size=512).to(DEV)
    with torch.no_grad():
        net.lin0.weight_v[:, 3:].normal_(0, 1e-2)
        net.encoding.embeddings.uniform_(-0.5, 0.5)
        net.lin2.weight_v.add_(0.05 * torch.randn_like(net.lin2.weight_v))
        net.lin2.bias.add_(0.1 * torch.randn_like(net.lin2.bias))
    net.set_mlp_precision("bf16")
    x = (torch.rand(B, 3, device=DEV) * 2.4 - 1.2)
    sub = [3, 31, 32, d_out - 1]
    calls = []
    orig = be_mod._backend.sdf_mlp2_fwd_wide
    monkeypatch.setattr(be_mod._backend, "sdf_mlp2_fwd_wide", staticmethod(lambda *a, **k: calls.append(1) or orig(*a, **k)))

    def queries():
        net.invalidate_packed_weights()
        with torch.no_grad():
            out = {"min": net.get_sdf_vals(x), "raw": net.get_sdf_raw(x), "lo": net.get_object_sdf_vals(x, 31), "hi": net.get_object_sdf_vals(x, 32),
                   "last": net.get_object_sdf_vals(x, d_out - 1), "sub": net.get_multi_object_sdf_vals(x, sub)}
            R, S = (8, B // 8) if B >= 64 else (1, B)
            xx = x[:R * S].contiguous()
            x01 = ((xx / net.divide_factor + 1.0) / 2.0).contiguous()
            a, b = torch.tensor([1.0], device=DEV), torch.tensor([2.0], device=DEV)
            out["rays"] = net.sdf_at_points(xx, x01, R, S, gate=(b, a)).reshape(-1, 1)
        return out
    wide = queries()
    assert len(calls) >= 7, calls
    assert torch.equal(wide["min"], wide["raw"].min(-1, keepdim=True)[0]) and torch.equal(wide["lo"], wide["raw"][:, 31])
    assert torch.equal(wide["hi"], wide["raw"][:, 32]) and torch.equal(wide["last"], wide["raw"][:, d_out - 1])
    assert torch.equal(wide["sub"], wide["raw"][:, sub].min(-1, keepdim=True)[0])
    assert torch.equal(wide["rays"], wide["min"][:wide["rays"].shape[0]])
    monkeypatch.setattr(N.ops, "SDF_WIDE", False)
    n = len(calls)
    tile = queries()
    assert len(calls) == n
    scale = float(tile["raw"].abs().max())
    for k in wide:
        err = float((wide[k] - tile[k]).abs().max())
        print(f"PARITY wide SDF sweep K={d_out} B={B} {k}: max abs diff from the workgroup-tile kernel {err:.3e} (scale {scale:.2f})")
        assert wide[k].shape == tile[k].shape and err < 3e-3 * scale, (k, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_input_builders_vs_torch(dtype):
    """csrc/encode_ops.hip vs the plain torch formulation (Embedder + cat), forward and backward."""
    from holoscene_amd.hashencoder import backend
    from holoscene_amd.model.embedder import Embedder
    be = backend._backend
    g = torch.Generator().manual_seed(3)
    B, nf, L, C = 777, 6, 16, 2
    x = (torch.rand(B, 3, generator=g) * 2 - 1).to(DEV)
    feat = torch.randn(B, L * C, generator=g).to(DEV)
    dydx = torch.randn(L, B, 3 * C, generator=g).to(DEV)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    out = torch.empty(B, 4, 3 + 6 * nf + L * C, device=DEV, dtype=dtype)
    be.trunk_input_fwd(x, feat, dydx, out, nf, L, C, 0.5)
    emb, ejac = Embedder(nf, 3).embed_jacobian(x)
    fj = dydx.view(L, B, 3, C).permute(1, 2, 0, 3).reshape(B, 3, L * C) * 0.5
    ref = torch.cat([torch.cat([emb, feat], -1).unsqueeze(1), torch.cat([ejac, fj], -1)], 1)
    assert torch.allclose(out.float(), ref, **tol)
    G = torch.randn(B, 4, out.shape[-1], generator=g).to(DEV).to(dtype)
    gf = torch.empty(B, L * C, device=DEV)
    gj = torch.empty(L, B, 3 * C, device=DEV)
    be.trunk_input_bwd(G, gf, gj, nf, L, C, 0.5)
    P = 3 + 6 * nf
    assert torch.allclose(gf, G[:, 0, P:].float())
    assert torch.allclose(gj, (G[:, 1:, P:].float() * 0.5).reshape(B, 3, L, C).permute(2, 0, 1, 3).reshape(L, B, 3 * C))
    # rendering-network input
    from holoscene_amd.model.network import _render_input
    pts, dirs = x, torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).to(DEV)
    nrm = torch.randn(B, 3, generator=g).to(DEV).requires_grad_(True)
    fv = torch.randn(B, 256, generator=g).to(DEV).to(dtype).requires_grad_(True)
    got = _render_input.apply(pts, dirs, nrm, fv, 4)
    e = Embedder(4, 3)
    ref = torch.cat([e.embed(pts), e.embed(dirs), e.embed(nrm), fv.float()], -1)
    assert torch.allclose(got.float(), ref, **tol)
    cot = torch.randn(ref.shape, generator=g).to(DEV)
    g_n, g_f = torch.autograd.grad(got, (nrm, fv), cot.to(dtype))
    r_n, r_f = torch.autograd.grad(ref, (nrm, fv), cot.to(dtype).float())
    assert torch.allclose(g_n, r_n, rtol=1e-4 if dtype == torch.float32 else 3e-2, atol=1e-4 if dtype == torch.float32 else 0.3)
    assert torch.allclose(g_f.float(), r_f.float(), **tol)


@pytest.mark.parametrize("R,N,K", [(1024, 98, 32), (40, 14, 5), (2048, 50, 21)])
def test_fused_loss_matches_torch_formulation(R, N, K, monkeypatch):
    """csrc/loss.hip (value + analytic gradient) vs the whole-tensor torch loss + autograd on random inputs."""
    from holoscene_amd.model import loss as loss_mod
    g = torch.Generator().manual_seed(R + K)
    H = (K + 1) * 2 * R
    outs = {"rgb_values": torch.rand(R, 3, generator=g), "depth_values": torch.rand(R, 1, generator=g) * 2 + 0.2,
            "normal_map": torch.randn(R, 3, generator=g) * 0.5, "object_opacity": torch.rand(R, K, generator=g) ** 3,
            "grad_theta": torch.randn(H, 3, generator=g), "grad_theta_nei": torch.randn(H, 3, generator=g),
            "sdf": torch.randn(R, N, generator=g), "iter_step": 5}
    outs["object_opacity"][0, :] = torch.tensor([0.0, 1.0] + [5e-5] * (K - 2))   # clip boundaries
    outs["sdf"][: R // 4] = outs["sdf"][: R // 4].abs()                         # rays that never cross a surface
    gt = {"rgb": torch.rand(1, R, 3, generator=g), "depth": torch.rand(1, R, 1, generator=g) + 0.1,
          "normal": torch.randn(1, R, 3, generator=g), "mask": (torch.rand(1, R, 1, generator=g) > 0.2).float(),
          "segs": torch.randint(0, K, (1, R, 1), generator=g)}
    gt = {k: v.to(DEV) for k, v in gt.items()}
    names = ("rgb_values", "depth_values", "normal_map", "object_opacity", "grad_theta", "grad_theta_nei")
    res = {}
    for impl in ("torch", "hip"):
        monkeypatch.setattr(loss_mod, "LOSS_IMPL", impl)
        o = {k: (v.to(DEV).clone().requires_grad_(k in names) if torch.is_tensor(v) else v) for k, v in outs.items()}
        lo = build_loss()(o, gt)
        grads = torch.autograd.grad(lo["loss"], [o[k] for k in names])
        res[impl] = (lo, grads)
    for k in ("loss", "rgb_loss", "depth_loss", "normal_l1", "normal_cos", "semantic_loss", "eikonal_loss", "smooth_loss"):
        close(res["hip"][0][k], res["torch"][0][k], 2e-4, 1e-6, k)
    for n_, a, b in zip(names, res["hip"][1], res["torch"][1]):
        close(a, b, 2e-3, 2e-5 * max(1e-6, float(b.abs().max())), "d/d" + n_)


@pytest.mark.parametrize("rays,S,K,precision", [(1024, 128, 21, "fp32"), (1024, 128, 21, "bf16"), (2048, 192, 32, "bf16"), (512, 128, 32, "bf16")])
def test_other_baseline_config_shapes(rays, S, K, precision):
    """BASELINE configs[3] (K=21 on the shared net), configs[4] (2 048 rays x 192 samples -> 146 pts/ray) and the per-rank
    shape of configs[2] (512 rays): two graph-replayed training iterations, size-independent invariants."""
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    tr = Stage1Trainer(stock_conf(num_rays=rays, S=S, d_out=K, beta=0.01, mlp_precision=precision), device=DEV, optimizer="flat", graph=True)
    benchmark_model_state(tr.model, 0.01)
    scene = SyntheticScene(rays, K, num_frames=2, ring=4, device=DEV)
    N = S // 2 + S // 4 + 2
    for it in range(2):
        idx, mi, gt = scene.next_batch()
        out, lo = tr.train_step(idx, mi, gt)
        torch.cuda.synchronize()
        z = out["z_vals"]
        assert z.shape == (rays, N) and bool((z[:, 1:] >= z[:, :-1]).all()) and float(z.max()) <= 3.5 + 1e-6
        w = out["weights"]
        assert bool((w >= -1e-6).all()) and bool((w.sum(-1) <= 1 + 1e-3).all())
        assert out["object_opacity"].shape == (rays, K) and out["grad_theta"].shape == ((K + 1) * 2 * rays, 3)
        assert bool(torch.isfinite(lo["loss"])) and float(lo["eikonal_loss"]) >= 0
        assert bool(torch.isfinite(tr.flat.flat_p).all())


def test_dense_sdf_volume_fused_vs_fp32():
    """dense-grid SDF sweep (mesh extraction input); bf16 matrix-core path vs the fp32 trunk, and the
    shift/min variants' defining properties."""
    from holoscene_amd.model.network import ObjectImplicitNetworkGrid
    from holoscene_amd.utils.sdf_grid import evaluate_sdf_volume
    torch.manual_seed(1)
    net = ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=7, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                    divide_factor=1.0, sigmoid=10, color_grid_feature=True, num_levels=16, logmap=15, end_size=512).to(DEV)
    with torch.no_grad():
        net.lin0.weight_v[:, 3:].normal_(0, 1e-2)
    res = 48
    net.set_mlp_precision("fp32")
    ref = evaluate_sdf_volume(net, res, (-1.0, 1.0), "raw", chunk=50000)
    net.set_mlp_precision("bf16")
    raw = evaluate_sdf_volume(net, res, (-1.0, 1.0), "raw", chunk=30000)      # ragged last chunk
    assert raw.shape == (res ** 3, 7)
    assert (raw - ref).abs().max() < 1e-2 * float(ref.abs().max())
    mn = evaluate_sdf_volume(net, res, (-1.0, 1.0), "min")
    assert torch.allclose(mn, raw.min(-1, keepdim=True)[0], atol=1e-6)
    sh = evaluate_sdf_volume(net, res, (-1.0, 1.0), "shift")
    inside = (mn < 0).squeeze(-1)
    assert torch.allclose(sh.min(-1)[0], mn.squeeze(-1), atol=1e-6)           # the minimal object keeps its value
    srt = torch.sort(sh[inside], -1)[0]
    assert bool((srt[:, 1] >= -srt[:, 0] - 1e-6).all())                       # every other object pushed outside it


@pytest.mark.parametrize("train", [True, False])
def test_fused_ray_setup_vs_torch_formulation(train):
    """k_ray_setup vs rend_util.get_camera_params (x2, with the 2x-offset depth-scale rays) + UniformSampler + Lemma-2 beta."""
    from holoscene_amd.utils import rend_util
    rec = load("iter_k5")
    model = build_model(rec, DEV).train(train)
    ins = _dev(section(rec, "in."))
    g = torch.Generator().manual_seed(9)
    R = ins["uv"].shape[1]
    S = model.ray_sampler.N_samples_eval
    off = (torch.rand(1, R, 2, generator=g) - 0.5).to(DEV) if train else None
    t_rand = torch.rand(R, S, generator=g).to(DEV) if train else None
    fused = model._setup_rays_fused(ins["uv"], off, ins["pose"], ins["intrinsics"], t_rand)
    dirs, loc = rend_util.get_camera_params(ins["uv"], ins["pose"], ins["intrinsics"], ray_offset=off)
    tmp, _ = rend_util.get_camera_params(ins["uv"], torch.eye(4, device=DEV)[None], ins["intrinsics"], ray_offset=None if off is None else 2 * off)
    close(fused["ray_dirs"], dirs[0], 1e-5, 1e-6, "ray_dirs")
    close(fused["cam_loc"], loc.expand(R, 3), 0, 0, "cam_loc")
    close(fused["depth_scale"], tmp[0, :, 2:], 1e-5, 1e-6, "depth_scale")
    z_ref, _, _ = model.ray_sampler.uniform_sampler.get_z_vals(dirs[0], loc.expand(R, 3).contiguous(), model, t_rand=t_rand)
    close(fused["z0"], z_ref, 1e-5, 1e-6, "z0")
    d0 = z_ref[:, 1:] - z_ref[:, :-1]
    beta_ref = torch.sqrt((1.0 / (4.0 * torch.log(torch.tensor(model.ray_sampler.eps + 1.0)))) * (d0 ** 2).sum(-1))
    close(fused["beta_init"], beta_ref, 1e-5, 1e-7, "beta_init")
    # positions of the first sweep: bit-identical to the separate positions launch on the kernel's own rays / depths
    from holoscene_amd.hashencoder import backend
    x, x01 = torch.empty(R * S, 3, device=DEV), torch.empty(R * S, 3, device=DEV)
    backend._backend.ray_points(fused["cam_loc"], fused["ray_dirs"], fused["z0"], x, x01, float(model.implicit_network.divide_factor))
    assert torch.equal(fused["x0"], x) and torch.equal(fused["x0_grid"], x01)


@pytest.mark.parametrize("d_out,B", [(32, 25088), (5, 1000), (21, 37)])
def test_fused_mfma_training_trunk_vs_gemm_path(d_out, B, monkeypatch):
    """k_trunk_fwd + the H-based backward vs (a) the library-GEMM bf16 trunk and (b) the fp32 trunk, same weights:
    values, Jacobians and the gradients of every parameter.  Tolerances: bf16 operand rounding -- 1e-2 of the scale
    of each quantity against fp32; the two bf16 paths round at different places, so they get the same bound."""
    from holoscene_amd.model import network as N
    torch.manual_seed(d_out)
    net = N.ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=d_out, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                      divide_factor=1.0, sigmoid=10, color_grid_feature=True, num_levels=16, logmap=15, end_size=512).to(DEV)
    with torch.no_grad():
        net.lin0.weight_v[:, 3:].normal_(0, 1e-2)
        net.encoding.embeddings.uniform_(-0.5, 0.5)
    x = (torch.rand(B, 3, device=DEV) * 2.4 - 1.2)
    cy, cJ = torch.randn(B, d_out, device=DEV), torch.randn(B, d_out, 3, device=DEV)
    params = [net.encoding.embeddings] + [p for l in net._lins() for p in (l.weight_v, l.weight_g, l.bias)]
    names = ["table"] + [f"lin{i}.{n}" for i in range(3) for n in ("v", "g", "bias")]

    def run(prec, impl):
        net.set_mlp_precision(prec)
        monkeypatch.setattr(N.ops, "TRUNK_IMPL", impl)
        y, J = net.sdf_and_jacobian(x)
        gr = torch.autograd.grad((y * cy).sum() + (J * cJ).sum(), params)
        return y.detach(), J.detach(), [g.float() for g in gr]

    ref = run("fp32", "gemm")
    gem = run("bf16", "gemm")
    assert net._fused_trunk_supported(x)
    got = run("bf16", "mfma")
    monkeypatch.setattr(N.ops, "TRUNK_INPUT_IN_KERNEL", True)     # the kernel assembling its own input rows: same numbers up to the
    got_in = run("bf16", "mfma")                               # hardware vs libm sin/cos before the bf16 rounding
    monkeypatch.setattr(N.ops, "TRUNK_INPUT_IN_KERNEL", False)

    def errs(a, b):
        return float((a - b).abs().max()), float(b.abs().max()), float((a - b).norm() / (b.norm() + 1e-20))

    flat = lambda r: [r[0], r[1]] + r[2]  # noqa: E731
    for a, b, n in zip(flat(got_in), flat(got), ["y", "J"] + names):
        assert errs(a, b)[2] < 5e-3, ("in-kernel input", n, errs(a, b))
    for a, g, b, n in zip(flat(got), flat(gem), flat(ref), ["y", "J"] + names):
        err, scale, rel = errs(a, b)
        gerr, _, grel = errs(g, b)
        print(f"{n:10s} mfma-vs-fp32 max {err:.3e} (scale {scale:.3e}) rel_l2 {rel:.3e} | gemm-bf16-vs-fp32 max {gerr:.3e} rel_l2 {grel:.3e}")
        # the fused path may not be worse than the established bf16 path by more than 2x, and both stay within 2e-2 of scale
        assert err <= 2e-2 * scale + 1e-6 and rel < 2e-2, (n, err, scale, rel)
        assert rel <= 2.0 * grel + 1e-3, (n, rel, grel)


@pytest.mark.parametrize("select", [-1, 3])
def test_ray_mode_sdf_query_is_bit_identical_to_point_mode(select):
    """hs_ray_points + hash encode + fused trunk == the same query on explicitly built points (same roundings)."""
    from holoscene_amd.model.network import ObjectImplicitNetworkGrid
    torch.manual_seed(5)
    net = ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=7, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                    divide_factor=1.5, sigmoid=10, color_grid_feature=True, num_levels=16, logmap=15, end_size=512).to(DEV)
    net.set_mlp_precision("bf16")
    with torch.no_grad():
        net.lin0.weight_v[:, 3:].normal_(0, 1e-2)
        net.encoding.embeddings.uniform_(-0.5, 0.5)
        R, S = 333, 17
        o = torch.randn(R, 3, device=DEV) * 0.3
        d = torch.nn.functional.normalize(torch.randn(R, 3, device=DEV), dim=-1)
        z = torch.rand(R, S, device=DEV).sort(-1)[0] * 3.5
        pts = (o.unsqueeze(1) + z.unsqueeze(2) * d.unsqueeze(1)).reshape(-1, 3)
        ref = (net.get_sdf_vals(pts) if select < 0 else net.get_object_sdf_vals(pts, select)).reshape(R, S)
        got = net.sdf_along_rays(o, d, z, select)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("name", [f"sampler_{i}" for i in range(5)] + ["sampler_eval"])
def test_sampler_control_variants_agree(name, monkeypatch):
    """Three ways to drive Algorithm 1's loop over the same kernels (bf16 fused SDF queries): the host reads a flag per round
    (reference control flow); the loop test entirely on the device with one draw + control-step launch per round between
    hsSamplerCtl slots; and the form the whole-iteration HIP graph replays -- the next round's draw fused into the update launch,
    rounds gated directly on the previous round's max beta, realised state derived once by the final draw.  Identical depths and
    realised round counts, for states that stop after 1..5 rounds, train and eval."""
    from holoscene_amd.model import ray_sampler as RS
    rec = load(name)
    model = build_model(rec, DEV)
    model.train(bool(rec["meta.train"]))
    model.implicit_network.set_mlp_precision("bf16")
    ins = _dev(section(rec, "in."))
    res = {}
    for control, spec in (("host", False), ("device", False), ("device", True)):
        monkeypatch.setattr(RS, "CONTROL", control)
        monkeypatch.setattr(RS, "FUSE_DRAW", spec)
        z, z_eik = model.ray_sampler.get_z_vals(ins["ray_dirs"], ins["cam_loc"], model, rng=_dev(rand_dict(rec)))
        res[(control, spec)] = (z, z_eik, model.ray_sampler.last_rounds)
    ref = res[("host", False)]
    for k, v in res.items():
        assert v[2] == ref[2], (k, v[2], ref[2])
        assert torch.equal(v[0], ref[0]) and torch.equal(v[1], ref[1]), k


def test_device_side_extra_sample_pick():
    """hs_sampler_pick: n distinct indices in [0, m) from uniforms (partial Fisher-Yates), and the eval-mode linspace."""
    from holoscene_amd.hashencoder import backend
    be = backend._backend
    for m in (128, 640, 37):
        ctl = torch.tensor([1.0, 0.5, 0.0, 0.0], device=DEV)
        ctl.view(torch.int32)[2] = m
        n = 32
        u = torch.rand(n, device=DEV)
        pick = torch.empty(n, device=DEV, dtype=torch.int64)
        be.sampler_pick(ctl, u, n, pick)
        p = pick.cpu().tolist()
        assert len(set(p)) == n and min(p) >= 0 and max(p) < m
        arr, uh = list(range(m)), u.cpu().tolist()      # the same shuffle on the host
        for j in range(n):
            k = min(j + int(torch.tensor(uh[j], dtype=torch.float32) * torch.tensor(float(m - j), dtype=torch.float32)), m - 1)
            arr[j], arr[k] = arr[k], arr[j]
        assert p == arr[:n]
        be.sampler_pick(ctl, None, n, pick)
        assert torch.equal(pick.cpu(), torch.linspace(0, m - 1, n).long())


def test_fused_draw_step_equals_separate_launches():
    """hs_sampler_draw_step (loop-control step + draw + positions in one launch) vs hs_sampler_step, hs_sampler_draw and
    hs_ray_points run one after the other: identical control state, depths and positions -- for a round that continues, one that
    converged (mode 0 must leave its outputs untouched, mode 1 must still draw) and one that exhausted the round budget."""
    from holoscene_amd.hashencoder import backend
    be = backend._backend
    torch.manual_seed(3)
    R, S, nr, df = 64, 16, 4, 1.5
    ld = S * nr
    for m_old, bmax, rounds_before in ((16, 0.2, 0), (32, 0.01, 1), (48, 0.2, 3)):
        m_new = m_old + S
        z = torch.sort(torch.rand(R, ld, device=DEV) * 3 + 0.1, dim=1).values.contiguous()
        sdf = torch.randn(R, ld, device=DEV) * 0.3
        beta = torch.rand(R, device=DEV) * 0.1 + 0.02
        beta0 = torch.tensor([0.05], device=DEV)
        beta_max = torch.tensor([bmax], device=DEV)
        cam, dirs = torch.randn(R, 3, device=DEV), torch.nn.functional.normalize(torch.randn(R, 3, device=DEV), dim=-1)
        for mode, n_out in ((0, S), (1, 24)):
            u = torch.rand(R, n_out, device=DEV) if mode == 1 else None
            ctl = torch.tensor([1.0, 0.5, 0.0, 0.0], device=DEV)
            ctl.view(torch.int32)[2], ctl.view(torch.int32)[3] = m_old, rounds_before
            # separate launches
            c1 = ctl.clone()
            be.sampler_step(c1, beta_max, beta0, S, nr)
            running = bool(c1[0] > c1[1])
            assert running == (bmax > 0.05 and rounds_before + 1 < nr)
            out1 = torch.full((R, n_out), -7.0, device=DEV)
            x1, x011 = torch.full((R * n_out, 3), -7.0, device=DEV), torch.full((R * n_out, 3), -7.0, device=DEV)
            gate = (c1[0:1], c1[1:2]) if mode == 0 else None
            be.sampler_draw(z, sdf, 0, beta, mode, 1e-4, u, n_out, out1, gate=gate, m_dev=c1.view(torch.int32)[2:3])
            if mode == 1 or running:
                be.ray_points(cam, dirs, out1, x1, x011, df)
            # fused
            c_out = torch.zeros(4, device=DEV)
            out2 = torch.full((R, n_out), -7.0, device=DEV)
            x2, x012 = torch.full((R * n_out, 3), -7.0, device=DEV), torch.full((R * n_out, 3), -7.0, device=DEV)
            be.sampler_draw_step(z, sdf, beta, mode, 1e-4, u, n_out, out2, ctl, c_out, beta_max, beta0, S, nr, cam, dirs, df, x2, x012)
            assert torch.equal(c_out.view(torch.int32), c1.view(torch.int32))
            assert int(c_out.view(torch.int32)[2]) == m_new
            assert torch.equal(out1, out2) and torch.equal(x1, x2) and torch.equal(x011, x012)
            assert (out2 != -7.0).all() == (mode == 1 or running)


@pytest.mark.parametrize("form", ["wave", "tile"])
@pytest.mark.parametrize("B", [25088, 1000, 37])
def test_fused_mfma_appearance_vs_gemm_path(B, form):
    """k_appear_fwd / k_appear_bwd (+ hs_pack_bf16) vs (a) the library-GEMM bf16 path and (b) the fp32 path on the same weights:
    rgb, d/d normals and the gradient of every parameter (colour table, colour MLP, rendering MLP).  bf16 operand rounding:
    relative L2 error against fp32 below 0.1 and not worse than 1.5x that of the established bf16 (library GEMM) path."""
    from holoscene_amd.model import network as N
    torch.manual_seed(B)
    net = N.ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=5, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                      divide_factor=1.5, sigmoid=10, color_grid_feature=True, num_levels=16, logmap=15, end_size=512).to(DEV)
    rn = N.RenderingNetwork(256, "idr", 9, 3, [256, 256], weight_norm=True, multires_view=4, multires_point=4, multires_normal=4).to(DEV)
    with torch.no_grad():
        net.color_encoding.embeddings.uniform_(-0.5, 0.5)
    pts = torch.rand(B, 3, device=DEV) * 2.4 - 1.2
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1)
    nrm = (torch.randn(B, 3, device=DEV) * 0.7).requires_grad_(True)
    cot = torch.randn(B, 3, device=DEV)
    mlp, enc = net.color_grid_feature_map_mlp, net.color_encoding
    params = [enc.embeddings, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias] + [p for l in (rn.lin0, rn.lin1, rn.lin2)
                                                                                          for p in (l.weight_v, l.weight_g, l.bias)]
    names = ["table", "c0.w", "c0.b", "c1.w", "c1.b"] + [f"r{i}.{n}" for i in range(3) for n in ("v", "g", "bias")]

    def run(prec, fused):
        net.set_mlp_precision(prec)
        rn.set_mlp_precision(prec)
        if fused:      # both forms of the fused colour branch: csrc/appearance2.hip (wave tiles) and csrc/appearance_mlp.hip (workgroup tiles)
            fn = N._fused_appearance_wave if form == "wave" else N._fused_appearance
            rgb = fn.apply(pts, dirs, nrm, enc.embeddings, enc.offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                           float(net.divide_factor), mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, rn.lin0.weight,
                           rn.lin0.bias, rn.lin1.weight, rn.lin1.bias, rn.lin2.weight, rn.lin2.bias)
        else:
            rgb = rn(pts, nrm, dirs, net._color_features(pts))
        gr = torch.autograd.grad((rgb * cot).sum(), [nrm] + params)
        return [rgb.detach()] + [g.float() for g in gr]

    ref, gem, got = run("fp32", False), run("bf16", False), run("bf16", True)

    def errs(a, b):
        return float((a - b).abs().max()), float(b.abs().max()), float((a - b).norm() / (b.norm() + 1e-20))

    for a, g, b, n in zip(got, gem, ref, ["rgb", "d_normals"] + names):
        err, scale, rel = errs(a, b)
        gerr, _, grel = errs(g, b)
        print(f"{n:10s} mfma-vs-fp32 max {err:.3e} (scale {scale:.3e}) rel_l2 {rel:.3e} | gemm-bf16-vs-fp32 max {gerr:.3e} rel_l2 {grel:.3e}")
        assert a.shape == b.shape
        # this random state (cotangents ~N(0,1) on saturating sigmoids) loses 5-7 % to bf16 rounding in BOTH bf16 paths for the
        # quantities deep in the chain (d/d normals, table); the criterion is therefore relative to the established path
        assert rel < 0.1, (n, err, scale, rel)
        assert rel <= 1.5 * grel + 1e-3, (n, rel, grel)


def _encode_relu_masks(layers, Bn):
    """bool [Bn, 256] sign tensors of (hc, r0, r1) -> the ballot words hs_appearance_fwd leaves for the backward kernel (the inverse of the
    decoding in test_appearance_relu_masks_equal_saved_signs)."""
    ntiles = (Bn + 127) // 128
    wave, k, lane = np.meshgrid(np.arange(8), np.arange(64), np.arange(64), indexing="ij")
    nq, ph, j, pt, q, nt = wave & 3, wave >> 2, k & 3, (k >> 2) & 1, (k >> 3) & 3, k >> 5
    row = ph * 64 + pt * 32 + (lane & 31)
    neuron = nq * 64 + nt * 32 + q * 8 + 4 * (lane >> 5) + j
    words = np.zeros((ntiles, 3, 8, 64), dtype=np.uint64)
    shifts = np.arange(64, dtype=np.uint64)
    for layer, m in enumerate(layers):
        full = np.zeros((ntiles * 128, 256), dtype=bool)
        full[:Bn] = m.cpu().numpy()
        for tile in range(ntiles):
            bits = full[tile * 128:(tile + 1) * 128][row, neuron].astype(np.uint64)        # [wave, k, lane]
            words[tile, layer] = (bits << shifts).sum(-1, dtype=np.uint64)
    return torch.from_numpy(words.view(np.int64).reshape(-1))


@pytest.mark.parametrize("B", [25088, 1000])
def test_fused_appearance_backward_on_fp32_relu_masks_is_tight(B, monkeypatch):
    """Where the bf16 colour branch loses its 3-7 % of gradient parity (tools/exp/bf16_colour_branch_ab.py): ReLU units whose fp32
    pre-activation lies within the bf16 rounding error of zero (7e-4 of the mask bits) switch their whole gradient on or off -- a
    relative-L2 effect of sqrt(fraction) -- while bf16-stored cotangents and bf16 per-slice partials add nothing measurable.  So the
    SHARP check of k_appear_bwd + k_wgrad_rows is on the masks of the fp32 forward: with those fed to the backward kernel (its ballot
    words overwritten) every gradient must agree with fp32 autograd ~10x more tightly than on its own masks -- a dropped 5 % term
    cannot hide in this bound."""
    from holoscene_amd.model import network as N
    from holoscene_amd.hashencoder import backend as Bk
    torch.manual_seed(B + 1)
    net = N.ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=5, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                      divide_factor=1.5, sigmoid=10, color_grid_feature=True, num_levels=16, logmap=15, end_size=512).to(DEV)
    rn = N.RenderingNetwork(256, "idr", 9, 3, [256, 256], weight_norm=True, multires_view=4, multires_point=4, multires_normal=4).to(DEV)
    with torch.no_grad():
        net.color_encoding.embeddings.uniform_(-0.5, 0.5)
    pts = torch.rand(B, 3, device=DEV) * 2.4 - 1.2
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, device=DEV), dim=-1)
    nrm = (torch.randn(B, 3, device=DEV) * 0.7).requires_grad_(True)
    cot = torch.randn(B, 3, device=DEV) * 0.05
    mlp, enc = net.color_grid_feature_map_mlp, net.color_encoding
    params = [enc.embeddings, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias] + [p for l in (rn.lin0, rn.lin1, rn.lin2)
                                                                                          for p in (l.weight_v, l.weight_g, l.bias)]
    names = ["table", "c0.w", "c0.b", "c1.w", "c1.b"] + [f"r{i}.{n}" for i in range(3) for n in ("v", "g", "bias")]
    # fp32 forward, layer by layer, for the masks
    net.set_mlp_precision("fp32")
    rn.set_mlp_precision("fp32")
    with torch.no_grad():
        feat = enc(pts / net.divide_factor)
        hc = torch.relu(feat @ mlp[0].weight.t() + mlp[0].bias)
        fv = hc @ mlp[2].weight.t() + mlp[2].bias
        emb = rn.embedview_fn
        x = torch.cat([emb(pts), emb(dirs), emb(nrm.detach()), fv], -1)
        r0 = torch.relu(x @ rn.lin0.weight.t() + rn.lin0.bias)
        r1 = torch.relu(r0 @ rn.lin1.weight.t() + rn.lin1.bias)
    words32 = _encode_relu_masks([hc > 0, r0 > 0, r1 > 0], B).to(DEV)
    rgb = rn(pts, nrm, dirs, net._color_features(pts))
    ref = [g.float() for g in torch.autograd.grad((rgb * cot).sum(), [nrm] + params)]

    def fused(swap):
        net.set_mlp_precision("bf16")
        rn.set_mlp_precision("bf16")
        be = Bk._backend
        raw = be.appearance_bwd
        if swap:
            def with_fp32_masks(*a, **k):
                a = list(a)
                assert a[-1] is not None and a[-1].numel() == words32.numel()
                a[-1] = words32
                return raw(*a, **k)
            monkeypatch.setattr(type(be), "appearance_bwd", staticmethod(with_fp32_masks))
        rgb = N._fused_appearance.apply(pts, dirs, nrm, enc.embeddings, enc.offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                                        float(net.divide_factor), mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, rn.lin0.weight,
                                        rn.lin0.bias, rn.lin1.weight, rn.lin1.bias, rn.lin2.weight, rn.lin2.bias)
        out = [g.float() for g in torch.autograd.grad((rgb * cot).sum(), [nrm] + params)]
        if swap:
            monkeypatch.setattr(type(be), "appearance_bwd", staticmethod(raw))
        return out

    own, aligned = fused(False), fused(True)
    worst_own = worst_al = 0.0
    for a, o, b, n in zip(aligned, own, ref, ["d_normals"] + names):
        ra = float((a - b).norm() / (b.norm() + 1e-20))
        ro = float((o - b).norm() / (b.norm() + 1e-20))
        print(f"PARITY appearance-bwd {n:10s} relL2 vs fp32: own bf16 masks {ro:.3e} | fp32 masks {ra:.3e}")
        if not n.startswith("r2."):       # (the output layer sits behind no mask: both columns are the same number, pure bf16 rounding)
            worst_own, worst_al = max(worst_own, ro), max(worst_al, ra)
        # measured: 2.7e-3 .. 5.6e-3 behind the masks (6.8e-2 on the kernel's own masks); output layer up to 2.0e-2 at B = 1000
        assert ra < (3e-2 if n.startswith("r2.") else 1.2e-2), (n, ra)
    assert worst_al < 0.2 * worst_own, (worst_al, worst_own)


def _full_graph_trainer(beta, freeze, rays=256):
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    tr = Stage1Trainer(stock_conf(num_rays=rays, S=32, d_out=4, num_levels=16, end_size=512, logmap=15, beta=beta, mlp_precision="bf16"), device=DEV,
                       optimizer="flat", graph=True, freeze_parameters=freeze)
    benchmark_model_state(tr.model, beta)
    assert tr._full_graph_ok()
    return tr, SyntheticScene(rays, 4, img_res=(64, 64), num_frames=3, ring=4, device=DEV)


def test_table_steps_inside_the_scatters_train_like_the_separate_adam_kernel(monkeypatch):
    """Reduce-and-step (hsTableStep: the hash tables' Adam update inside their scatter's reduction, FlatAdam.table_steps) against the plain
    whole-iteration graph (HOLOSCENE_TABLE_STEP=0): same batches, same draws.  Iteration 0 renders the background patch -- two producers
    for the geometry table: the first accumulates the plain way, the last one steps (hsTableStep.prior) --, iterations 1.. are single-producer.
    After two iterations the moments (linear in the gradients) and the parameter updates agree to what float-atomic ordering does to
    two runs of ONE path; the gradient tables are all zero after every iteration; the optimiser state advances once per iteration."""
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("HOLOSCENE_TABLE_STEP", mode)
        torch.manual_seed(0)
        tr, scene = _full_graph_trainer(0.05, False)
        assert tr._table_step == (mode == "1")
        start = tr.flat.flat_p.clone()
        torch.cuda.manual_seed(7)
        tr.model.rng_state(DEV)[1] = 7
        losses = []
        for it in range(2):
            idx, mi, gt = scene.next_batch()
            _, lo = tr.train_step(idx, mi, gt)
            losses.append(float(lo["loss"]))
            if mode == "1":
                assert not bool(tr.flat.flat_g[:tr.flat.tables_end].any()), it
        assert int(tr.flat.read_state().step) == 2
        snap = (tr.flat.flat_p.clone() - start, tr.flat.flat_m.clone(), tr.flat.flat_v.clone(), losses)
        for _ in range(58):
            idx, mi, gt = scene.next_batch()
            _, lo = tr.train_step(idx, mi, gt)
            losses.append(float(lo["loss"]))
        # (random images at the stock learning rates: every precision -- fp32 included -- spikes to 12-16 between iterations 10 and 40 and
        #  settles near 3.4 by iteration 50, tools/exp/loss_seq.py; medians, not sums of five)
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        assert all(l == l and abs(l) < 1e6 for l in losses) and med(losses[-10:]) < med(losses[:5]), losses
        if mode == "1":
            # both variants step inside their scatters: the background-patch one sends the geometry table two scatters (the patch's trunk, then
            # the main pass's): the first accumulates, the last steps and adds what it finds (hsTableStep.prior)
            assert tr._table_step_ok == {(True, False): True, (False, False): True}
            assert sorted(tr._table_producers[(True, False)]) == [1, 2] and tr._table_producers[(False, False)] == [1, 1]
            assert not bool(tr.flat.flat_g[:tr.flat.tables_end].any())
        res[mode] = (snap, tr.flat.tables_end)
    (du0, m0, v0, l0), te = res["0"]
    (du1, m1, v1, l1), _ = res["1"]
    assert abs(l0[0] - l1[0]) <= 1e-5 * abs(l0[0]) and abs(l0[1] - l1[1]) <= 2e-3 * abs(l0[1]), (l0[:2], l1[:2])
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))  # noqa: E731
    cos = lambda a, b: float((a.double() @ b.double()) / (a.double().norm() * b.double().norm()).clamp_min(1e-30))  # noqa: E731
    stats = {"tables m relL2": rel(m1[:te], m0[:te]), "tables v relL2": rel(v1[:te], v0[:te]), "tables 1 - cos(update)": 1 - cos(du1[:te], du0[:te]),
             "MLPs m relL2": rel(m1[te:], m0[te:]), "MLPs 1 - cos(update)": 1 - cos(du1[te:], du0[te:])}
    for k, v in stats.items():
        print(f"PARITY table steps vs separate Adam after 2 iterations: {k} {v:.3e}")
    # measured: 5e-6 / 7e-7 / 5e-11 for the tables, 3e-7 / 5e-13 for the MLPs (the one plain iteration in front leaves both runs in the
    # same state up to float-atomic ordering, and so does the stepped one)
    assert stats["tables m relL2"] < 1e-3 and stats["tables v relL2"] < 1e-3 and stats["tables 1 - cos(update)"] < 1e-4, stats
    assert stats["MLPs m relL2"] < 1e-3 and stats["MLPs 1 - cos(update)"] < 1e-4, stats


@pytest.mark.parametrize("density", [0.0, 0.08, 0.6, 1.0])
def test_fused_background_smoothness_vs_torch_formulation(density, monkeypatch):
    """k_bg_smooth (value + analytic gradient of HoloSceneLoss.get_bg_render_loss, loss.py:519-557) vs the whole-tensor
    formulation differentiated by autograd: empty, sparse, dense and full occlusion masks."""
    from holoscene_amd.model import loss as loss_mod
    torch.manual_seed(int(density * 100))
    L = build_loss()
    labels = (torch.rand(1024, 1, device=DEV) < density).long() * torch.randint(1, 5, (1024, 1), device=DEV)
    res = {}
    for impl in ("torch", "hip"):
        monkeypatch.setattr(loss_mod, "LOSS_IMPL", impl)
        d = (torch.rand(1024, 1, device=DEV) * 3).requires_grad_(True) if impl == "torch" else res["torch"][1].detach().clone().requires_grad_(True)
        n = torch.randn(1024, 3, device=DEV).requires_grad_(True) if impl == "torch" else res["torch"][2].detach().clone().requires_grad_(True)
        mask = (labels != 0).int()
        v = L.get_bg_render_loss(d, n, mask, labels=labels)
        (v * 1.7).backward()
        res[impl] = (v.detach(), d, n)
    close(res["hip"][0], res["torch"][0], 1e-5, 1e-6, "bg smoothness value")
    for i, name in ((1, "d/d depth"), (2, "d/d normal")):
        g_h, g_t = res["hip"][i].grad, res["torch"][i].grad
        if g_t is None:     # empty mask: the whole-tensor formulation may not reach the inputs at all
            assert float(g_h.abs().max()) == 0.0
        else:
            close(g_h, g_t, 1e-5, 1e-7, name)


@pytest.mark.parametrize("trunk_mode", ["jac", "rr"])
def test_fused_background_pass_vs_torch_formulation(trunk_mode, monkeypatch):
    """render()'s background-surface pass through the main pass's kernels (trunk + split, compositing twice) vs the whole-tensor
    formulation on the same bf16 trunk: label map, depth and normal map of the 32x32 patch, and the gradient of a scalar of them
    with respect to the geometry hash table and the trunk weights.  trunk_mode "jac": both sides run the value+Jacobian kernels (same
    arithmetic: tight bounds); "rr" (the default): the fused side takes d min / dx by the reverse-over-reverse kernels -- another
    order of bf16 roundings, so bf16-level bounds."""
    from holoscene_amd.model import network as N
    monkeypatch.setattr(N.ops, "TRUNK_MODE", trunk_mode)
    tight = trunk_mode == "jac"
    tr, scene = _full_graph_trainer(0.01, True)
    model = tr.model.train()
    _, ins, _ = scene.next_batch()
    torch.manual_seed(5)
    with torch.no_grad():
        rng = model.draw_uniforms(ins["uv"].shape[1], DEV)
        rays = model.prepare_rays(ins, rng)
        z, z_eik = model.sample(rays, rng)
        bg = model.prepare_background(ins)
    params = [model.implicit_network.encoding.embeddings] + [p for l in model.implicit_network._lins() for p in l.parameters()] + [model.density.beta]
    cot_d, cot_n = torch.randn(1024, 1, device=DEV), torch.randn(1024, 3, device=DEV)
    res = {}
    for impl in ("torch", "hip"):
        monkeypatch.setattr(N.ops, "BG_IMPL", impl)
        out = model.render(rays, z, z_eik, None, rng=rng, bg=dict(bg))
        val = (out["bg_depth_values"] * cot_d).sum() + (out["bg_normal_map"] * cot_n).sum()
        grads = torch.autograd.grad(val, params, allow_unused=True)
        res[impl] = (out["bg_mask"], out["bg_depth_values"].detach(), out["bg_normal_map"].detach(), grads)
    a, b = res["hip"], res["torch"]
    assert float((a[0] != b[0]).float().mean()) <= (0.01 if tight else 0.03)          # argmax labels: ties aside, identical
    if not tight:
        # the labels against an fp32 evaluation of the same patch.  At this model state the K rows of the last layer's matrix differ by ~1e-4
        # around 0.11 -- below the grid of ONE bf16 plane --, so the label arg-max is decided by what the second plane of W2 carries
        # (wave_tile.h): with single-plane value products (rounds 1-4) 4.1 % of these labels differed from fp32, now 0.3 %.
        net = model.implicit_network
        net.set_mlp_precision("fp32")
        model.rendering_network.set_mlp_precision("fp32")
        monkeypatch.setattr(N.ops, "BG_IMPL", "torch")
        with torch.no_grad():
            ref_mask = model.render(rays, z, z_eik, None, rng=rng, bg=dict(bg))["bg_mask"]
        net.set_mlp_precision("bf16")
        model.rendering_network.set_mlp_precision("bf16")
        off_fused, off_plain = float((a[0] != ref_mask).float().mean()), float((b[0] != ref_mask).float().mean())
        print(f"PARITY bg pass rr labels differing from the fp32 evaluation: fused {off_fused:.4f}, whole-tensor bf16 {off_plain:.4f}")
        assert off_fused <= 0.01 and off_plain <= 0.01, (off_fused, off_plain)
    close(a[1], b[1], 1e-4 if tight else 5e-3, 1e-5 if tight else 5e-3, "bg depth")
    close(a[2], b[2], 1e-4 if tight else 1e-2, 1e-5 if tight else 3e-2, "bg normal map")       # rr: measured max 1.3e-2
    for ga, gb, p in zip(a[3], b[3], params):
        assert (ga is None) == (gb is None)
        if ga is not None:
            rel = float((ga - gb).norm() / gb.norm().clamp(min=1e-20))
            print(f"PARITY bg pass {trunk_mode} grad {tuple(p.shape)} relL2 {rel:.3e}")
            assert rel < (2e-2 if tight else 6e-2), (tuple(p.shape), rel)       # bf16 cotangent images on both sides (split kernel vs autograd chain)


def test_resident_batch_gather_equals_indexed_batches():
    """SyntheticScene.write_batch (one hs_gather_rows launch into existing buffers) vs next_batch (torch fancy indexing) on two
    identically seeded scenes, and a trainer stepping through train_step_resident vs train_step on the same batches."""
    from holoscene_amd.training.synthetic import SyntheticScene
    a = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, ring=4, device=DEV)
    b = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, ring=4, device=DEV)
    _, mi, gt = a.next_batch()
    dst_i = {k: torch.full_like(v, -3) for k, v in mi.items()}
    dst_g = {k: torch.full_like(v, -3) for k, v in gt.items()}
    b.write_batch(dst_i, dst_g)
    for _ in range(5):   # walks the ring past its end
        for k in mi:
            assert torch.equal(mi[k], dst_i[k]), k
        for k in gt:
            assert torch.equal(gt[k], dst_g[k]), k
        _, mi, gt = a.next_batch()
        b.write_batch(dst_i, dst_g)
    tr1, s1 = _full_graph_trainer(0.05, True)      # parameters frozen: the two trainers stay on identical weights
    tr2, s2 = _full_graph_trainer(0.05, True)
    tr2.model.load_state_dict(tr1.model.state_dict())
    for i in range(4):
        torch.manual_seed(100 + i)
        _, l1 = tr1.train_step(*s1.next_batch())
        torch.manual_seed(100 + i)
        _, l2 = tr2.train_step_resident(s2)
        assert abs(float(l1["loss"]) - float(l2["loss"])) <= 2e-4 * abs(float(l1["loss"])), (i, float(l1["loss"]), float(l2["loss"]))


@pytest.mark.parametrize("name", ["multi_obj_k5", "multi_obj_k5_eval"])
def test_stage23_entry_points(name):
    """forward_multi_obj* / get_colors_* (network.py:1016-1801) on the HIP path (fused sampler incl. object-list
    and near/far variants, hash kernels, value+Jacobian trunk) against what the reference returned, draws injected."""
    from model_helpers import check_multi_obj
    rec = load(name)
    model = build_model(rec, DEV)
    model.train(bool(rec["meta.train"]))
    check_multi_obj(model, rec, DEV, rtol=5e-4, atol=1e-4, strict=False)      # on the reference depths: measured <= 1e-5


def test_stage23_entry_points_fused_colour_path(monkeypatch):
    """The Stage-2/3 entry points on the stock-shaped bf16 model: colour through the matrix-core appearance kernels vs through
    library GEMMs, same depths (the sampler is made to answer with fixed depths), values and table gradients."""
    from holoscene_amd.model import network as N
    tr, _ = _full_graph_trainer(0.02, True)
    model = tr.model.train()
    R = 256
    g = torch.Generator().manual_seed(2)
    o = (torch.randn(R, 3, generator=g) * 0.05 + torch.tensor([0.7, 0.0, 0.0])).to(DEV)
    d = -o + torch.randn(R, 3, generator=g).to(DEV) * 0.3
    pose = torch.eye(4, device=DEV)[None]
    z_fix = torch.sort(torch.rand(R, 40, generator=g) * 1.5 + 0.05, dim=1).values.to(DEV)
    sm = model.ray_sampler
    monkeypatch.setattr(sm, "get_z_vals", lambda *a, **k: (z_fix, None))
    monkeypatch.setattr(sm, "get_z_vals_near_far", lambda *a, **k: (z_fix, None))
    table = model.implicit_network.color_encoding.embeddings
    res = {}
    for impl in ("gemm", "mfma"):
        monkeypatch.setattr(N.ops, "APPEARANCE_IMPL", impl)
        outs = [model.forward_multi_obj_rays_subset_all_sdf(o, d, pose, [1, 2], [0, 1, 2]),
                model.forward_multi_obj_rays_subset_all_sdf_detach_rgb_for_geometry(o, d, pose, [1, 2], [0, 1, 2]),
                {"rgb_values": model.get_colors_normals_from_point_rays_obj(o, d, pose, 2)[0]}]
        val = sum(out["rgb_values"].square().sum() for out in outs)
        tr.flat.zero_grad()          # (the flat optimiser's tables receive their gradient in place, not through autograd's return path)
        val.backward()
        res[impl] = ([out["rgb_values"].detach() for out in outs], table.grad.detach().clone())
    for a, b in zip(res["mfma"][0], res["gemm"][0]):
        close(a, b, 2e-2, 2e-2, "rgb_values")
    rel = float((res["mfma"][1] - res["gemm"][1]).norm() / res["gemm"][1].norm())
    assert rel < 0.1, rel


@pytest.mark.parametrize("idx", [[1, 2], [3], 2])
def test_object_subset_sampler_device_vs_host_control(idx, monkeypatch):
    """Algorithm 1 against an object subset (the Stage-2/3 entry points' sampler call, ray_sampler.py:151-157 with idx = list): the
    subset minimum is taken inside the fused SDF sweep (object bit mask of hs_sdf_mlp_fwd), so the device-controlled loop applies;
    it must place exactly the depths the host-controlled loop places, and the masked sweep must equal min over the raw columns."""
    from holoscene_amd.model import ray_sampler as RS
    tr, scene = _full_graph_trainer(0.01, True)
    model = tr.model.train()
    _, ins, _ = scene.next_batch()
    with torch.no_grad():
        rays = model.prepare_rays(ins)
    net = model.implicit_network
    pts = torch.rand(5000, 3, device=DEV) * 1.6 - 0.8
    with torch.no_grad():
        raw = net.get_sdf_raw(pts)
        cols = idx if isinstance(idx, list) else [idx]
        got = net.get_multi_object_sdf_vals(pts, cols)
    assert torch.equal(got, raw[:, cols].min(dim=-1, keepdim=True)[0])
    g = torch.Generator().manual_seed(4)
    R, S, n, ne = rays["ray_dirs"].shape[0], model.ray_sampler.N_samples_eval, model.ray_sampler.N_samples, model.ray_sampler.N_samples_extra
    res = {}
    for control in ("host", "device"):
        monkeypatch.setattr(RS, "CONTROL", control)
        assert model.ray_sampler.device_control_ok(model, idx) == (control == "device")
        g.manual_seed(4)
        rng = {"t_rand": torch.rand(R, S, generator=g).to(DEV), "u_final": torch.rand(R, n, generator=g).to(DEV),
               "perm": torch.randperm(S * 5, generator=g)[:S].to(DEV), "eik_idx": torch.randint(n + 2 + ne, (R,), generator=g).to(DEV)}
        z, z_eik = model.ray_sampler.get_z_vals(rays["ray_dirs"], rays["cam_loc"], model, idx=idx, rng=rng)
        near = (torch.rand(R, 1, generator=g) * 0.2).to(DEV)
        far = (1.2 + torch.rand(R, 1, generator=g)).to(DEV)
        zb, zb_eik = model.ray_sampler.get_z_vals_near_far(rays["ray_dirs"], rays["cam_loc"], model, near, far, idx=idx, rng=rng)
        assert torch.equal(zb[:, :1], near) and torch.equal(zb[:, -1:], far)
        res[control] = (z, z_eik, int(model.ray_sampler.last_rounds), zb, zb_eik)
    assert res["host"][2] == res["device"][2]
    for i in (0, 1, 3, 4):      # free sampling and sampling between supplied per-ray bounds: identical depths either way
        assert torch.equal(res["host"][i], res["device"][i]), i


@pytest.mark.parametrize("name", ["net_k2", "net_k21", "net_k32"])
def test_network_query_methods(name):
    """G3 / G5 of SURVEY 8c on the HIP path: ObjectImplicitNetworkGrid's query methods (hash kernels, value+Jacobian trunk),
    RenderingNetwork.forward, volume_rendering, occlusion_opacity against direct calls of the reference's (K = 2, 21, 32)."""
    from model_helpers import check_network_methods
    rec = load(name)
    model = build_model(rec, DEV).eval()
    check_network_methods(model, rec, DEV, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("flat", [False, True])
def test_three_training_steps_match_reference(flat):
    """G6 / G7 of SURVEY 8c on the HIP path: three consecutive iterations (fresh batch and draws per step) with torch.optim.Adam +
    ExponentialLR, and with the fused flat Adam (csrc/optim.hip): losses, learning rates, and the parameter UPDATES after steps 1
    and 3 against the reference's run.  (Adam's update is ~lr * sign(g) where |g| >> eps: a sample that slides inside its bracket
    can flip the sign of a near-zero table gradient, so updates are compared in norm, not element by element.)"""
    import numpy as np
    from model_helpers import run_three_steps
    rec = load("steps3_k3")
    model = build_model(rec, DEV).train()
    before = {k: p.detach().cpu().clone() for k, p in model.named_parameters()}
    losses, lrs, snaps = run_three_steps(model, rec, DEV, flat=flat)
    for i, l in enumerate(losses):
        assert abs(l - float(rec[f"s{i}.loss"])) <= 1e-3 * abs(float(rec[f"s{i}.loss"])), (i, l, float(rec[f"s{i}.loss"]))     # measured: <= 1e-5
    assert np.allclose(np.array(lrs), rec["lr_after_step"], rtol=1e-6, atol=0)
    report = {}
    for n_steps in (1, 3):
        for k, v in section(rec, f"adam{n_steps}.").items():
            d_ref, d_got = v - before[k], snaps[n_steps][k] - before[k]
            if float(d_ref.norm()) == 0.0:
                assert float(d_got.norm()) == 0.0, k
                continue
            report[(n_steps, k)] = float((d_got - d_ref).norm() / d_ref.norm())
    worst = max(report.values())
    assert worst < 2e-2, sorted(report.items(), key=lambda kv: -kv[1])[:5]      # measured: 1.3e-3 (lin2.weight_v after 3 steps)


def test_pooled_uniform_draws_equal_explicit_draws():
    """HoloSceneNetwork.draw_uniforms hands raw U[0,1) slices of one generator launch to the kernels, which shift / scale / quantise
    them themselves (hs_ray_setup offset_shift, hs_sampler_final eik_u, hs_render_points eik_scale/shift).  The same iteration fed
    with the host-side conversions of those draws through the explicit-draw keys must give identical rays, depths and outputs."""
    tr, scene = _full_graph_trainer(0.01, True)
    model = tr.model.train()
    _, ins, _ = scene.next_batch()
    R = ins["uv"].shape[1]
    torch.manual_seed(11)
    pooled = model.draw_uniforms(R, DEV)
    n_out = model.ray_sampler.N_samples + 2 + model.ray_sampler.N_samples_extra
    b = float(model.scene_bounding_sphere)
    explicit = {"ray_offset": pooled["ray_offset_u"] - 0.5, "t_rand": pooled["t_rand"], "u_final": pooled["u_final"], "u_pick": pooled["u_pick"],
                "eik_idx": (pooled["eik_u"] * n_out).long().clamp(max=n_out - 1), "eik_uniform": pooled["eik_uniform_u"] * (2.0 * b) - b,
                "eik_jitter": pooled["eik_jitter"]}
    res = []
    for rng in (pooled, explicit):
        with torch.no_grad():
            rays = model.prepare_rays(ins, rng)
            z, z_eik = model.sample(rays, rng)
        out = model.render(rays, z, z_eik, None, rng=rng)
        res.append((rays, z, z_eik, out))
    (ra, za, ea, oa), (rb, zb, eb, ob) = res
    for k in ("ray_dirs", "cam_loc", "depth_scale", "z0", "beta_init"):
        assert torch.equal(ra[k], rb[k]), k
    assert torch.equal(za, zb) and torch.equal(ea, eb)
    for k in ("rgb_values", "depth_values", "normal_map", "grad_theta", "grad_theta_nei"):
        assert torch.equal(oa[k], ob[k]), k


@pytest.mark.parametrize("beta", [0.05, 0.002])
def test_whole_iteration_graph_matches_eager_execution(beta):
    """Rays + device-controlled sampler + render + loss + backward captured as ONE graph vs the eager execution of the same
    body on the same static inputs and generator state: same realised sampler rounds, same depths, outputs and gradients."""
    tr, scene = _full_graph_trainer(beta, True)
    for key_iter in (0, 3):  # iteration 0 renders the background patch, iteration 3 does not
        tr.iter_step = key_iter
        idx, mi, gt = scene.next_batch()
        tr.train_step(idx, mi, gt)
        entry = tr._graphs[("full", key_iter == 0, False)]

        def same_draws():       # torch's generator (background patch) and the model's own Philox stream (hs_iter_prologue) at one position
            torch.cuda.manual_seed(7)
            tr.model.rng_state(DEV)[1] = 7

        same_draws()
        entry["graph"].replay()
        torch.cuda.synchronize()
        g_graph = tr.flat.flat_g.clone()
        out_graph = {k: v.clone() for k, v in entry["out"].items() if torch.is_tensor(v)}
        loss_graph, rounds_graph = float(entry["loss"]["loss"]), int(entry["rounds"])
        same_draws()
        out_eager, loss_eager = tr._full_body(entry["static"], key_iter == 0, False)
        g_eager = tr.flat.flat_g.clone()
        assert rounds_graph == tr.model.ray_sampler.last_rounds and 1 <= rounds_graph <= 5
        assert torch.equal(out_graph["z_vals"], out_eager["z_vals"])
        z = out_graph["z_vals"]
        assert bool((z[:, 1:] >= z[:, :-1]).all()) and float(z.min()) >= 0 and float(z.max()) <= 3.5 + 1e-6
        assert abs(loss_graph - float(loss_eager["loss"])) <= 1e-4 * abs(loss_graph)
        for k in ("rgb_values", "depth_values", "normal_map", "grad_theta", "sample_sdf"):
            close(out_graph[k], out_eager[k], 1e-4, 1e-5, k)
        assert float(g_graph.abs().max()) > 0
        rel = float((g_graph - g_eager).norm() / g_eager.norm())
        assert rel < 1e-3, rel


def test_whole_iteration_graph_training_reduces_loss():
    tr, scene = _full_graph_trainer(0.05, False)
    losses = []
    for _ in range(60):
        idx, mi, gt = scene.next_batch()
        _, lo = tr.train_step(idx, mi, gt)
        losses.append(float(lo["loss"]))
    assert all(l == l and abs(l) < 1e6 for l in losses)
    # (random images at the stock learning rates: every precision, fp32 included, spikes to 12-16 between iterations 10 and 40 and settles near
    #  3.4 by iteration 50 -- tools/exp/loss_seq.py --, so medians over the settled tail, not means of five)
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    assert med(losses[-10:]) < med(losses[:5]), losses
    assert tr.flat.read_state().step >= 60


@pytest.mark.parametrize("d_out,B,n_main", [(32, 4096, 3072), (5, 1000, 1000), (21, 130, 0)])
def test_fused_trunk_render_split_equals_torch_ops_on_the_same_trunk(d_out, B, n_main):
    """_fused_trunk_render (hs_trunk_split_fwd/_bwd around the MFMA trunk) vs _fused_trunk followed by the slicing / min / gather
    ops of HoloSceneNetwork.render: same kernels underneath, so outputs agree bit for bit and gradients to bf16-image rounding."""
    from holoscene_amd.model import network as N
    torch.manual_seed(d_out)
    net = N.ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=d_out, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                      divide_factor=1.0, sigmoid=10, color_grid_feature=True, num_levels=16, logmap=15, end_size=512).to(DEV)
    net.set_mlp_precision("bf16")
    with torch.no_grad():
        net.lin0.weight_v[:, 3:].normal_(0, 1e-2)
        net.encoding.embeddings.uniform_(-0.5, 0.5)
    x = torch.rand(B, 3, device=DEV) * 2.4 - 1.2
    Be = B - n_main
    cot = [torch.randn(n_main, d_out, device=DEV), torch.randn(n_main, 1, device=DEV), torch.randn(n_main, 3, device=DEV),
           torch.randn(Be, d_out, device=DEV), torch.randn(Be, 1, device=DEV), torch.randn((d_out + 1) * Be, 3, device=DEV)]
    params = [net.encoding.embeddings] + [p for l in net._lins() for p in (l.weight_v, l.weight_g, l.bias)]
    enc = net.encoding
    l0, l1, l2 = net._lins()

    def args():   # fresh weight-norm nodes per run
        return (enc.embeddings, enc.offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), 6, 1.0, l0.weight, l0.bias, l1.weight,
                l1.bias, l2.weight, l2.bias)

    def ref():
        y, J = N._fused_trunk.apply(x, *args())
        sdf_raw, Jm = y[:n_main], J[:n_main]
        sdf, idx = sdf_raw.min(dim=-1, keepdim=True)
        grad = torch.gather(Jm, 1, idx.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
        ye, Je = y[n_main:], J[n_main:]          # the Eikonal block of HoloSceneNetwork.render / ObjectImplicitNetworkGrid.gradient
        min_e, idx_e = ye.min(dim=-1, keepdim=True)
        g_min = torch.gather(Je, 1, idx_e.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
        gtheta = torch.cat([Je.transpose(0, 1).reshape(-1, 3), g_min], 0)
        return [sdf_raw, sdf, grad, ye, min_e, gtheta], torch.cat([idx, idx_e], 0)

    def new():
        sdf_raw, sdf, idx, grad, y_e, min_e, gtheta = N._fused_trunk_render.apply(x, n_main, *args())
        return [sdf_raw, sdf, grad, y_e, min_e, gtheta], idx

    res = {}
    for name, fn in (("ref", ref), ("new", new)):
        outs, idx = fn()
        loss = sum((o * c).sum() for o, c in zip(outs, cot))
        res[name] = ([o.detach() for o in outs], idx, [g.float() for g in torch.autograd.grad(loss, params)])
    for a, b, n in zip(res["new"][0], res["ref"][0], ("sdf_raw", "sdf", "grad", "y_eik", "min_eik", "grad_theta")):
        assert torch.equal(a, b), n
    assert torch.equal(res["new"][1], res["ref"][1])
    for a, b, n in zip(res["new"][2], res["ref"][2], ["table"] + [f"lin{i}.{k}" for i in range(3) for k in ("v", "g", "bias")]):
        rel = float((a - b).norm() / (b.norm() + 1e-20))
        assert rel < 2e-3, (n, rel)    # the two routes round the same cotangent to bf16 at the same place; only summation orders differ


@pytest.mark.parametrize("d_out,B,n_main", [(64, 4096, 3072), (40, 1000, 1000), (33, 130, 0)])
def test_wide_trunk_forward_vs_workgroup_tile_kernels(d_out, B, n_main, monkeypatch):
    """33..64 objects: the training trunk's forward on the wave-tile kernel with the last layer's second tile (k_trunk_fwd2<true, true>,
    hs_trunk_mlp2_fwd_wide) against the workgroup-tile kernels + split kernel it replaces -- the seven outputs (raw SDFs, minimum, its index, its
    gradient; the Eikonal points' outputs and stacked gradient rows) and, through the SAME backward kernels, every parameter gradient.  Both
    forwards round operands to bf16 and accumulate in fp32; the index may differ only where two objects tie within that rounding."""
    from holoscene_amd.hashencoder import backend as be_mod
    from holoscene_amd.model import network as N
    torch.manual_seed(d_out)
    net = N.ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=d_out, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                      divide_factor=1.0, sigmoid=10, color_grid_feature=True, num_levels=16, logmap=15, end_size=512).to(DEV)
    net.set_mlp_precision("bf16")
    with torch.no_grad():
        net.lin0.weight_v[:, 3:].normal_(0, 1e-2)
        net.encoding.embeddings.uniform_(-0.5, 0.5)
        net.lin2.weight_v.add_(0.05 * torch.randn_like(net.lin2.weight_v))
        net.lin2.bias.add_(0.1 * torch.randn_like(net.lin2.bias))
    x = torch.rand(B, 3, device=DEV) * 2.4 - 1.2
    Be = B - n_main
    cot = [torch.randn(n_main, d_out, device=DEV), torch.randn(n_main, 1, device=DEV), torch.randn(n_main, 3, device=DEV),
           torch.randn(Be, d_out, device=DEV), torch.randn(Be, 1, device=DEV), torch.randn((d_out + 1) * Be, 3, device=DEV)]
    params = [net.encoding.embeddings] + [p for l in net._lins() for p in (l.weight_v, l.weight_g, l.bias)]
    enc = net.encoding
    l0, l1, l2 = net._lins()
    calls = []
    orig = be_mod._backend.trunk_mlp2_fwd_wide
    monkeypatch.setattr(be_mod._backend, "trunk_mlp2_fwd_wide", staticmethod(lambda *a, **k: calls.append(1) or orig(*a, **k)))

    def run(wide):
        monkeypatch.setattr(N.ops, "TRUNK_WIDE", wide)
        sdf_raw, sdf, idx, grad, y_e, min_e, gtheta = N._fused_trunk_render.apply(
            x, n_main, enc.embeddings, enc.offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), 6, 1.0, l0.weight, l0.bias, l1.weight,
            l1.bias, l2.weight, l2.bias)
        outs = [sdf_raw, sdf, grad, y_e, min_e, gtheta]
        loss = sum((o * c).sum() for o, c in zip(outs, cot))
        return [o.detach() for o in outs], idx, [g.float() for g in torch.autograd.grad(loss, params)]
    wo, wi, wg = run(True)
    assert len(calls) == 1
    to, ti, tg = run(False)
    assert len(calls) == 1
    # self-consistency of the wide outputs, bit for bit: the minimum is the minimum of the raw outputs at the reported index
    raw_all = torch.cat([wo[0], wo[3]], 0)
    min_all = torch.cat([wo[1], wo[4]], 0)
    assert torch.equal(min_all, raw_all.min(-1, keepdim=True)[0]) and torch.equal(torch.gather(raw_all, 1, wi), min_all)
    assert torch.equal(wi, raw_all.argmin(-1, keepdim=True)), "lowest index among equal minima"
    if Be:      # the last Be rows of grad_theta are the minimum's gradient = the reported object's rows
        gt = wo[5]
        per_obj = gt[:d_out * Be].view(d_out, Be, 3)
        assert torch.equal(gt[d_out * Be:], per_obj[wi[n_main:, 0], torch.arange(Be, device=DEV)])
    scale = float(to[0].abs().max()) if n_main else float(to[3].abs().max())
    same = (wi == ti).float().mean()
    print(f"PARITY wide trunk forward K={d_out}: argmin agrees on {float(same):.4f} of the points")
    assert float(same) > 0.995
    for a, b, n in zip(wo, to, ("sdf_raw", "sdf", "grad", "y_eik", "min_eik", "grad_theta")):
        if a.numel() == 0:
            continue
        keep = slice(None)
        if n == "grad":      # the gradient of the minimum follows the index
            keep = (wi[:n_main, 0] == ti[:n_main, 0])
        elif n == "grad_theta":
            keep = torch.cat([torch.ones(d_out * Be, dtype=torch.bool, device=DEV), wi[n_main:, 0] == ti[n_main:, 0]])
        err = float((a[keep] - b[keep]).abs().max())
        ref = float(b.abs().max())
        print(f"PARITY wide trunk forward K={d_out} {n}: max abs diff from the workgroup-tile path {err:.3e} (scale {ref:.2f})")
        assert err < 4e-3 * max(ref, 1.0), (n, err)
    for a, b, n in zip(wg, tg, ["table"] + [f"lin{i}.{k}" for i in range(3) for k in ("v", "g", "bias")]):
        rel = float((a - b).norm() / (b.norm() + 1e-20))
        print(f"PARITY wide trunk forward K={d_out} gradient {n}: relL2 {rel:.3e}")
        assert rel < 3e-2, (n, rel)


@pytest.mark.parametrize("Bn", [256, 300])
def test_appearance_relu_masks_equal_saved_signs(Bn):
    """The ballots hs_appearance_fwd leaves for the backward (3 layers x 8 waves x 64 words per 128-point tile) decode to exactly the signs
    of the saved layer outputs hc, r0, r1 -- the information hs_appearance_bwd used to read back from them."""
    from holoscene_amd.hashencoder import backend as B
    be = B._backend
    bf = torch.bfloat16
    torch.manual_seed(Bn)
    new = lambda c: torch.empty(Bn, c, device=DEV, dtype=bf)  # noqa: E731
    featc, pts, dirs, nrm = torch.randn(16, Bn, 2, device=DEV), torch.randn(Bn, 3, device=DEV), torch.randn(Bn, 3, device=DEV), torch.randn(Bn, 3, device=DEV)
    W = {"Wc0": (torch.randn(256, 32, device=DEV) * 0.2).to(bf), "Wc1": (torch.randn(256, 256, device=DEV) * 0.06).to(bf),
         "Wr0f": (torch.randn(256, 256, device=DEV) * 0.06).to(bf), "Wr0p": (torch.randn(256, 96, device=DEV) * 0.1).to(bf),
         "Wr1": (torch.randn(256, 256, device=DEV) * 0.06).to(bf), "Wr2": (torch.randn(32, 256, device=DEV) * 0.06).to(bf)}
    biases = [torch.randn(256, device=DEV) * 0.1 for _ in range(4)] + [torch.randn(3, device=DEV)]
    xin, hc, fv, r0, r1 = new(128), new(256), new(256), new(256), new(256)
    rgb = torch.empty(Bn, 3, device=DEV)
    words = be.appearance_mask_words(Bn)
    ntiles = (Bn + 127) // 128
    assert words == ntiles * 3 * 8 * 64
    masks = torch.zeros(words, device=DEV, dtype=torch.int64)
    be.appearance_fwd(featc, pts, dirs, nrm, W, biases, xin, hc, fv, r0, r1, rgb, masks)
    m = masks.cpu().numpy().view(np.uint64).reshape(ntiles, 3, 8, 64)
    bits = ((m[..., None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool)      # [tile, layer, wave, k, lane]
    # element (wave, k, lane) -> (row, neuron) of the tile: the decomposition both kernels' epilogues walk
    wave, k, lane = np.meshgrid(np.arange(8), np.arange(64), np.arange(64), indexing="ij")
    nq, ph, j, pt, q, nt = wave & 3, wave >> 2, k & 3, (k >> 2) & 1, (k >> 3) & 3, k >> 5
    row = ph * 64 + pt * 32 + (lane & 31)
    neuron = nq * 64 + nt * 32 + q * 8 + 4 * (lane >> 5) + j
    for layer, t in enumerate((hc, r0, r1)):
        ref = np.zeros((ntiles * 128, 256), dtype=bool)
        ref[:Bn] = (t.float() > 0).cpu().numpy()
        for tile in range(ntiles):
            got = np.zeros((128, 256), dtype=bool)
            got[row, neuron] = bits[tile, layer]
            valid = min(128, Bn - tile * 128)
            assert np.array_equal(got[:valid], ref[tile * 128:tile * 128 + valid]), (layer, tile)


@pytest.mark.parametrize("n,slices,dtype", [(256 * 256, 128, "bf16"), (32 * 256, 256, "f32"), (1028, 37, "bf16"), (4, 3, "f32"), (132, 9, "bf16")])
def test_sum_slices_vs_torch(n, slices, dtype):
    """hs_sum_slices (eight slice groups per element quad, LDS meeting point) vs torch.sum over dim 0: a partly filled last workgroup,
    fewer slices than groups; n % 4 != 0 is refused."""
    from holoscene_amd.hashencoder import backend as B
    torch.manual_seed(n + slices)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    a = torch.randn(slices, n, device=DEV).to(dt)
    b = torch.randn(slices, 2, n // 2, device=DEV).to(dt) if n % 2 == 0 else None
    outs = B._backend.sum_slices([a] + ([b] if b is not None else []))
    ref = a.float().sum(0)
    assert outs[0].shape == ref.shape and float((outs[0] - ref).abs().max()) <= 1e-5 * (1 + float(ref.abs().max())) * slices ** 0.5
    if b is not None:
        assert outs[1].shape == (2, n // 2)
        assert float((outs[1] - b.float().sum(0)).abs().max()) <= 1e-5 * (1 + float(ref.abs().max())) * slices ** 0.5
    with pytest.raises(RuntimeError):
        B._backend.sum_slices([torch.zeros(3, 6, device=DEV)])


def test_copy_many_moves_every_tensor_bit_for_bit():
    """hs_copy_many (the iteration's small parameter gradients -> their views in the flat gradient buffer, one launch): sizes from 1
    element to more than one grid pass, 16-byte aligned and unaligned views, more pairs than one launch takes (64); nothing outside the
    destinations is touched."""
    from holoscene_amd.hashencoder import backend as B
    torch.manual_seed(3)
    sizes = [1, 3, 4, 5, 255, 256, 1023, 65536, 86272, 300001] + [7 + 13 * i for i in range(60)]
    flat = torch.full((sum(sizes) + 3 * len(sizes) + 8,), -7.0, device=DEV)
    src, dst, off = [], [], 1          # start one element in: most destinations are not 16-byte aligned
    for n in sizes:
        src.append(torch.randn(n, device=DEV))
        dst.append(flat[off:off + n])
        off += n + 3
    B._backend.copy_many(dst, src)
    torch.cuda.synchronize()
    want = torch.full_like(flat, -7.0)
    off = 1
    for n, t in zip(sizes, src):
        want[off:off + n] = t
        off += n + 3
    assert torch.equal(flat, want)
    with pytest.raises(RuntimeError):
        B._backend.copy_many([flat[:4]], [torch.zeros(5, device=DEV)])


@pytest.mark.parametrize("M", [128 * 784, 128 * 64, 128 * 3])
def test_wgrad_rows_vs_matmul(M):
    """hs_wgrad_rows (csrc/wgrad.hip: split-M streaming reduction with transposing LDS reads, several products per launch) vs the fp32 product
    of the same bf16 operands: slices end in a ragged chunk (784 = 12 x 64 + 16 rows), fill exactly one, or are shorter than one."""
    from holoscene_amd.hashencoder import backend as B
    be = B._backend
    torch.manual_seed(M % 1000)
    bf = torch.bfloat16
    mats = {w: (torch.randn(M, w, device=DEV) * (torch.rand(1, w, device=DEV) + 0.1)).to(bf) for w in (32, 128, 256)}
    other = (torch.randn(M, 256, device=DEV) * 0.3).to(bf)
    pairs = [(mats[256], other), (mats[256], mats[128]), (mats[32], other), (other, mats[256])]
    stacks = be.wgrad_rows(pairs, 128)
    sums = be.sum_slices(stacks)
    for (A, Bm), st, got in zip(pairs, stacks, sums):
        assert st.shape == (128, A.shape[1], Bm.shape[1]) and st.dtype == bf
        rows = M // 128
        # every slice against its own rows (bf16 rounding of a slice's result: 2^-9 relative)
        for s_ in (0, 77, 127):
            ref = A[s_ * rows:(s_ + 1) * rows].float().t() @ Bm[s_ * rows:(s_ + 1) * rows].float()
            err = (st[s_].float() - ref).abs().max() / ref.abs().max()
            assert err < 6e-3, (A.shape, Bm.shape, s_, float(err))
        ref = A.float().t() @ Bm.float()
        rel = float((got - ref).norm() / ref.norm())
        assert rel < 2e-3, (A.shape, Bm.shape, rel)


def test_batched_weight_norm_vs_torch():
    """hs_weight_norm (all layers of a network in one launch per direction) vs torch._weight_norm and its autograd backward."""
    from holoscene_amd.model import network as N
    torch.manual_seed(3)
    shapes = [(256, 71), (256, 256), (32, 256), (3, 256), (256, 337)]
    vs = [torch.randn(*s, device=DEV, requires_grad=True) for s in shapes]
    gs = [(torch.rand(s[0], 1, device=DEV) + 0.5).requires_grad_(True) for s in shapes]
    cots = [torch.randn(*s, device=DEV) for s in shapes]
    Ws = N._weight_norm_many.apply(*[t for pair in zip(vs, gs) for t in pair])
    got = torch.autograd.grad(sum((W * c).sum() for W, c in zip(Ws, cots)), vs + gs)
    ref_W = [torch._weight_norm(v, g, 0) for v, g in zip(vs, gs)]
    ref = torch.autograd.grad(sum((W * c).sum() for W, c in zip(ref_W, cots)), vs + gs)
    for a, b in zip(Ws, ref_W):
        close(a, b, 1e-6, 1e-7, "W")
    for a, b, n in zip(got, ref, ["gv"] * 5 + ["gg"] * 5):
        assert a.shape == b.shape
        close(a, b, 2e-5, 1e-5 * float(b.abs().max()), n)


@pytest.mark.parametrize("idx", [None, 1, [0, 1]])
@pytest.mark.parametrize("per_ray", [False, True])
def test_sampler_between_supplied_bounds(idx, per_ray, monkeypatch):
    """ErrorBoundSampler.get_z_vals_near_far (reference ray_sampler.py:290-447): fused kernels vs the whole-tensor
    formulation on identical draws, scalar and per-ray bounds, scene / single-object / object-subset SDF queries."""
    from holoscene_amd.model import ray_sampler as RS
    rec = load("sampler_2")
    model = build_model(rec, DEV).train()
    ins = _dev(section(rec, "in."))
    R = ins["ray_dirs"].shape[0]
    g = torch.Generator().manual_seed(4)
    if per_ray:
        near = (torch.rand(R, 1, generator=g) * 0.3).to(DEV)
        far = (1.0 + torch.rand(R, 1, generator=g) * 1.5).to(DEV)
    else:
        near, far = 0.1, 2.2
    sm = model.ray_sampler
    n_out = sm.N_samples + 2 + sm.N_samples_extra
    rng = {"t_rand": torch.rand(R, sm.N_samples_eval, generator=g).to(DEV), "u_final": torch.rand(R, sm.N_samples, generator=g).to(DEV),
           "eik_idx": torch.randint(n_out, (R,), generator=g).to(DEV)}
    res = {}
    for impl in ("hip", "torch"):
        monkeypatch.setattr(RS, "SAMPLER_IMPL", impl)
        z, z_eik = sm.get_z_vals_near_far(ins["ray_dirs"], ins["cam_loc"], model, near, far, idx=idx,
                                          rng=dict(rng, perm=torch.arange(sm.N_samples_eval * sm.max_total_iters)))
        res[impl] = (z, z_eik, sm.last_rounds)
    z, z_eik, _ = res["hip"]
    assert z.shape == (R, n_out) and bool((z[:, 1:] >= z[:, :-1]).all())
    lo = near if per_ray else torch.full((R, 1), near, device=DEV)
    hi = far if per_ray else torch.full((R, 1), far, device=DEV)
    assert bool((z >= lo - 1e-6).all()) and bool((z <= hi + 1e-6).all())
    assert torch.equal(z[:, :1], lo) and torch.equal(z[:, -1:], hi)          # the bounds themselves are samples (:261-263)
    assert res["hip"][2] == res["torch"][2]
    z_close(z, res["torch"][0], frac_loose=0.05)
    assert torch.equal(z_eik, torch.gather(z, 1, rng["eik_idx"][:, None]))


def test_eval_mode_forward_through_fused_paths():
    """model.eval(): no Eikonal set (the split kernel sees n_main == B), linspace draws in the sampler, no ray jitter -- bf16 fused
    path vs fp32 path on the same parameters."""
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    outs = {}
    for prec in ("fp32", "bf16"):
        tr = Stage1Trainer(stock_conf(num_rays=256, S=32, d_out=4, num_levels=16, end_size=512, logmap=15, beta=0.05, mlp_precision=prec),
                           device=DEV, optimizer="torch")
        benchmark_model_state(tr.model, 0.05)
        scene = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, ring=4, device=DEV)
        _, mi, _ = scene.next_batch()
        tr.model.eval()
        with torch.no_grad():
            outs[prec] = tr.model(mi, torch.tensor([0]), iter_step=3)
    a, b = outs["fp32"], outs["bf16"]
    assert "grad_theta" not in b and b["rgb_values"].shape == (256, 3)
    z = b["z_vals"]
    assert bool((z[:, 1:] >= z[:, :-1]).all())
    # (the two runs place their samples from SDF queries of different precision, so single rays can differ visibly; the means may not)
    for k, tol in (("rgb_values", 2e-2), ("depth_values", 5e-2), ("normal_map", 5e-2)):
        d = (a[k] - b[k]).abs()
        assert float(d.mean()) < 0.2 * tol and float(d.max()) < 5 * tol, (k, float(d.mean()), float(d.max()))


def test_resident_ns_dataset_gather_equals_indexed_batches():
    """ResidentNSDataset.write_batch (one hs_gather_rows launch into a static block) == next_batch() on the same
    ring; a ragged batch (class below its quota, ns_dataset.py:422-427) is refused for the static block."""
    from test_dataset_cpu import _dataset
    rec = load("ns_sampler")
    a, b = _dataset(rec, DEV, seed=3), _dataset(rec, DEV, seed=3)
    R = int(rec["meta.R"])
    dst_in = {"uv": torch.zeros(1, R, 2, device=DEV), "pose": torch.zeros(1, 4, 4, device=DEV), "intrinsics": torch.zeros(1, 4, 4, device=DEV)}
    dst_gt = {"rgb": torch.zeros(1, R, 3, device=DEV), "depth": torch.zeros(1, R, 1, device=DEV), "normal": torch.zeros(1, R, 3, device=DEV),
              "mask": torch.zeros(1, R, 1, device=DEV), "segs": torch.zeros(1, R, 1, device=DEV)}
    done = ragged = 0
    for _ in range(16):
        _, si, gi = a.next_batch()
        if si["uv"].shape[1] != R:
            with pytest.raises(RuntimeError, match="static block"):
                b.write_batch(dst_in, dst_gt)
            ragged += 1
            continue
        b.write_batch(dst_in, dst_gt)
        for k, v in si.items():
            assert torch.equal(dst_in[k], v), k
        for k, v in gi.items():
            assert torch.equal(dst_gt[k], v), k
        done += 1
    assert done >= 8


def test_scheduled_draw_serves_the_batches_the_eager_paths_serve():
    """hs_draw_gather_sched (the batch draw driven from device memory -- a node of the training graph): next_batch() on one scene, the scheduled launch
    on an identically seeded one, 40 batches incl. ring refills (n_sched = 16) and an eager draw in between (the cursor is put back): same frames, same
    pixels, same rows; peek_batch() does not consume.  Same for ResidentNSDataset on the reference-shaped fixture (per-frame class lists and images)."""
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.datasets import pixel_sampler as ps
    a = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, seed=5, device=DEV)
    b = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, seed=5, device=DEV)
    _, mi, gt = a.peek_batch()
    _, mi2, gt2 = a.peek_batch()
    assert all(torch.equal(mi[k], mi2[k]) for k in mi) and all(torch.equal(gt[k], gt2[k]) for k in gt), "peek_batch() must not consume"
    dst_i = {k: torch.full_like(v, -3) for k, v in mi.items()}
    dst_g = {k: torch.full_like(v, -3) for k, v in gt.items()}
    sd = b.scheduled_draw(dst_i, dst_g)
    assert isinstance(sd, ps.ScheduledDraw)
    sd.schedule.n = 16
    sd.schedule.sched = torch.zeros(16, dtype=torch.int32, device=DEV)
    sd2 = b.scheduled_draw(dst_i, dst_g)         # (a plan on the resized ring; one schedule per dataset)
    assert sd2.schedule is sd.schedule
    for i in range(40):
        _, mi, gt = a.next_batch()
        if i == 21:             # an eager batch in between: both paths walk one sequence
            _, mj, gj = b.next_batch()
            assert all(torch.equal(mi[k], mj[k]) for k in mi) and all(torch.equal(gt[k], gj[k]) for k in gt)
            continue
        sd2.before_replay()
        sd2.launch()
        sd2.after_replay()
        for k in mi:
            assert torch.equal(mi[k], dst_i[k]), (i, k)
        for k in gt:
            assert torch.equal(gt[k], dst_g[k]), (i, k)
    from test_dataset_cpu import _dataset
    rec = load("ns_sampler")
    c, d = _dataset(rec, DEV, seed=3), _dataset(rec, DEV, seed=3)
    R = int(rec["meta.R"])
    dst_in = {"uv": torch.zeros(1, R, 2, device=DEV), "pose": torch.zeros(1, 4, 4, device=DEV), "intrinsics": torch.zeros(1, 4, 4, device=DEV)}
    dst_gt = {"rgb": torch.zeros(1, R, 3, device=DEV), "depth": torch.zeros(1, R, 1, device=DEV), "normal": torch.zeros(1, R, 3, device=DEV),
              "mask": torch.zeros(1, R, 1, device=DEV), "segs": torch.zeros(1, R, 1, device=DEV)}
    sn = d.scheduled_draw(dst_in, dst_gt)
    if sn is None:          # the fixture has a frame with a class below its quota: ragged batches cannot fill a static block (ns_dataset.py:422-427)
        assert any(d._sampler.count(f) != R for f in range(d.n_images))
        return
    for i in range(12):
        _, si, gi = c.next_batch()
        sn.before_replay()
        sn.launch()
        sn.after_replay()
        for k, v in si.items():
            assert torch.equal(dst_in[k], v), (i, k)
        for k, v in gi.items():
            assert torch.equal(dst_gt[k], v), (i, k)


def test_the_three_places_of_the_batch_draw_train_on_the_same_batches():
    """Stage1Trainer(draw_in_graph = True / "ahead": iteration k's graph draws batch k + 1 in its colour-table scatter launch (hs_hash_bwd_draw), one static
    batch block for all graph variants | "head": batch k by iteration k's first launch (hs_iter_prologue_draw) | False: a launch in front of every replay):
    identically seeded scenes and frozen parameters, 13 iterations across both graph variants (iterations 0 and 10 render the background patch) and one
    eagerly drawn batch in between -- the three see the same batches: equal objectives iteration by iteration."""
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    losses = {}
    for mode in (True, "head", False):
        torch.manual_seed(3)
        tr = Stage1Trainer(stock_conf(num_rays=256, S=32, d_out=4, num_levels=16, end_size=512, logmap=15, beta=0.05, mlp_precision="bf16"), device=DEV,
                           optimizer="flat", graph=True, freeze_parameters=True, draw_in_graph=mode)
        benchmark_model_state(tr.model, 0.05)
        scene = SyntheticScene(256, 4, img_res=(64, 64), num_frames=3, seed=9, device=DEV)
        out = []
        for i in range(13):
            if i == 5:      # a batch taken eagerly (and thrown away): every mode skips the same one
                scene.next_batch()
            torch.manual_seed(50 + i)       # (the model's own draws come from its Philox stream; this pins anything left to torch's generator)
            _, lo = tr.train_step_resident(scene)
            out.append(float(lo["loss"]))
        assert any(len(k) == 4 and k[3] == "sched" for k in tr._graphs) == (mode is not False), list(tr._graphs)
        losses[mode] = out
    ref = losses[False]
    for mode in (True, "head"):
        for i, (a, b) in enumerate(zip(ref, losses[mode])):
            assert abs(a - b) <= 2e-4 * abs(a), (mode, i, a, b)
    print("PARITY batch draw ahead / at the head / eager: 13 iterations, max relative objective difference",
          max(abs(a - b) / abs(a) for m in (True, "head") for a, b in zip(ref, losses[m])))


def test_background_patch_rays_placed_by_the_ray_kernel():
    """hs_ray_setup(patch_u, patch): the rays of the 32 x 32 patch two U[0, 1) draws place == the rays of the pixel grid the host-side formulation
    builds from the same draws (network.py:919-925: origin = floor(u * (floor(2 c) - patch + 1)))."""
    from holoscene_amd.training.trainer import Stage1Trainer, stock_conf
    conf = stock_conf(num_rays=256, S=32, d_out=4, num_levels=16, end_size=512, logmap=15, beta=0.05, mlp_precision="bf16")
    tr = Stage1Trainer(conf, device=DEV, optimizer="flat")
    m = tr.model.train()
    intr = torch.eye(4, device=DEV)[None].clone()
    intr[0, 0, 0] = intr[0, 1, 1] = 160.0
    intr[0, 0, 2], intr[0, 1, 2] = 150.5, 101.0
    pose = torch.eye(4, device=DEV)[None].clone()
    pose[0, :3, 3] = torch.tensor([0.1, -0.2, 0.6], device=DEV)
    P = m.BG_PATCH
    for u in ([0.0, 0.0], [0.9999, 0.9999], [0.37, 0.81]):
        pu = torch.tensor(u, device=DEV)
        t_rand = torch.rand(P * P, m.ray_sampler.N_samples_eval, device=DEV)
        span = (intr[0, :2, 2] * 2.0).floor() - P + 1
        xy0 = torch.floor(pu * span)
        gy, gx = torch.meshgrid(torch.arange(P, device=DEV), torch.arange(P, device=DEV), indexing="ij")
        uv0 = torch.stack([gx, gy], -1).reshape(1, -1, 2).float() + xy0
        want = m._setup_rays_fused(uv0, None, pose, intr, t_rand)
        got = m._setup_rays_fused(None, None, pose, intr, t_rand, patch_u=pu)
        for k in want:
            assert torch.equal(want[k], got[k]), (u, k)


def test_resident_ns_dataset_equals_the_references_batches_on_the_device():
    """SURVEY 8f rank 4 on the MI355X: the HBM-resident frames, indexed on the device with the reference's own permutation draws,
    give the reference's batches (datasets/ns_dataset.py:409-453; fixture ns_sampler, incl. its ragged batch)."""
    from test_dataset_cpu import check_batches_equal_reference
    check_batches_equal_reference(DEV)


def test_resident_ns_dataset_draws_a_new_batch_every_iteration():
    """Every batch is a new device-side draw (csrc/batch_ops.hip: hs_draw_pixels): none served twice, one sequence per seed, another per rank."""
    from test_dataset_cpu import check_batches_are_redrawn
    check_batches_are_redrawn(DEV)


@pytest.mark.parametrize("R,K,res", [(1024, 32, 512), (256, 4, 64), (64, 3, 16), (4096, 32, 512)])
def test_device_pixel_draw_follows_the_class_balanced_rule(R, K, res):
    """hs_draw_pixels against the rule of ns_dataset.py:409-430: class by class min(size, quota) DISTINCT pixels of that class (a class
    at or below its quota contributes all of its pixels -- the ragged case), then the uniform share, distinct, inside the image; the same
    (seed, counter) gives the same batch, another counter another one; every pixel of a class is reachable (coverage over many draws)."""
    from holoscene_amd.datasets.pixel_sampler import PixelSampler
    g = torch.Generator().manual_seed(R + K)
    npix = res * res
    labels = torch.randint(0, K, (npix,), generator=g)
    labels[labels == K - 1] = 0                       # class K-1 is absent from this frame
    tiny = torch.nonzero(labels == 1).reshape(-1)
    labels[tiny[3:]] = 0                              # class 1 keeps 3 pixels: below any quota -> ragged
    classes = sorted(set(labels.tolist()))
    lists = [torch.nonzero(labels == c).reshape(-1) for c in classes]
    a, b = PixelSampler([lists], npix, R, DEV, seed=5), PixelSampler([lists], npix, R, DEV, seed=5)
    n_cls, per_class, n_bg = a.quotas(0)
    n = a.count(0)
    seen = torch.zeros(npix, dtype=torch.bool)
    prev = None
    for it in range(6):
        ia, na = a.draw(0)
        ib, nb = b.draw(0)
        assert na == nb == n and torch.equal(ia[:n], ib[:n]), "one seed, one sequence"
        idx = ia[:n].cpu()
        assert prev is None or not torch.equal(idx, prev), "a new counter must give a new batch"
        prev = idx.clone()
        off = 0
        for i, (c, pix) in enumerate(zip(classes, lists)):
            want = min(len(pix), n_bg if i == 0 else per_class)
            seg = idx[off:off + want]
            assert bool((labels[seg] == c).all()), (it, c)
            assert seg.unique().numel() == want, "sampling within a class is without replacement"
            if len(pix) <= (n_bg if i == 0 else per_class):
                assert torch.equal(seg.sort()[0], pix.sort()[0]), "a class at or below its quota contributes every pixel"
            off += want
        uni = idx[off:]
        assert uni.numel() == R - R // 2 and uni.unique().numel() == uni.numel() and int(uni.min()) >= 0 and int(uni.max()) < npix
        seen[uni] = True
    if npix <= 4096:          # coverage: with enough draws the uniform share reaches (nearly) every pixel
        for _ in range(200):
            ia, _ = a.draw(0)
            seen[ia[:n].cpu()[-(R - R // 2):]] = True
        assert float(seen.float().mean()) > 0.98
    # uniformity of the uniform share: mean index of many draws ~ (npix - 1) / 2
    acc, cnt = 0.0, 0
    for _ in range(50):
        ia, _ = a.draw(0)
        u = ia[:n][-(R - R // 2):].double()
        acc, cnt = acc + float(u.sum()), cnt + u.numel()
    assert abs(acc / cnt - (npix - 1) / 2) < 4 * npix / (12 * cnt) ** 0.5


@pytest.mark.parametrize("name", ["object_sdf_fg", "object_sdf_bg"])
def test_per_object_networks_match_reference(name):
    """SURVEY 8f rank 3 on the HIP path: the per-object grid through csrc/hash_encode.hip (object-frame lookups, fused value+Jacobian
    scatter), the fused sampler kernels, K = 1 compositing (csrc/composite.hip)."""
    from object_helpers import check_object_model
    check_object_model(load(name), DEV, strict=False)


def test_eight_per_object_networks_through_one_batched_hash_launch_each_way(monkeypatch):
    """SURVEY 8(f) rank 3, "many small hash grids": ObjectSDFNetworkSet evaluates EIGHT per-object networks (fixture object_set8: the imported
    reference's ObjectSDFNetwork.forward, one by one) with ONE batched-over-grids hash gather for all their rendered + Eikonal points and ONE
    fused value+Jacobian scatter for all their table gradients (csrc/hash_encode.hip: hsHashLayout::grid_id).  Every member's outputs and
    parameter gradients must be the reference's, the members' tables must be views of the stacked parameter, and the launch counts must be 1 / 1."""
    from object_helpers import build_object_model, compare_forward_backward, sub_record
    from holoscene_amd.model.object_network import ObjectSDFNetworkSet
    from holoscene_amd.hashencoder import backend as Bk
    rec = load("object_set8")
    n = int(rec["meta.n"])
    subs = [sub_record(rec, i) for i in range(n)]
    nets = [build_object_model(r, DEV).train() for r in subs]
    group = ObjectSDFNetworkSet(nets).to(DEV)
    for g_, m in enumerate(nets):
        assert m.implicit_network.encoding.embeddings.data_ptr() == group.tables[g_].data_ptr()
        assert torch.equal(m.implicit_network.encoding.embeddings.cpu(), torch.from_numpy(subs[g_]["state.implicit_network.encoding.embeddings"]))
    be = Bk._backend
    calls = {"fwd_grids": 0, "fwd_dydx_single": 0, "jac_grids": 0, "jac_single": 0}
    raw_fwd, raw_jac = be.fwd, be.bwd_jac

    def fwd(*a, **k):
        if k.get("grids") is not None:
            calls["fwd_grids"] += 1
        elif a[10] is not None:         # a per-object lookup WITH derivatives would be a member's own value+Jacobian pass
            calls["fwd_dydx_single"] += 1
        return raw_fwd(*a, **k)

    def jac(*a, **k):
        calls["jac_grids" if k.get("grids") is not None else "jac_single"] += 1
        return raw_jac(*a, **k)
    monkeypatch.setattr(type(be), "fwd", classmethod(lambda cls, *a, **k: fwd(*a, **k)))
    monkeypatch.setattr(type(be), "bwd_jac", classmethod(lambda cls, *a, **k: jac(*a, **k)))
    dv = lambda d: {k: v.to(DEV) for k, v in d.items()}  # noqa: E731
    ins = [dv(section(r, "in.")) for r in subs]
    outs = group([i_["ray_origins"] for i_ in ins], [i_["ray_dirs"] for i_ in ins], [dv(section(r, "rand.")) for r in subs])
    loss = 0.0
    for out, r in zip(outs, subs):
        loss = loss + sum((out[k] * c.to(DEV)).sum() for k, c in section(r, "cot.").items())
    loss.backward()
    assert calls == {"fwd_grids": 1, "fwd_dydx_single": 0, "jac_grids": 1, "jac_single": 0}, calls
    tg = group.table_grads()
    for g_, (m, out, r) in enumerate(zip(nets, outs, subs)):
        grads = {k: p.grad for k, p in m.named_parameters()}
        grads["implicit_network.encoding.embeddings"] = tg[g_]           # the member's table gradient lives in the stacked parameter's
        compare_forward_backward(out, grads, r, strict=False)


def test_opt_in_finite_difference_eikonal_mode():
    """BASELINE configs[4] words a "4-tap Eikonal finite-difference"; the reference's gradients are analytic, so the FD
    form is an opt-in extra (model conf `eikonal_mode = fd`, fp32 only): on a smooth state it agrees with the analytic rows to O(h^2),
    trains (finite loss, non-zero table gradient), leaves every non-Eikonal output untouched, and refuses bf16 operands (taps 1e-3
    apart are below their resolution: measured relative distance 1.4 from the analytic rows)."""
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    precision = "fp32"
    outs = {}
    for mode in ("analytic", "fd"):
        tr = Stage1Trainer(stock_conf(num_rays=256, S=32, d_out=5, logmap=14, beta=0.05, mlp_precision=precision, eikonal_mode=mode, use_bg_reg=False),
                           device=DEV, optimizer="flat", freeze_parameters=True)
        benchmark_model_state(tr.model, 0.05)
        scene = SyntheticScene(256, 5, seed=7, device=DEV)
        idx, mi, gt = scene.next_batch()
        torch.manual_seed(3)
        out, lo = tr.train_step(idx, mi, gt)
        assert bool(torch.isfinite(lo["loss"])) and float(tr.flat.flat_g.abs().max()) > 0
        outs[mode] = (out, lo)
    a, f = outs["analytic"][0], outs["fd"][0]
    for k in ("rgb_values", "depth_values", "normal_map", "sample_sdf"):
        assert torch.allclose(a[k], f[k], rtol=1e-5, atol=1e-6), k
    ga, gf = torch.cat([a["grad_theta"], a["grad_theta_nei"]]), torch.cat([f["grad_theta"], f["grad_theta_nei"]])
    assert ga.shape == gf.shape
    rel = float((ga - gf).norm() / ga.norm())
    print(f"PARITY fd-vs-analytic Eikonal rows ({precision}) relL2 {rel:.3e}")
    assert rel < 2e-2, rel
    ea, ef = float(outs["analytic"][1]["eikonal_loss"]), float(outs["fd"][1]["eikonal_loss"])
    assert abs(ea - ef) < 5e-2 * abs(ea) + 1e-4, (ea, ef)
    tr = Stage1Trainer(stock_conf(num_rays=256, S=32, d_out=5, logmap=14, beta=0.05, mlp_precision="bf16", eikonal_mode="fd", use_bg_reg=False),
                       device=DEV, optimizer="flat", freeze_parameters=True)
    with pytest.raises(ValueError, match="fp32"):
        tr.train_step(idx, mi, gt)
    # ... and the mode survives graph capture (what `bench.py --eikonal fd --precision fp32` runs: the tap table must not be
    # uploaded inside the capture)
    from holoscene_amd.model.network import HoloSceneNetwork
    HoloSceneNetwork._fd_taps_dev.clear()
    tr = Stage1Trainer(stock_conf(num_rays=256, S=32, d_out=5, logmap=14, beta=0.05, mlp_precision="fp32", eikonal_mode="fd", use_bg_reg=False),
                       device=DEV, optimizer="flat", graph=True)
    benchmark_model_state(tr.model, 0.05)
    for _ in range(3):
        out, lo = tr.train_step(*scene.next_batch())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(lo["loss"])) and bool(torch.isfinite(tr.flat.flat_p).all())


@pytest.mark.gpu
def test_colour_branch_wave_tile_kernels_are_run_to_run_identical_at_full_size():
    """k_appear2_fwd / k_appear2_bwd at 100 352 samples, five runs each: every output bit-identical from run to run.  The kernels' chunk
    pipeline waits for its LDS-DMA requests with COUNTED vector-memory waits and bare barriers; a request consumed before it
    landed would show as a run that differs."""
    from holoscene_amd.hashencoder.backend import _backend as be
    dev, bf, n = "cuda", torch.bfloat16, 100352
    g = torch.Generator().manual_seed(5)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731
    featc = rn(16, n, 2, sc=0.3)
    points = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev)
    dirs, normals = torch.nn.functional.normalize(rn(n, 3), dim=-1), torch.nn.functional.normalize(rn(n, 3), dim=-1)
    P = be.appearance2_pack(rn(256, 32, sc=0.2), rn(256, 256, sc=0.07), rn(256, 337, sc=0.06), rn(256, 256, sc=0.07), rn(3, 256, sc=0.1),
                            (rn(256, sc=0.1), rn(256, sc=0.1), rn(256, sc=0.1), rn(256, sc=0.1), rn(3, sc=0.1)))
    tiles = (n + 31) // 32
    g_rgb = rn(n, 3)

    def once():
        tp = lambda ks: torch.full((tiles * ks * 64 * 8,), 3.0, device=dev, dtype=bf)  # noqa: E731
        f = dict(XAt=tp(8), HCt=tp(16), FVt=tp(16), R0t=tp(16), R1t=tp(16), masks=torch.zeros(tiles * 3 * 64 * 4, device=dev, dtype=torch.int32),
                 rgb=torch.empty(n, 3, device=dev))
        be.appearance2_fwd(featc, points, dirs, normals, P, f["XAt"], f["HCt"], f["FVt"], f["R0t"], f["R1t"], f["masks"], f["rgb"])
        b = dict(gy=torch.empty(n, 32, device=dev, dtype=bf), GR1=tp(16), GR0=tp(16), GFV=tp(16), GHC=tp(16), d_n=torch.empty(n, 3, device=dev),
                 g_fc=torch.empty(16, n, 2, device=dev), gb2=torch.zeros(tiles, 4, device=dev))
        be.appearance2_bwd(g_rgb, f["rgb"], normals, f["masks"], P["streamT"], b["gy"], b["GR1"], b["GR0"], b["GFV"], b["GHC"], b["d_n"], b["g_fc"], b["gb2"])
        return {**f, **b}
    first = once()
    assert torch.isfinite(first["rgb"]).all() and float(first["rgb"].min()) >= 0 and float(first["rgb"].max()) <= 1
    for run in range(4):
        again = once()
        diff = [k for k in first if not torch.equal(first[k].view(torch.int16) if first[k].dtype == bf else first[k],
                                                    again[k].view(torch.int16) if again[k].dtype == bf else again[k])]
        assert not diff, f"run {run + 1} differs from run 0 in {diff}"


@pytest.mark.parametrize("B,d_out,L", [(131072, 32, 16), (1000, 21, 16), (33, 5, 16), (4000, 2, 8), (1000, 7, 13)])
def test_fp32_fused_sdf_sweep_vs_library_gemms(B, d_out, L, monkeypatch):
    """csrc/sdf_mlp32.hip (fp32 operands on v_mfma_f32_32x32x2_f32, fp32 activations in registers, torch's Softplus formula) against the same
    queries through library GEMMs (the reference's arithmetic, model/network.py:169-210, 305-326): the minimum over all objects, one object,
    an object subset, the raw SDFs; points inside and outside the cube; ragged sizes; gated launches."""
    from holoscene_amd.model import network as N
    torch.manual_seed(d_out)
    net = N.ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=d_out, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                      divide_factor=1.0, sigmoid=10, color_grid_feature=True, num_levels=L, logmap=15, end_size=2048 if L == 16 else 256).to(DEV)       # (L < 16: empty levels behind the grid's own, zero columns in lin0 -- DESIGN 14.11)
    net.set_mlp_precision("fp32")
    with torch.no_grad():
        net.encoding.embeddings.uniform_(-0.3, 0.3)
        v = net.lin0.weight_v
        v[:, 3:] = torch.randn_like(v[:, 3:]) * 0.05
        net.lin2.weight_v.add_(0.05 * torch.randn_like(net.lin2.weight_v))
        net.lin2.bias.add_(0.1 * torch.randn_like(net.lin2.bias))
    x = torch.rand(B, 3, device=DEV) * 2.3 - 1.15
    sub = [0, d_out - 1, d_out // 2]
    with torch.no_grad():
        assert net._fused_sdf32_supported(x)
        got = {"min": net.get_sdf_vals(x), "raw": net.get_sdf_raw(x), "one": net.get_object_sdf_vals(x, d_out - 2), "sub": net.get_multi_object_sdf_vals(x, sub)}
        net.invalidate_packed_weights()
        monkeypatch.setattr(N.ops, "FP32_SDF", "gemm")
        assert not net._fused_sdf32_supported(x)
        ref = {"min": net.get_sdf_vals(x), "raw": net.get_sdf_raw(x), "one": net.get_object_sdf_vals(x, d_out - 2), "sub": net.get_multi_object_sdf_vals(x, sub)}
        monkeypatch.setattr(N.ops, "FP32_SDF", "mfma")
    for k in got:
        assert got[k].shape == ref[k].shape, k
        err = float((got[k] - ref[k]).abs().max())
        print(f"PARITY fp32 fused SDF sweep B={B} K={d_out} {k}: max abs err {err:.3e} (max |sdf| {float(ref[k].abs().max()):.2f})")
        assert err < 2e-5, (k, err)         # two fp32 implementations of the same 71 -> 256 -> 256 -> K sums (summation order differs)
    # gated launch: a closed gate leaves the output untouched
    with torch.no_grad():
        R, S = 8, B // 8
        if S > 0:
            xx = x[:R * S].contiguous()
            x01 = ((xx / net.divide_factor + 1.0) / 2.0).contiguous()
            a, b = torch.tensor([1.0], device=DEV), torch.tensor([2.0], device=DEV)
            closed = net.sdf_at_points(xx, x01, R, S, gate=(a, b))           # a > b is false: nothing runs (outputs are whatever empty() held)
            opened = net.sdf_at_points(xx, x01, R, S, gate=(b, a))
            assert closed.shape == opened.shape and torch.equal(opened.reshape(-1, 1), got["min"][:R * S])


def test_whole_iteration_at_baseline_config0_shape_vs_oracle():
    """BASELINE configs[0]'s own shape -- 256 rays x 64 samples, K = 2 (background + one object), L = 8 grid, the shape bench.py times the CPU
    oracle on -- as a whole training iteration on the HIP path (fp32, then bf16 operands; eager -- the fused L = 16 kernels and with them the
    whole-iteration graph do not apply to an L = 8 grid) against the CPU oracle on
    the same state, batch and draws: rendered outputs, every loss term, every parameter gradient.  (The reference itself never ran at this
    shape here: this pins the product to the oracle, which the reference-generated fixtures pin at theirs.)"""
    import numpy as np
    from holoscene_amd.model.network import HoloSceneNetwork
    from holoscene_amd.training.synthetic import look_at_pose
    from model_helpers import conf_from_meta
    from oracle.stage1_oracle import Cfg, Stage1Oracle, make_state
    cfg = Cfg(d_out=2, num_levels=8, base_size=16, end_size=256, logmap=15, N_samples=32, N_samples_eval=64, N_samples_extra=16, beta_init=0.05)
    sd = make_state(cfg, seed=42, perturb=1e-2)
    g = torch.Generator().manual_seed(6)
    for k in ("implicit_network.encoding.embeddings", "implicit_network.color_encoding.embeddings"):
        sd[k] = (torch.rand(sd[k].shape, generator=g) * 2 - 1) * 0.3            # surfaces inside the volume, as the measurement state has them
    meta = {"meta.S": np.array(64), "meta.feat": np.array(256), "meta.K": np.array(2), "meta.width": np.array(256), "meta.L": np.array(8),
            "meta.base": np.array(16), "meta.end": np.array(256), "meta.logmap": np.array(15), "state.density.beta": np.array(float(sd["density.beta"]))}
    R, res = 256, 512
    uv = torch.randint(0, res, (1, R, 2), generator=g).float()
    pose = look_at_pose((0.7, 0.0, 0.0))[None]
    K = torch.eye(4)[None].clone()
    K[0, 0, 0] = K[0, 1, 1] = K[0, 0, 2] = K[0, 1, 2] = res / 2
    rand = {"ray_offset": torch.rand(1, R, 2, generator=g) - 0.5, "t_rand": torch.rand(R, 64, generator=g), "u_final": torch.rand(R, 32, generator=g),
            "perm": torch.randperm(64 * 5, generator=g), "eik_idx": torch.randint(0, 50, (R,), generator=g),
            "eik_uniform": torch.rand(R, 3, generator=g) * 2 - 1, "eik_jitter": torch.rand(2 * R, 3, generator=g)}
    gt = {"rgb": torch.rand(1, R, 3, generator=g), "depth": torch.rand(1, R, 1, generator=g) * 0.9 + 0.1,
          "normal": torch.nn.functional.normalize(torch.randn(1, R, 3, generator=g), dim=-1), "mask": torch.ones(1, R, 1),
          "segs": torch.randint(0, 2, (1, R, 1), generator=g)}
    # ---- oracle
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
    full = dict(sd)
    full.update(params)
    orc = Stage1Oracle(cfg, full)
    oout = orc.forward(uv, pose, K, rand, iter_step=1)
    oloss = orc.loss(oout, gt)
    oloss["loss"].backward()
    assert 1 <= int(oout["sampler_rounds"]) <= 5
    # ---- product, fp32
    model = HoloSceneNetwork(conf=conf_from_meta(meta), graph_node_dict=None, num_images=4)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    dv = lambda d: {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()}  # noqa: E731
    out = model({"uv": uv.to(DEV), "pose": pose.to(DEV), "intrinsics": K.to(DEV)}, torch.tensor([0]), iter_step=1, rng=dv(rand))
    out["iter_step"] = 0
    lo = build_loss()(out, dv(gt))
    lo["loss"].backward()
    z_close(out["z_vals"] if "z_vals" in out else oout["z_vals"], oout["z_vals"], atol=1e-5, frac_loose=0.05)
    stats = {}
    for k in ("rgb_values", "depth_values", "normal_map"):
        stats[k] = float((out[k].detach().cpu().reshape(oout[k].shape) - oout[k].detach()).abs().max() / oout[k].detach().abs().max())
    for k in ("loss", "rgb_loss", "eikonal_loss", "depth_loss", "normal_l1", "normal_cos"):
        if k in lo and k in oloss:
            stats["loss." + k] = abs(float(lo[k]) - float(oloss[k])) / max(abs(float(oloss[k])), 1e-12)
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        if k not in params or params[k].grad is None:
            continue
        assert p.grad is not None, k
        r = float((p.grad.cpu() - params[k].grad).norm() / params[k].grad.norm().clamp_min(1e-30))
        if r > worst[1]:
            worst = (k, r)
    stats["worst gradient relL2"] = worst[1]
    for k, v in stats.items():
        print(f"PARITY config0 shape fp32 vs oracle: {k} {v:.3e}" + (f" ({worst[0]})" if k.startswith("worst") else ""))
    # measured: outputs 2e-4 / 4e-5 / ~5e-3 of their largest value (a handful of depths slide inside their reference bracket: z_close above),
    # every loss term <= 2e-4, the worst gradient (the colour table, through those depths) 1.7 %
    assert stats["rgb_values"] < 2e-3 and stats["depth_values"] < 2e-3 and stats["normal_map"] < 2e-2, stats
    assert all(v < 2e-3 for k, v in stats.items() if k.startswith("loss.")), stats
    assert stats["worst gradient relL2"] < 5e-2, (worst, stats)
    # ---- bf16 operands at this shape (the L = 16 fused kernels do not apply: the eager path with its library products), loss where fp32's is
    from holoscene_amd.training.trainer import Stage1Trainer, stock_conf
    tr = Stage1Trainer(stock_conf(num_rays=R, S=64, d_out=2, num_levels=8, end_size=256, logmap=15, beta=0.05, mlp_precision="bf16", use_bg_reg=False),
                       device=DEV, optimizer="flat", graph=False, freeze_parameters=True)
    tr.model.load_state_dict(sd)
    tr.iter_step = 1
    _, lb = tr.train_step(torch.tensor([0]), {"uv": uv.to(DEV), "pose": pose.to(DEV), "intrinsics": K.to(DEV)}, dv(gt), rng=dv(rand))
    rel = abs(float(lb["loss"]) - float(oloss["loss"])) / abs(float(oloss["loss"]))
    print(f"PARITY config0 shape bf16 vs oracle: loss rel {rel:.3e}")
    assert rel < 3e-2, rel


@pytest.mark.parametrize("L", [8, 13, 5])
def test_grid_of_fewer_levels_runs_on_the_fused_sixteen_level_kernels(L, monkeypatch):
    """A hash grid of FEWER than 16 levels (BASELINE configs[0]: L = 8) on the hand-written bf16 kernels: the hash kernels are handed empty levels
    behind the grid's own -- also an ODD count, and one that ends inside a lane half's eight levels (L = 13, 5) -- (HashEncoder.fused_offsets -> zeros, nothing scattered), the layers that read hash features zero columns (fused_cols).
    (1) hash kernels: the padded call = the plain call on the real levels, zeros elsewhere, the same table gradient; (2) the model takes the
    fused kernels (counted) and the whole-iteration graph; (3) its losses and gradients agree with the fp32 library path of the same state as
    the stock grid's bf16 path does with its own."""
    import numpy as np
    from holoscene_amd.hashencoder import backend as be_mod
    from holoscene_amd.hashencoder.hashgrid import HashEncoder
    from holoscene_amd.training.trainer import Stage1Trainer, stock_conf
    be = be_mod._backend
    torch.manual_seed(3)
    # ---- (1)
    enc = HashEncoder(3, L, 2, 2, 16, 15, 256).to(DEV)
    with torch.no_grad():
        enc.embeddings.uniform_(-0.5, 0.5)
    B = 5000
    x = torch.rand(B, 3, device=DEV)
    S, H = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
    f8, d8 = torch.empty(L, B, 2, device=DEV), torch.empty(L, B, 6, device=DEV)
    f16, d16 = torch.full((16, B, 2), 7.0, device=DEV), torch.full((16, B, 6), 7.0, device=DEV)
    be.fwd(x, enc.embeddings, enc.offsets, f8, B, 3, 2, L, S, H, d8, level_major=True)
    be.fwd(x, enc.embeddings, enc.fused_offsets, f16, B, 3, 2, 16, S, H, d16, level_major=True)
    assert torch.equal(f16[:L], f8) and torch.equal(d16[:L], d8) and not f16[L:].any() and not d16[L:].any()
    w16 = torch.empty(16, B, device=DEV, dtype=torch.int32)
    be.fwd(x, enc.embeddings, enc.fused_offsets, w16, B, 3, 2, 16, S, H, None, level_major=True, out_bf16=True)
    assert not w16[L:].any() and w16[:L].any()
    g8, gj8 = torch.randn(L, B, 2, device=DEV), torch.randn(L, B, 6, device=DEV)
    g16, gj16 = torch.randn(16, B, 2, device=DEV), torch.randn(16, B, 6, device=DEV)
    g16[:L], gj16[:L] = g8, gj8
    t8, t16 = torch.zeros_like(enc.embeddings), torch.zeros_like(enc.embeddings)
    be.bwd_jac(g8, gj8, x, enc.offsets, t8, B, 3, 2, L, S, H, level_major=True)
    be.bwd_jac(g16, gj16, x, enc.fused_offsets, t16, B, 3, 2, 16, S, H, level_major=True)
    assert float((t16 - t8).abs().max()) <= 1e-5 * float(t8.abs().max())       # (float atomics: order)
    # ---- (2), (3)
    from holoscene_amd.training.synthetic import look_at_pose
    from holoscene_amd.training.trainer import benchmark_model_state
    R, res = 256, 512
    conf = lambda prec: stock_conf(num_rays=R, S=64, d_out=2, num_levels=L, end_size=256, logmap=15, beta=0.05, mlp_precision=prec, use_bg_reg=False)  # noqa: E731
    tr = Stage1Trainer(conf("bf16"), device=DEV, optimizer="flat", graph=True, freeze_parameters=True, inject_draws=True)
    net = tr.model.implicit_network
    assert net.fused_trunk_blockers() == [] and tr.model.fused_path_report() == [] and tr._full_graph_ok()
    benchmark_model_state(tr.model, 0.05)
    with torch.no_grad():       # a SMOOTH field: the sampler's depths are then the same in both precisions and the gradients comparable
        for e in (net.encoding, net.color_encoding):
            e.embeddings.uniform_(-1e-2, 1e-2)
    ref = Stage1Trainer(conf("fp32"), device=DEV, optimizer="flat", graph=False, freeze_parameters=True)
    ref.model.load_state_dict(tr.model.state_dict())
    g = torch.Generator().manual_seed(6)
    uv = torch.randint(0, res, (1, R, 2), generator=g).float()
    pose = look_at_pose((0.7, 0.0, 0.0))[None]
    Km = torch.eye(4)[None].clone()
    Km[0, 0, 0] = Km[0, 1, 1] = Km[0, 0, 2] = Km[0, 1, 2] = res / 2
    rand = {"ray_offset": torch.rand(1, R, 2, generator=g) - 0.5, "t_rand": torch.rand(R, 64, generator=g), "u_final": torch.rand(R, 32, generator=g),
            "perm": torch.randperm(64 * 5, generator=g), "eik_idx": torch.randint(0, 50, (R,), generator=g),
            "eik_uniform": torch.rand(R, 3, generator=g) * 2 - 1, "eik_jitter": torch.rand(2 * R, 3, generator=g)}
    gt = {"rgb": torch.rand(1, R, 3, generator=g), "depth": torch.rand(1, R, 1, generator=g) * 0.9 + 0.1,
          "normal": torch.nn.functional.normalize(torch.randn(1, R, 3, generator=g), dim=-1), "mask": torch.ones(1, R, 1),
          "segs": torch.randint(0, 2, (1, R, 1), generator=g)}
    dv = lambda d: {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()}  # noqa: E731
    mi = {"uv": uv.to(DEV), "pose": pose.to(DEV), "intrinsics": Km.to(DEV)}
    calls = {}
    names = ("sdf_mlp2_fwd", "trunk_rr_fwd", "trunk_mlp2_fwd", "wgrad_pairs")

    def wrap(n):
        o = getattr(be, n)

        def counted(*a, **k):
            calls[n] = calls.get(n, 0) + 1
            return o(*a, **k)
        return counted
    for n in names:
        monkeypatch.setattr(be, n, wrap(n))
    tr.iter_step = ref.iter_step = 1
    _, lb = tr.train_step(torch.tensor([0]), mi, dv(gt), rng=dv(rand))
    monkeypatch.undo()
    assert calls.get("sdf_mlp2_fwd", 0) >= 1 and calls.get("trunk_rr_fwd", 0) >= 1 and calls.get("wgrad_pairs", 0) >= 1, calls
    assert any(k[0] == "full" for k in tr._graphs), list(tr._graphs)
    _, lf = ref.train_step(torch.tensor([0]), mi, dv(gt), rng=dv(rand))
    for k in ("loss", "rgb_loss", "eikonal_loss", "depth_loss"):
        rel = abs(float(lb[k]) - float(lf[k])) / max(abs(float(lf[k])), 1e-12)
        print(f"PARITY L = {L} grid on the fused kernels (whole-iteration graph), bf16 vs fp32 library path: {k} rel {rel:.3e}")
        assert rel < 3e-2, (k, rel)
    # ---- (4) the padding machinery on its own: the same network written as a SIXTEEN-level grid whose levels 8..15 hold zero tables (same
    # level scales: end = base x pls^15) and whose feature-reading layers have zero columns for them runs the stock path with no padding at all
    # -- same features, same products: losses and every shared gradient must agree to rounding, not to bf16
    pls = float(net.encoding.per_level_scale)
    big = Stage1Trainer(stock_conf(num_rays=R, S=64, d_out=2, num_levels=16, end_size=16.0 * pls ** 15, logmap=15, beta=0.05, mlp_precision="bf16", use_bg_reg=False),
                        device=DEV, optimizer="flat", graph=True, freeze_parameters=True, inject_draws=True)
    bn = big.model.implicit_network
    assert abs(float(bn.encoding.per_level_scale) - pls) < 1e-6 * pls and getattr(bn.lin0, "fused_cols", None) is None
    n8 = int(net.encoding.offsets[-1])
    assert torch.equal(bn.encoding.offsets[:L + 1].cpu(), net.encoding.offsets.cpu())
    sd8 = tr.model.state_dict()
    with torch.no_grad():
        sdb = big.model.state_dict()
        for k, v in sd8.items():
            if v.shape == sdb[k].shape:
                sdb[k].copy_(v)
        for e8, e16 in ((net.encoding, bn.encoding), (net.color_encoding, bn.color_encoding)):
            e16.embeddings.zero_()
            e16.embeddings[:n8].copy_(e8.embeddings)
        for name, cols in (("lin0.weight_v", 39 + 2 * L), ("color_grid_feature_map_mlp.0.weight", 2 * L)):
            w8, w16 = dict(net.named_parameters())[name], dict(bn.named_parameters())[name]
            w16.zero_()
            w16[:, :cols].copy_(w8)
        bn.lin0.weight_g.copy_(net.lin0.weight_g)          # (row norms of v are unchanged by zero columns)
    big.iter_step = 1
    _, l16 = big.train_step(torch.tensor([0]), mi, dv(gt), rng=dv(rand))
    for k in ("loss", "rgb_loss", "eikonal_loss", "depth_loss", "normal_l1"):
        rel = abs(float(lb[k]) - float(l16[k])) / max(abs(float(l16[k])), 1e-12)
        print(f"PARITY L = {L} grid padded to the fused kernels vs the same network as an explicit 16-level grid: {k} rel {rel:.3e}")
        assert rel < 1e-5, (k, rel)
    gb = dict(tr.model.named_parameters())
    g16 = dict(big.model.named_parameters())
    worst = ("", 0.0)
    for k, p in gb.items():
        a, b = p.grad, g16[k].grad
        assert a is not None and b is not None, k
        if k.endswith("encoding.embeddings"):
            b = b[:n8]
        elif k.endswith("lin0.weight_v") and "implicit" in k:
            assert not b[:, 39 + 2 * L:].any(), "zero features give zero weight gradients"
            b = b[:, :39 + 2 * L]
        elif k.endswith("color_grid_feature_map_mlp.0.weight"):
            assert not b[:, 2 * L:].any()
            b = b[:, :2 * L]
        rel = float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))
        if rel > worst[1]:
            worst = (k, rel)
    print(f"PARITY L = {L} grid padded vs explicit 16-level grid: worst gradient relL2 {worst[1]:.3e} ({worst[0]})")
    assert worst[1] < 1e-4, worst       # (table scatters: float-atomic order)
    assert gb["implicit_network.lin0.weight_v"].grad.shape == (256, 39 + 2 * L)


@pytest.mark.parametrize("d_out,levels,logmap,end_size,B", [(32, 16, 19, 2048, 131072), (21, 16, 15, 512, 1000), (64, 16, 19, 2048, 65536 + 33), (40, 16, 15, 512, 65),
                                                            (3, 8, 15, 256, 4099)])
def test_sweep_with_the_gather_inside_is_bit_identical_to_gather_plus_trunk(d_out, levels, logmap, end_size, B, monkeypatch):
    """hs_sdf_sweep_fwd (k_sdf_mlp2<., true>: every lane gathers the eight hash levels of its own point) against the two launches it replaces --
    hs_hash_fwd(out_bf16) + hs_sdf_mlp2_fwd / _wide -- through the sampler's entry point: bit for bit, for the minimum over all objects, one
    object, a subset; stock grid (dense levels 0-4 with tables that are not powers of two, hashed levels above), a grid of fewer levels (empty
    levels behind its own), points outside the cube, on its faces (x = 1: the reference's modulo wrap on integer-scale levels) and a ragged tail."""
    from holoscene_amd.hashencoder import backend as be_mod
    from holoscene_amd.model import network as N
    torch.manual_seed(d_out + levels)
    net = N.ObjectImplicitNetworkGrid(256, 1.0, d_in=3, d_out=d_out, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], multires=6,
                                      divide_factor=1.0, sigmoid=10, color_grid_feature=True, num_levels=levels, logmap=logmap, end_size=end_size).to(DEV)
    with torch.no_grad():
        net.lin0.weight_v[:, 3:].normal_(0, 1e-2)
        net.encoding.embeddings.uniform_(-0.5, 0.5)
        net.lin2.weight_v.add_(0.05 * torch.randn_like(net.lin2.weight_v))
        net.lin2.bias.add_(0.1 * torch.randn_like(net.lin2.bias))
    net.set_mlp_precision("bf16")
    x = (torch.rand(B, 3, device=DEV) * 2.4 - 1.2)
    x[:64] = (torch.randint(0, 2, (64, 3), device=DEV).float() * 2 - 1)         # cube corners: x01 in {0, 1} exactly
    x[64:128, 0] = 1.0                                                           # a face
    x01 = ((x / net.divide_factor + 1.0) / 2.0).contiguous()
    calls = []
    orig = be_mod._backend.sdf_sweep_fwd
    monkeypatch.setattr(be_mod._backend, "sdf_sweep_fwd", staticmethod(lambda *a, **k: calls.append(1) or orig(*a, **k)))
    a, b = torch.tensor([1.0], device=DEV), torch.tensor([2.0], device=DEV)
    sels = [-1, d_out - 1, 0, [0, d_out - 1] + ([31, 32] if d_out > 33 else [])]

    def queries():
        net.invalidate_packed_weights()
        with torch.no_grad():
            return [net.sdf_at_points(x, x01, 1, B, sel, gate=(b, a)).clone() for sel in sels]
    monkeypatch.setattr(N.ops, "SDF_SWEEP_FUSED", True)
    fused = queries()
    assert len(calls) == len(sels)
    monkeypatch.setattr(N.ops, "SDF_SWEEP_FUSED", False)
    pair = queries()
    assert len(calls) == len(sels)
    inside = ((x01 >= 0) & (x01 <= 1)).all(-1).float().mean().item()
    for sel, f, p in zip(sels, fused, pair):
        same = torch.equal(f, p)
        print(f"PARITY fused sweep K={d_out} L={levels} B={B} select={sel}: bit-identical to gather + trunk: {same} ({inside:.2f} of the points inside the cube)")
        assert same, (sel, float((f - p).abs().max()))
    # a closed gate leaves the output untouched
    out = net.sdf_at_points(x, x01, 1, B, -1, gate=(a, b))
    assert out.shape == (1, B)
