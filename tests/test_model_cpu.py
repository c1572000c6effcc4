"""CPU: host-side model logic (samplers, value+Jacobian trunk, compositing, loss, optimiser wiring) of
holoscene_amd against the reference-generated golden fixtures.  The hash encoder is served by the
CPU oracle through tests/oracle_backend.py (test-only); the GPU tests run the same checks on the HIP kernels."""
import pytest
import torch

import oracle_backend
from helpers import load, rand_dict, section
from model_helpers import build_model, build_loss, close, z_close


@pytest.fixture(autouse=True)
def _oracle_hash(monkeypatch):
    oracle_backend.install(monkeypatch)
    from holoscene_amd.model import ray_sampler
    monkeypatch.setattr(ray_sampler, "SAMPLER_IMPL", "torch")  # explicit opt-in: host-logic check of the whole-tensor formulation
    from holoscene_amd.model import network
    monkeypatch.setattr(network.ops, "COMPOSITE_IMPL", "torch")
    from holoscene_amd.model import loss
    monkeypatch.setattr(loss, "LOSS_IMPL", "torch")


@pytest.mark.parametrize("name", [f"sampler_{i}" for i in range(5)] + ["sampler_eval", "stock_sampler_0"])
def test_sampler(name):
    rec = load(name)
    model = build_model(rec)
    model.train(bool(rec["meta.train"]))
    ins = section(rec, "in.")
    z, z_eik = model.ray_sampler.get_z_vals(ins["ray_dirs"], ins["cam_loc"], model, rng=rand_dict(rec))
    assert model.ray_sampler.last_rounds == int(rec["meta.rounds"])
    z_close(z, torch.from_numpy(rec["out.z_vals"]))
    assert z_eik.shape == rec["out.z_samples_eik"].shape


def test_sampler_inverse_sphere_branch():
    """inverse_sphere_bg=True (ray_sampler.py:127-128, 262-265, 282-285; set by no conf of the reference): the far sample appended to the
    final set is each ray's exit from the bounding sphere, and the call returns (z_vals, z_vals_inverse_sphere) -- against the reference's own
    call on its draws.  A ray that misses the sphere ends the reference's process; here it raises with the same words."""
    rec = load("sampler_inv")
    model = build_model(rec).train()
    assert model.ray_sampler.inverse_sphere_bg
    ins = section(rec, "in.")
    (z, z_inv), z_eik = model.ray_sampler.get_z_vals(ins["ray_dirs"], ins["cam_loc"], model, rng=rand_dict(rec))
    assert model.ray_sampler.last_rounds == int(rec["meta.rounds"])
    z_close(z, torch.from_numpy(rec["out.z_vals"]))
    close(z_inv, rec["out.z_vals_inverse_sphere"], 1e-6, 1e-7, "z_vals_inverse_sphere")
    assert torch.equal(z_eik, torch.gather(z, 1, torch.from_numpy(rec["rand.eik_idx"])[:, None]))
    # the sphere exit is one of the final depths of every ray, the constant far bound (3.5) is not
    from holoscene_amd.utils.rend_util import get_sphere_intersections
    far = get_sphere_intersections(ins["cam_loc"], ins["ray_dirs"], r=1.0)[:, 1:]
    assert bool(((z - far).abs().min(dim=1)[0] == 0).all()) and float(z.max()) < 2.0
    with pytest.raises(RuntimeError, match="BOUNDING SPHERE PROBLEM"):
        model.ray_sampler.get_z_vals(ins["ray_dirs"], ins["cam_loc"] + torch.tensor([5.0, 0.0, 0.0]), model, rng=rand_dict(rec))


@pytest.mark.parametrize("name", ["iter_k3_bg", "iter_k5", "stock_k32_bg", "stock_k21", "stock_k40", "stock_k64_bg", "stock_l8_k3_bg"])
def test_iteration(name):
    rec = load(name)
    model = build_model(rec)
    model.train()
    loss_fn = build_loss()
    ins, gt = section(rec, "in."), section(rec, "gt.")
    out = model(ins, torch.tensor([0]), iter_step=int(rec["meta.iter_step"]), rng=rand_dict(rec))
    ref = section(rec, "out.")
    assert ("bg_depth_values" in out) == bool(rec["meta.has_bg"])
    z_close(out["z_vals"], ref["z_vals"])
    for k, v in ref.items():
        if k == "bg_mask":
            assert (out[k] == v).float().mean() > 0.99
        elif k not in ("z_vals",):
            close(out[k], v, 1e-3, 2e-4, k)
    out["iter_step"] = int(rec["meta.iter_step"])
    lo = loss_fn(out, gt, call_reg=bool(rec["meta.call_reg"]))
    for k, v in section(rec, "loss.").items():
        close(lo[k], v, 1e-3, 1e-5, "loss." + k)
    lo["loss"].backward()
    params = dict(model.named_parameters())
    for k, v in section(rec, "grad.").items():
        assert params[k].grad is not None, k
        close(params[k].grad, v, 5e-3, 2e-4 * max(1e-3, float(v.abs().max())), "grad." + k)
    from holoscene_amd.training.optim import build_optimizer
    opt = build_optimizer(model, lr=5e-4, lr_factor_for_grid=20.0)
    opt.step()
    for k, v in section(rec, "adam1.").items():
        close(params[k].detach(), v, 1e-4, 2e-5, "adam1." + k)


@pytest.mark.parametrize("name", ["multi_obj_k5", "multi_obj_k5_eval"])
def test_stage23_entry_points(name):
    """forward_multi_obj* / get_colors_* (network.py:1016-1801) against what the reference returned."""
    from model_helpers import check_multi_obj
    rec = load(name)
    model = build_model(rec)
    model.train(bool(rec["meta.train"]))
    check_multi_obj(model, rec)


@pytest.mark.parametrize("name", ["net_k2", "net_k21", "net_k32", "stock_net_k21", "stock_net_k32"])
def test_network_query_methods(name):
    """G3 / G5 of SURVEY 8c: ObjectImplicitNetworkGrid's query methods, RenderingNetwork.forward, volume_rendering and
    occlusion_opacity against direct calls of the reference's (K = 2, 21, 32; perturbed, object-distinct weights; points incl.
    the cube boundary and outside it)."""
    from model_helpers import check_network_methods
    rec = load(name)
    model = build_model(rec).eval()
    check_network_methods(model, rec)


def test_three_training_steps_match_reference():
    """G6 / G7 of SURVEY 8c: forward, loss, backward, Adam and the ExponentialLR schedule over three consecutive iterations with
    fresh batches and draws per step: losses, learning rates and every parameter after steps 1 and 3 vs the reference's run."""
    import numpy as np
    from model_helpers import run_three_steps
    rec = load("steps3_k3")
    model = build_model(rec).train()
    losses, lrs, snaps = run_three_steps(model, rec)
    for i, l in enumerate(losses):
        assert abs(l - float(rec[f"s{i}.loss"])) <= 2e-4 * abs(float(rec[f"s{i}.loss"])), (i, l, float(rec[f"s{i}.loss"]))
    assert np.allclose(np.array(lrs), rec["lr_after_step"], rtol=1e-12, atol=0)
    for n_steps in (1, 3):
        for k, v in section(rec, f"adam{n_steps}.").items():
            close(snaps[n_steps][k], v, 2e-4, 4e-5 * n_steps, f"adam{n_steps}.{k}")


def test_state_dict_keys_match_reference():
    rec = load("iter_k5")
    model = build_model(rec)
    assert sorted(model.state_dict().keys()) == sorted(section(rec, "state.").keys())
    for k, v in section(rec, "state.").items():
        assert tuple(model.state_dict()[k].shape) == tuple(v.shape), k


def test_reference_style_double_backward_matches_jacobian_path():
    """HashEncoder used the reference's way (autograd.grad with create_graph, then backward through it)
    must agree with the value+Jacobian path on both d sdf/dx and the parameter gradients."""
    rec = load("iter_k5")
    m1, m2 = build_model(rec), build_model(rec)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(64, 3, generator=g) * 1.6 - 0.8
    # reference style
    xa = x.clone().requires_grad_(True)
    y = m1.implicit_network(xa)[:, :m1.implicit_network.d_out]
    sdf = y.min(-1, keepdim=True)[0]
    (ga,) = torch.autograd.grad(sdf, xa, torch.ones_like(sdf), create_graph=True)
    ((ga.norm(dim=-1) - 1) ** 2).mean().backward()
    # jacobian style
    sdf2, _, gb, _, _ = m2.implicit_network.get_outputs(x)
    ((gb.norm(dim=-1) - 1) ** 2).mean().backward()
    close(gb, ga.detach(), 1e-4, 1e-5, "d sdf/dx")
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for k in p1:
        if p1[k].grad is None:
            continue
        if p2[k].grad is None:  # colour branch: evaluated (with zero gradient) only by the reference-style forward
            assert float(p1[k].grad.abs().max()) == 0.0, k
        else:
            close(p2[k].grad, p1[k].grad, 2e-3, 1e-4 * float(p1[k].grad.abs().max()), k)


def test_fused_sampler_refuses_cpu_tensors(monkeypatch):
    from holoscene_amd.model import ray_sampler
    monkeypatch.setattr(ray_sampler, "SAMPLER_IMPL", "hip")
    rec = load("sampler_0")
    model = build_model(rec)
    ins = section(rec, "in.")
    with pytest.raises(RuntimeError, match="CUDA tensors"):
        model.ray_sampler.get_z_vals(ins["ray_dirs"], ins["cam_loc"], model)


@pytest.mark.parametrize("name", ["object_sdf_fg", "object_sdf_bg"])
def test_per_object_networks_match_reference(name):
    """SingleObjectImplicitNetworkGrid / SingleObjectRenderingNetwork / ObjectSDFNetwork (network.py:1835-2209) --
    query methods, forward on the reference's rays and draws, parameter gradients of a fixed cotangent; an object (fg) and a
    background (bg) initialisation."""
    from object_helpers import check_object_model
    check_object_model(load(name), "cpu", strict=True)


@pytest.mark.parametrize("i", [0, 3, 7])
def test_eight_object_fixture_members_match_reference_one_by_one(i):
    """tests/golden/object_set8.npz: eight of the reference's ObjectSDFNetwork.forward, each with its own centre, scale, initialisation, rays
    and draws (what ObjectSDFNetworkSet evaluates together on the GPU, tests/test_model_gpu.py): a member through the single-object class."""
    from object_helpers import check_object_model, sub_record
    check_object_model(sub_record(load("object_set8"), i), "cpu", strict=True, queries=False)


# ------------------------------------------------------------------------------------ host logic added in round 3
def test_pair_slices_fill_the_chip_in_proportion_to_bytes():
    """_pair_slices: the jobs of one hs_wgrad_pairs launch get workgroups in proportion to their (costed) bytes, the counts add up to the budget,
    no job gets more slices than it has tiles, and no slice is more than ~15 % above the mean load."""
    import math
    from holoscene_amd.model.fused_ops import _PAIR_TILE_BYTES, _pair_slices
    for T, Te, npair in [(3136, 512, 2), (3136, 512, 1), (12288, 1024, 2), (40, 8, 2), (3, 1, 2)]:
        jobs = [((256, 256), T, npair, True), ((256, 80), T, npair, True), ((32, 256), T, npair, True), ((256, 80), T, 1, False),
                ((256, 256, "rm"), Te, 1, True), ((256, 80, "rm"), Te, 1, True)]
        cut = _pair_slices(jobs)
        assert all(1 <= c <= j[1] for c, j in zip(cut, jobs))
        total_tiles = sum(j[1] for j in jobs)
        assert sum(cut) == min(256, total_tiles)
        if T >= 3136:
            from holoscene_amd.model.fused_ops import _PAIR_REG_COST        # row-major-only jobs: the slower register-staged form, costed per byte
            per = [(_PAIR_TILE_BYTES[tuple(j[0][:2])] if j[3] else 64 * j[0][0]) * j[2] * (_PAIR_REG_COST if "rm" in j[0] else 1.0) for j in jobs]
            load = [math.ceil(j[1] / c) * p for j, c, p in zip(jobs, cut, per)]
            mean = sum(j[1] * p for j, p in zip(jobs, per)) / 256
            assert max(load) <= 1.15 * mean, (cut, load, mean)


def test_render_outputs_defer_entries_until_read():
    from holoscene_amd.model.network import _Outputs
    calls = []
    o = _Outputs({"a": 1})
    o.defer("b", lambda: calls.append(1) or 2)
    assert "b" in o and not calls                  # membership does not evaluate
    assert o["a"] == 1 and not calls
    assert o["b"] == 2 and calls == [1] and o["b"] == 2 and calls == [1]      # once
    o.defer("c", lambda: 3)
    assert sorted(o.keys()) == ["a", "b", "c"] and dict(o) == {"a": 1, "b": 2, "c": 3} and len(o) == 3
    o.defer("d", lambda: 4)
    assert o.pop("d") == 4 and "d" not in o         # pop reads like any other access (dict semantics), then the entry is gone
    assert o.pop("d", 9) == 9
    assert o.get("zz", 7) == 7
    with pytest.raises(KeyError):
        o["zz"]


def test_off_fused_path_is_reported_with_its_reason():
    """A conf whose shapes the hand-written bf16 kernels do not cover must SAY so: fused_path_report names the
    reason, render() warns once per model in training.  Stock shape: nothing to report."""
    from holoscene_amd.model.network import HoloSceneNetwork
    from holoscene_amd.training.trainer import stock_conf
    stock = HoloSceneNetwork(stock_conf(num_rays=8, S=8, d_out=32, mlp_precision="bf16", logmap=8, end_size=64).get_config("model"))
    assert stock.implicit_network.fused_trunk_blockers() == [] and stock.fused_path_report() == []
    k40 = HoloSceneNetwork(stock_conf(num_rays=8, S=8, d_out=40, mlp_precision="bf16", logmap=8, end_size=64).get_config("model"))
    rep = k40.fused_path_report()
    assert k40.implicit_network.fused_trunk_blockers() == [] and len(rep) == 1 and "d_out = 40 > 32" in rep[0]
    # (a grid of FEWER than 16 levels x 2 channels is on the fused path since round 5: empty levels behind its own, HashEncoder.fused_offsets)
    l8ok = HoloSceneNetwork(stock_conf(num_rays=8, S=8, d_out=2, num_levels=8, mlp_precision="bf16", logmap=8, end_size=64).get_config("model"))
    assert l8ok.implicit_network.fused_trunk_blockers() == [] and l8ok.implicit_network.lin0.fused_cols == 71
    assert l8ok.implicit_network.encoding.fused_offsets.shape[0] == 17 and l8ok.implicit_network.encoding.offsets.shape[0] == 9
    c20 = stock_conf(num_rays=8, S=8, d_out=2, num_levels=20, mlp_precision="bf16", logmap=8, end_size=64)
    l8 = HoloSceneNetwork(c20.get_config("model"))
    assert any("hash grid 20 levels" in w for w in l8.implicit_network.fused_trunk_blockers())
    assert "library GEMMs" in l8.fused_path_report()[0]
    with pytest.warns(UserWarning, match="does not run on the benchmarked bf16 kernels"):
        l8._warn_off_fused_path()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        l8._warn_off_fused_path()           # once per model
        stock._warn_off_fused_path()        # nothing to say
    fp32 = HoloSceneNetwork(stock_conf(num_rays=8, S=8, d_out=2, num_levels=20, mlp_precision="fp32", logmap=8, end_size=64).get_config("model"))
    assert fp32.fused_path_report() == []   # the reference's precision is a choice, not a fallback
