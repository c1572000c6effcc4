"""GPU: the benchmarked path against the reference AT THE BENCHMARKED SIZE.

tests/golden/full_c1.npz is ONE training iteration of BASELINE configs[1] -- 1 024 rays x 128 samples (98 rendered points per ray),
K = 32, 16 levels 16 -> 2048 with the full T = 2^19 tables, background-patch pass and collision term on, 5 sampler rounds -- run by the
imported reference on the CPU (tests/golden/make_golden.py::run_iteration(shape="full")).  The two 48.8 MB tables are regenerated
from their seeds and verified against stored digests; table-sized results (gradients, parameters after Adam) are stored as digests
(every 97th row + per-level sums of g and g^2, helpers.table_digest); everything else -- all outputs of forward, every loss term, every
MLP gradient, every random draw -- is stored in full.

What this pins that the small fixtures cannot: the multi-workgroup machinery bench.py actually runs -- the ~8 700-workgroup binned
table scatter with its record lists, the XCD-affine level schedule, the persistent tile loops of the MFMA kernels over 100 352 points
(417 792 value+Jacobian rows), the 4-waves-per-ray sampler at 640 sections, the flat gradient buffer at 24.7 M parameters.
"""
import numpy as np
import pytest
import torch

from helpers import TABLE_KEYS, TABLE_SAMPLE_STRIDE, digest_sections, load_full, rand_dict, section
from model_helpers import build_model, build_loss
from test_stock_gpu import (BF16_GRAD_REL_L2, BF16_ITER_TOL, BF16_LOSS_RTOL, PER_SAMPLE, Checker, _dev, _graph_trainer, _reference_depths,
                            _report, q_err, rel_l2, rel_max)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def plain(sec):
    return {k: v for k, v in sec.items() if "#" not in k}


def check_table(chk, what, t, dg, offsets, lim_l2, lim_level):
    """A table-sized tensor against its digest: relative L2 over the sampled rows, and every level's L2 norm."""
    t = t.detach().cpu()
    chk(f"{what} sampled rows relL2", rel_l2(t[::TABLE_SAMPLE_STRIDE], torch.from_numpy(dg["vals"])), lim_l2)
    td = t.double()
    worst = 0.0
    for lv, (a, b) in enumerate(zip(offsets[:-1], offsets[1:])):
        ref = float(np.sqrt(dg["level_sq"][lv]))
        got = float(td[int(a):int(b)].pow(2).sum().sqrt())
        if ref > 0:
            worst = max(worst, abs(got - ref) / ref)
        else:
            assert got == 0.0, (what, lv)
    chk(f"{what} worst level-norm rel", worst, lim_level)


_LOADED = {}


@pytest.fixture(scope="module", params=["full_c1", "full_c4"])
def rec_c1(request):
    """full_c1: BASELINE configs[1] (1 024 rays x 128 samples, background patch + collision term).  full_c4: configs[4]'s shape on one GPU --
    2 048 rays x 192 samples (146 rendered points per ray, 299 008 samples; analytic Eikonal set, which is what the reference computes:
    SURVEY D1) -- a regular iteration (no background patch).  Both K = 32, L = 16, T = 2^19."""
    if request.param not in _LOADED:
        _LOADED.clear()                         # one 100 MB record at a time
        _LOADED[request.param] = load_full(request.param)
    rec = _LOADED[request.param]
    rec["name"] = request.param
    return rec


def test_full_size_fixture_is_the_benchmarked_configuration(rec_c1):
    m = {k[5:]: int(v) for k, v in rec_c1.items() if k.startswith("meta.") and np.ndim(v) == 0 and k != "meta.emb_scale"}
    assert (m["K"], m["L"], m["logmap"], m["base"], m["end"]) == (32, 16, 19, 16, 2048)
    if rec_c1["name"] == "full_c1":
        assert (m["R"], m["S"]) == (1024, 128) and m["rounds"] == 5 and m["has_bg"] == 1 and m["call_reg"] == 1
        assert rec_c1["out.z_vals"].shape == (1024, 98)
    else:
        assert (m["R"], m["S"]) == (2048, 192) and m["rounds"] >= 1 and rec_c1["out.z_vals"].shape == (2048, 146)
    assert rec_c1[f"state.{TABLE_KEYS[0]}"].shape == (6098108, 2)


def test_full_size_fp32_iteration_on_reference_depths(rec_c1):
    """The reference's own precision at configs[1]'s size, reference depths injected: every output, loss term and
    parameter gradient at SURVEY 8(d)'s fp32 tolerances; the product's own sampler (host-controlled in fp32) compared by z_close."""
    rec = rec_c1
    model = build_model(rec, DEV).train()
    ins, gt = _dev(section(rec, "in.")), _dev(section(rec, "gt."))
    dep = _dev(_reference_depths(rec))
    sm = model.ray_sampler
    orig = sm.get_z_vals
    own = []

    def on_reference_depths(d, o, m, idx=None, **k):
        own.append(orig(d, o, m, idx=idx, **k))
        return (dep["z_vals"], dep["z_eik"]) if idx is None else (dep["bg_z"], None)
    sm.get_z_vals = on_reference_depths
    out = model(ins, torch.tensor([0]), iter_step=int(rec["meta.iter_step"]), rng=_dev(rand_dict(rec)))
    chk = Checker(f"{rec['name']} fp32")
    # the product's own fp32 sampler against the reference's depths.  This fixture's SDF is deliberately rough (table noise of 2e-2 on all
    # 16 levels, so that every level's gradient is exercised): many sections carry ~zero weight, and inverse-CDF placement inside such a
    # section follows the last bits of the SDF (measured: 88.5 % of the depths within 1e-5; the well-conditioned per-round quantities --
    # merged set, beta after the line search -- are pinned exactly by test_sampler_round_intermediates_vs_reference)
    zo, zr = own[0][0].cpu(), dep["z_vals"].cpu()
    chk("own sampler: fraction of depths off by more than 1e-5", float(((zo - zr).abs() > 1e-5 + 1e-5 * zr.abs()).float().mean()), 0.20)
    lo_b = torch.cat([zr[:, :1], zr[:, :-1]], 1) - 1e-4
    hi_b = torch.cat([zr[:, 1:], zr[:, -1:]], 1) + 1e-4
    chk("own sampler: fraction of depths outside their reference bracket", float((~((zo >= lo_b) & (zo <= hi_b))).float().mean()), 0.05)
    for k, v in section(rec, "out.").items():
        if k == "bg_mask":
            assert float((out[k].cpu() == v).float().mean()) > 0.99
            continue
        if k == "z_vals":
            continue
        # rendered values atol 1e-4.  Per-sample tensors at 100 352 points: a last-bit difference in two nearly equal
        # per-object SDFs hands a point to the other object (its gradient row changes discontinuously), hence the 99.9 % quantile for those
        if k in PER_SAMPLE:     # (gradient rows reach |g| ~ 3 on this rough SDF: the absolute bound scales with the tensor's magnitude)
            chk(f"out.{k} q999 abs", float(torch.quantile((out[k].detach().cpu().reshape(v.shape) - v).abs().flatten().double()[:4_000_000], 0.999)),
                1e-4 * max(1.0, float(v.abs().max())))
        else:
            chk(f"out.{k} abs", float((out[k].detach().cpu().reshape(v.shape) - v).abs().max()), 1e-4 * max(1.0, float(v.abs().max())))
    out["iter_step"] = int(rec["meta.iter_step"])
    lo = build_loss()(out, gt, call_reg=bool(rec["meta.call_reg"]))
    for k, v in section(rec, "loss.").items():
        chk(f"loss.{k} rel", abs(float(lo[k]) - float(v)) / max(abs(float(v)), 1e-12), 1e-4)
    lo["loss"].backward()
    params = dict(model.named_parameters())
    for k, v in plain(section(rec, "grad.")).items():
        g = params[k].grad
        assert g is not None and torch.isfinite(g).all(), k
        chk(f"grad.{k} relL2", rel_l2(g, v), FP32_GRAD_L2)
    for k, dg in digest_sections(rec, "grad.").items():
        check_table(chk, f"grad.{k}", params[k].grad, dg, rec["aux.offsets"], FP32_GRAD_L2, FP32_GRAD_L2)
    chk.done()


# fp32 gradients at full size: sums over 100 352 points (trunk, colour branch) with a handful of points on a derivative discontinuity
# (ReLU masks, arg-min hand-overs) that a 1e-7 forward difference can cross -- bound = ~3x the measured worst tensor
FP32_GRAD_L2 = 2e-3


@pytest.mark.parametrize("depths", ["reference", "own"])
def test_full_size_bf16_whole_iteration_graph_vs_reference(rec_c1, depths):
    """THE benchmarked configuration at THE benchmarked size: Stage1Trainer(graph=True), mlp_precision = bf16, one replay of the captured
    whole-iteration graph with the reference's draws in its static input block, against the reference's outputs, loss terms and
    parameter gradients -- same bounds as the stock-shape fixtures (tests/test_stock_gpu.py).  depths="own" also runs the graph's Adam
    node and compares the parameter update with the reference's first Adam step."""
    rec = rec_c1
    name = rec["name"]
    tr = _graph_trainer(rec, freeze=(depths == "reference"), add_objectvio_iter=0 if bool(rec["meta.call_reg"]) else 10 ** 9)
    tr.iter_step = int(rec["meta.iter_step"])
    ins, gt = _dev(section(rec, "in.")), _dev(section(rec, "gt."))
    dep = _dev(_reference_depths(rec)) if depths == "reference" else None
    start = {k: p.detach().cpu().clone() for k, p in tr.model.named_parameters()}
    out, lo = tr.train_step(torch.tensor([0]), ins, gt, rng=rand_dict(rec), depths=dep)
    torch.cuda.synchronize()
    assert ("full", bool(rec["meta.has_bg"]), bool(rec["meta.call_reg"])) in tr._graphs, "the step must have gone through the whole-iteration graph"
    # Algorithm 1 stops when max over the rays of beta <= beta0: ONE ray of 2 048 on the threshold decides whether another round runs, and a
    # bf16 SDF may put it on the other side (full_c4: the reference runs 2 rounds, its second for a handful of rays).  The depths a ray
    # that HAS converged gets from another round move little; the bounds below hold either way.
    rounds_got, rounds_ref = int(tr.model.ray_sampler.last_rounds), int(rec["meta.rounds"])
    _report(f"{name} graph sampler rounds (reference {rounds_ref})", rounds_got)
    assert abs(rounds_got - rounds_ref) <= (0 if name == "full_c1" else 1)
    same_rounds = rounds_got == rounds_ref
    if not same_rounds and depths == "own":
        pytest.skip("the bf16 sampler stopped a round earlier / later than the reference: its depths are another, equally valid sample set")
    ref = section(rec, "out.")
    zs = out["sampled"]["z_vals"].cpu()
    err = (zs - ref["z_vals"]).abs()
    _report(f"{name} graph sampler frac|dz|<1e-3", float((err < 1e-3).float().mean()))
    lo_b = torch.cat([ref["z_vals"][:, :1], ref["z_vals"][:, :-1]], 1) - 5e-3
    hi_b = torch.cat([ref["z_vals"][:, 1:], ref["z_vals"][:, -1:]], 1) + 5e-3
    inside = float(((zs >= lo_b) & (zs <= hi_b)).float().mean())
    _report(f"{name} graph sampler frac inside reference bracket", inside)
    # (rough SDF, see the fp32 test: measured 94 % inside the bracket widened by the bf16 SDF tolerance, 65 % within 1e-3; the stock-shape
    # fixtures, whose SDF is smoother, keep 100 % / 94-98 %)
    assert not same_rounds or (inside > 0.90 and float((err < 1e-3).float().mean()) > 0.55)
    chk = Checker(f"{name} graph/{depths}")
    params = dict(tr.model.named_parameters())
    offsets = rec["aux.offsets"]
    if depths == "own":
        for k in ("rgb_values", "depth_values", "normal_map", "object_opacity", "semantic_values"):
            chk(f"out.{k} relL2", rel_l2(out[k], ref[k]), 5 * BF16_ITER_TOL[k])
        for k, v in section(rec, "loss.").items():
            chk(f"loss.{k} rel", abs(float(lo[k]) - float(v)) / max(abs(float(v)), 1e-12), 10 * BF16_LOSS_RTOL[k])
        # the graph's Adam node: direction of the first update (eps = 1e-15 makes it ~lr sign(g) element by element).  These cosines are a SMOKE
        # bound, not a precision statement: the first Adam step is the sign of every gradient element, so the many near-zero elements of a
        # table flip with any perturbation (fp32 on another summation order already gives 0.98 after three steps) and 0.47-0.50 is what a
        # correct bf16 gradient on its OWN depths produces; a dropped term or a wrong sign convention shows as <= 0.2.  The precision
        # statement is test_full_size_bf16_gradients_with_aligned_discontinuities below (<= 1 % per tensor on aligned inputs).
        worst, worst_k = 1.0, None
        for k, v in plain(section(rec, "adam1.")).items():
            du, dr = (params[k].detach().cpu() - start[k]).double().flatten(), (v - start[k]).double().flatten()
            if float(dr.norm()) == 0:
                continue
            c = float(du @ dr / (du.norm() * dr.norm()).clamp_min(1e-30))
            _report(f"{name} cos(update) {k}", c)
            if c < worst:
                worst, worst_k = c, k
        for k, dg in digest_sections(rec, "adam1.").items():
            du = (params[k].detach().cpu()[::TABLE_SAMPLE_STRIDE] - start[k][::TABLE_SAMPLE_STRIDE]).double().flatten()
            dr = (torch.from_numpy(dg["vals"]) - start[k][::TABLE_SAMPLE_STRIDE]).double().flatten()
            c = float(du @ dr / (du.norm() * dr.norm()).clamp_min(1e-30))
            _report(f"{name} cos(update) {k} (sampled rows)", c)
            chk(f"1 - cos(update) {k}", 1.0 - c, 1.0 - 0.4)
        print(f"PARITY {name} worst MLP update: {worst_k}")
        chk("1 - cos(update) worst MLP tensor", 1.0 - worst, 1.0 - 0.55)      # measured 0.665 (colour MLP layer 0, whose gradient is 5 % off), others 0.81-1.0
        chk.done()
        return
    # Per-ray outputs are compared by their WORST ray: 1 024 rays here against 32 in the stock-shape fixtures, on a rougher SDF (K = 32
    # objects whose two smallest SDFs lie within the bf16 tolerance of each other at a fraction of the samples: the arg-min, and with it the
    # sample's normal, flips).  Bounds = the stock-shape ones x 1.6, each next to its measured value.
    worst_ray = {"semantic_values": 1.6e-2,     # 1.26e-2
                 "depth_values": 1.6e-2,        # 1.31e-2
                 "normal_map": 4e-2}            # 3.09e-2
    for k, v in ref.items():
        if k == "bg_mask":
            assert float((out[k].cpu() == v).float().mean()) > 0.98
            continue
        if k == "z_vals":
            assert torch.equal(out[k].cpu(), v)
            continue
        if k in PER_SAMPLE:
            chk(f"out.{k} q99", q_err(out[k].reshape(v.shape), v, 0.99), BF16_ITER_TOL[k])
        else:
            chk(f"out.{k}", rel_max(out[k].reshape(v.shape), v), worst_ray.get(k, BF16_ITER_TOL[k]))
    for k, v in section(rec, "loss.").items():
        chk(f"loss.{k} rel", abs(float(lo[k]) - float(v)) / max(abs(float(v)), 1e-12), BF16_LOSS_RTOL[k])
    for k, v in plain(section(rec, "grad.")).items():
        g = params[k].grad
        assert g is not None and torch.isfinite(g).all(), k
        kind = "beta" if k == "density.beta" else ("colour" if ("color" in k or k.startswith("rendering_network")) else "trunk")
        chk(f"grad.{k} relL2", rel_l2(g, v), BF16_GRAD_REL_L2[kind])
    # Table gradients ELEMENT by element: at T = 2^19 a fine-level entry is touched by one or two samples, so the per-sample error of the
    # bf16 cotangent chain (and every flipped arg-min / ReLU unit, which replaces a sample's whole contribution) shows undiluted -- the
    # 2^12 tables of the stock-shape fixtures average hundreds of samples per entry (1 % there).  Measured: geometry table 1.0e-1 over the
    # sampled rows with every level's NORM within 3.9 % (a missing term would move the norms); colour table 3.9e-2 / 0.7 %.  The same
    # scatter kernels on fp32 cotangents agree with the reference to 1e-4 (test above).
    for k, dg in digest_sections(rec, "grad.").items():
        check_table(chk, f"grad.{k}", params[k].grad, dg, offsets, 1.5e-1, 6e-2)
    chk.done()


# ------------------------------------------------------------------------------------------------ sampler depths: tolerance as an argument
def _depth_stats(z, ref, atol, bracket):
    """(fraction of depths off by more than atol (+ 1e-5 relative), fraction outside the neighbouring reference depths widened by `bracket`)"""
    z, ref = z.detach().cpu().double(), ref.detach().cpu().double()
    off = float(((z - ref).abs() > atol + 1e-5 * ref.abs()).double().mean())
    lo_b = torch.cat([ref[:, :1], ref[:, :-1]], 1) - bracket
    hi_b = torch.cat([ref[:, 1:], ref[:, -1:]], 1) + bracket
    return off, float((~((z >= lo_b) & (z <= hi_b))).double().mean())


def test_full_size_smooth_sdf_sampler_against_the_survey_tolerance():
    """`z_vals atol 1e-5 with identical injected randoms`.  On a SMOOTH SDF -- the reference's table initialisation (+-1e-4)
    behind the benchmark state's refilled lin0 columns, K = 32 distinct objects, configs[1]'s 1 024 rays x 128 samples, all 5 sampler rounds
    (fixture full_sampler_smooth: the imported reference's ErrorBoundSampler.get_z_vals on the CPU).  What can be demanded is set by how far
    the SAME algorithm moves on ANOTHER host: the CPU oracle reproduces this fixture bit for bit in the container that made it (the -m "not
    gpu" suite checks that), and on the GPU box's host -- other BLAS kernels, other summation order in the fp32 layers -- a measured ~1 % of its
    depths differ by more than 1e-5.  So: the product's fp32 sampler may miss 1e-5 on at most 1.5x the fraction the oracle itself misses here
    (floor: SURVEY's 0.1 %), stays inside the reference bracket (<= 1e-4 of the depths outside) and runs the reference's 5 rounds."""
    rec = load_full("full_sampler_smooth")
    assert int(rec["meta.rounds"]) == 5 and rec["out.z_vals"].shape == (1024, 98)
    model = build_model(rec, DEV).train()
    ins = _dev(section(rec, "in."))
    z, z_eik = model.ray_sampler.get_z_vals(ins["ray_dirs"], ins["cam_loc"], model, rng=_dev(rand_dict(rec)))
    assert model.ray_sampler.last_rounds == 5
    ref = torch.from_numpy(rec["out.z_vals"])
    off, outside = _depth_stats(z, ref, 1e-5, 1e-4)
    host_off, host_out = _depth_stats(_oracle_sampler(rec)[0], ref, 1e-5, 1e-4)
    _report("full_sampler_smooth oracle on this host vs the fixture: fraction off by more than 1e-5", host_off)
    _report("full_sampler_smooth oracle on this host vs the fixture: fraction outside bracket", host_out)
    chk = Checker("full_sampler_smooth fp32")
    chk("fraction of depths off by more than 1e-5", off, max(1e-3, 1.5 * host_off))
    chk("fraction of depths outside their reference bracket", outside, max(1e-4, 1.5 * host_out))
    chk("worst depth error", float((z.cpu() - ref).abs().max()), 1e-2)
    eik_idx = torch.from_numpy(rec["rand.eik_idx"]).to(DEV)
    assert torch.equal(z_eik, torch.gather(z, 1, eik_idx[:, None]))
    chk.done()


def _oracle_sampler(rec, sdf_hook=None):
    """Stage1Oracle.sample_z on the fixture's state and draws; sdf_hook(sdf [B, 1]) -> sdf perturbs every SDF query of the sampler."""
    from helpers import oracle_cfg
    from oracle.stage1_oracle import Stage1Oracle
    orc = Stage1Oracle(oracle_cfg(rec), section(rec, "state."))
    orc.training = True
    if sdf_hook is not None:
        plain_vals = orc.sdf_vals
        orc.sdf_vals = lambda x: sdf_hook(x, plain_vals(x))
    ins = section(rec, "in.")
    rand = rand_dict(rec)
    if "ray_dirs" in ins:
        d, o = ins["ray_dirs"], ins["cam_loc"]
    else:       # as Stage1Oracle.forward sets its rays up (network.py:785-792)
        dirs, loc = orc.camera_rays(ins["uv"] + rand["ray_offset"], ins["pose"], ins["intrinsics"])
        d, o = dirs.reshape(-1, 3), loc[:, None].repeat(1, dirs.shape[1], 1).reshape(-1, 3)
    with torch.no_grad():
        return orc.sample_z(d, o, rand)


def test_full_size_rough_sdf_sampler_deviation_is_conditioning():
    """full_c1's sampler depths miss SURVEY 8(d)'s 1e-5 for ~11 % of the entries in fp32 (and sit within 1e-3 for only 65 % in bf16).  The
    claim that this is the CONDITIONING of inverse-CDF sampling on a rough SDF (table noise of 2e-2 on all 16 levels), not a defect of the
    kernels, is tested here instead of asserted, in two steps on the same rays, draws and state:
      (1) how far does the ALGORITHM move when its SDF values move in their last bits?  The CPU oracle (bit-identical to the fixture in the
          container that made it) is re-run on this host -- another BLAS, another summation order in the fp32 layers -- and with every SDF
          query perturbed by one unit in the last place: measured 7 % of its depths off by more than 1e-5 either way;
      (2) do the product's SAMPLER KERNELS implement Algorithm 1?  The oracle is re-run with every SDF query answered by the PRODUCT's SDF
          evaluation (fp32 kernels, then the fused bf16 kernels) -- same values in, so every difference that remains is the sampler's own
          arithmetic (cumulative sums, the line search, the inverse CDF) -- and must agree with the product's depths far more closely than
          the product agrees with the reference (measured: a third / a ninth of that deviation), at the level step (1) predicts for
          last-bit differences."""
    rec = load_full("full_c1")
    ref = torch.from_numpy(rec["out.z_vals"])
    chk = Checker("full_c1 sampler conditioning")
    base_off, base_out = _depth_stats(_oracle_sampler(rec)[0], ref, 1e-5, 1e-4)
    _report("full_c1 oracle on this host vs the fixture: fraction off by more than 1e-5", base_off)
    _report("full_c1 oracle on this host vs the fixture: fraction outside bracket", base_out)
    g = torch.Generator().manual_seed(5)

    def one_ulp(x, s):
        up = torch.rand(s.shape, generator=g) < 0.5
        return torch.where(up, torch.nextafter(s, torch.full_like(s, float("inf"))), torch.nextafter(s, torch.full_like(s, float("-inf"))))
    ulp_off, ulp_out = _depth_stats(_oracle_sampler(rec, one_ulp)[0], ref, 1e-5, 1e-4)
    _report("full_c1 oracle with +-1 ulp SDF vs the fixture: fraction off by more than 1e-5", ulp_off)
    _report("full_c1 oracle with +-1 ulp SDF vs the fixture: fraction outside bracket", ulp_out)
    model = build_model(rec, DEV).train()
    net = model.implicit_network
    ins = _dev(section(rec, "in."))
    for prec, atol, br in (("fp32", 1e-5, 1e-4), ("bf16", 1e-3, 5e-3)):
        if prec == "bf16":
            net.set_mlp_precision("bf16")
            model.rendering_network.set_mlp_precision("bf16")
        with torch.no_grad():
            rays = model.prepare_rays(ins, _dev(rand_dict(rec)))
            z_prod, _ = model.sample(rays, _dev(rand_dict(rec)))
            z_orc = _oracle_sampler(rec, lambda x, s: net.get_sdf_vals(x.to(DEV).contiguous()).cpu())[0]
        ref_off, ref_out = _depth_stats(z_prod, ref, atol, br)
        own_off, own_out = _depth_stats(z_prod, z_orc, atol, br)
        _report(f"full_c1 product {prec} sampler vs the fixture: fraction off by more than {atol:g}", ref_off)
        _report(f"full_c1 product {prec} sampler vs the fixture: fraction outside bracket", ref_out)
        # (2), measured: fp32 3.9 % off by more than 1e-5 (0.12 % outside the bracket) where the product is 6.6 % (0.24 %) from the reference;
        # bf16 3.9 % off by more than 1e-3 (0.6 %) against 35 % (5.9 %).  What remains with identical SDF values in is the same conditioning acting
        # on last-bit differences INSIDE the sampler (device expf / the association of its cumulative sums): step (1) moves 7 % of the depths
        # with one ulp of SDF.  Bounds = 1.5x measured.
        chk(f"{prec} sampler kernels vs the oracle on the SAME SDF values: fraction off by more than {atol:g}", own_off, 0.055 if prec == "fp32" else 0.06)
        chk(f"{prec} sampler kernels vs the oracle on the SAME SDF values: fraction outside bracket", own_out, 2.5e-3 if prec == "fp32" else 1e-2)
        if prec == "fp32":      # with the SDF sweeps on csrc/sdf_mlp32.hip the product's fp32 depths are AS close to the reference's as the oracle's own
                                # are after one ulp of SDF perturbation (measured 6.6 % against 7.0 %; through library GEMMs it was 11.5 %)
            chk("fp32 product vs the fixture, relative to the oracle's own sensitivity (+-1 ulp SDF / another host)", ref_off / max(ulp_off, base_off, 1e-4), 1.5)
            chk("fp32 product vs the fixture, outside bracket, relative to the same (floor 1e-3)", ref_out / max(ulp_out, base_out, 1e-3), 2.0)
        else:
            chk(f"{prec}: ... relative to the product's deviation from the reference", own_off / max(ref_off, 1e-6), 0.5)
    chk.done()


# ------------------------------------------------------------------------------------------------ bf16 gradients with the discontinuities aligned
def _encode_relu_masks_wave(pre_list, B):
    """fp32 pre-activations [B, 256] of the colour branch's three ReLU layers -> the mask words k_appear2_fwd writes (appearance2.hip: relu_pair,
    store_mask): int32 [tiles, 3, 64 lanes, 4 words]; lane (h, row), neuron tile nd, register pair r cover neurons
    32 nd + 8 (r >> 1) + 4 h + 2 (r & 1) (+1): bit 8 (nd & 1) + r (+16) of word nd >> 1; SET = unit off (negative, or -0)."""
    tiles = (B + 31) // 32
    out = np.zeros((tiles, 3, 64, 4), dtype=np.uint32)
    for layer, pre in enumerate(pre_list):
        p = pre.detach().float().cpu()
        off = torch.zeros(tiles * 32, 256, dtype=torch.bool)
        off[:B] = torch.signbit(p)                  # bf16(x) keeps the sign of x, incl. -0; a value rounding to +-0 keeps it too
        off = off.view(tiles, 32, 256).numpy()
        for h in range(2):
            for nd in range(8):
                for r in range(8):
                    na = 32 * nd + 8 * (r >> 1) + 4 * h + 2 * (r & 1)
                    bit = 8 * (nd & 1) + r
                    out[:, layer, 32 * h:32 * h + 32, nd >> 1] |= (off[:, :, na].astype(np.uint32) << np.uint32(bit)) | \
                                                                  (off[:, :, na + 1].astype(np.uint32) << np.uint32(bit + 16))
    return torch.from_numpy(out.view(np.int32).reshape(-1).copy())


def test_full_size_bf16_gradients_with_aligned_discontinuities(monkeypatch):
    """What is left of the bf16 gradient error once the discontinuities are taken out.  At the benchmarked size the bf16 graph's MLP gradients
    sit 0.5-5 % (relL2) and its table gradients 4-10 % (element-wise) from the reference's (test above): bounds a dropped term of a few per
    cent would pass.  The diagnosis -- ReLU units of the colour branch and arg-min hand-overs of the K = 32 SDFs whose fp32 decision lies
    within the bf16 rounding error of the threshold switch a whole sample's contribution -- is made testable here, on full_c1's state, points
    (the reference's depths) and realistic cotangents, Function by Function against fp32 autograd through the same product modules:
      * colour branch (k_appear2_fwd / _bwd, k_wgrad_pairs, hs_assemble, colour-table scatter): the fp32 forward's ReLU signs, encoded in the
        kernel's own mask layout, replace the masks the bf16 forward wrote;
      * trunk (k_rr_fwd / k_rr_bwd_*, k_trunk_fwd2 / k_trunk_bwd, k_wgrad_pairs, value+Jacobian scatter): samples whose bf16 arg-min differs
        from the fp32 one get a zero cotangent on the outputs that depend on the arg-min (the minimum, its gradient) in BOTH runs.
    Required: every MLP tensor of the colour branch within 1 % relL2 (measured 0.17-0.40 %: 4-7 % on the kernel's own masks), every trunk tensor
    within 1.5 % (measured <= 1.02 %), every table level norm within 2 % (measured 0.2 %; element-wise the tables agree to 0.45 % / 0.53 % here,
    against 4 % / 10 % in the end-to-end comparison above -- that remainder is not in the backward kernels: it is the bf16 SDF error, 5e-3,
    entering the Laplace density through s / beta with beta = 0.01, i.e. the compositing weights the cotangents are built from)."""
    from holoscene_amd.model import network as N
    from holoscene_amd.hashencoder import backend as Bk
    rec = load_full("full_c1")
    model = build_model(rec, DEV).train()
    net, rn = model.implicit_network, model.rendering_network
    ins = _dev(section(rec, "in."))
    dep = _dev(_reference_depths(rec))
    rng = _dev(rand_dict(rec))
    captured = {}
    orig_rgb_at = model._rgb_at

    def grab(points_flat, dirs_flat, gradients, indices=None, x01=None):
        captured.update(pts=points_flat.detach(), dirs=dirs_flat.detach(), nrm=gradients.detach())
        rgb = orig_rgb_at(points_flat, dirs_flat, gradients, indices, x01)
        rgb.register_hook(lambda g: captured.__setitem__("g_rgb", g.detach().clone()))
        return rgb
    model._rgb_at = grab
    sm = model.ray_sampler
    sm.get_z_vals = lambda d, o, m, idx=None, **k: (dep["z_vals"], dep["z_eik"]) if idx is None else (dep["bg_z"], None)
    out = model(ins, torch.tensor([0]), iter_step=3, rng=rng)       # (a regular iteration: no background patch)
    out["iter_step"] = 3
    lo = build_loss()(out, _dev(section(rec, "gt.")), call_reg=False)
    lo["loss"].backward()
    model.zero_grad(set_to_none=True)
    pts, dirs, nrm0, cot = captured["pts"], captured["dirs"], captured["nrm"], captured["g_rgb"].reshape(-1, 3).contiguous()
    B = pts.shape[0]
    assert B == 1024 * 98
    chk = Checker("full_c1 aligned")
    rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))  # noqa: E731

    def level_norm_err(a, b, offsets):
        worst = 0.0
        for lv, (s_, e_) in enumerate(zip(offsets[:-1], offsets[1:])):
            nb = float(b[int(s_):int(e_)].double().norm())
            if nb > 0:
                worst = max(worst, abs(float(a[int(s_):int(e_)].double().norm()) - nb) / nb)
        return worst
    offsets = rec["aux.offsets"]
    # ---------------------------------------------------------------- colour branch
    mlp, enc = net.color_grid_feature_map_mlp, net.color_encoding
    params = [enc.embeddings, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias] + [p for l in (rn.lin0, rn.lin1, rn.lin2)
                                                                                          for p in (l.weight_v, l.weight_g, l.bias)]
    names = ["colour table", "c0.w", "c0.b", "c1.w", "c1.b"] + [f"r{i}.{n}" for i in range(3) for n in ("v", "g", "bias")]
    nrm = nrm0.clone().requires_grad_(True)
    with torch.no_grad():
        feat = enc(pts / net.divide_factor)
        a_hc = feat @ mlp[0].weight.t() + mlp[0].bias
        fv = torch.relu(a_hc) @ mlp[2].weight.t() + mlp[2].bias
        emb = rn.embedview_fn
        x = torch.cat([emb(pts), emb(dirs), emb(nrm0), fv], -1)
        a_r0 = x @ rn.lin0.weight.t() + rn.lin0.bias
        a_r1 = torch.relu(a_r0) @ rn.lin1.weight.t() + rn.lin1.bias
    words32 = _encode_relu_masks_wave([a_hc, a_r0, a_r1], B).to(DEV)
    rgb = rn(pts, nrm, dirs, net._color_features(pts))
    ref = [g.float() for g in torch.autograd.grad((rgb * cot).sum(), [nrm] + params)]
    net.set_mlp_precision("bf16")
    rn.set_mlp_precision("bf16")
    be = Bk._backend
    raw_bwd = be.appearance2_bwd
    flips = {}

    def with_fp32_masks(g_rgb, rgb_, normals, masks, *a, **k):
        flips["colour"] = float((masks.view(torch.int32) != words32).float().mean())
        return raw_bwd(g_rgb, rgb_, normals, words32, *a, **k)
    monkeypatch.setattr(type(be), "appearance2_bwd", staticmethod(with_fp32_masks))
    R0, R1, R2 = N.effective_weights([rn.lin0, rn.lin1, rn.lin2])
    rgb16 = N.fused_appearance(pts, dirs, nrm, enc.embeddings, enc.offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                               float(net.divide_factor), mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias, R0, rn.lin0.bias, R1, rn.lin1.bias,
                               R2, rn.lin2.bias)
    got = [g.float() for g in torch.autograd.grad((rgb16 * cot).sum(), [nrm] + params)]
    monkeypatch.setattr(type(be), "appearance2_bwd", staticmethod(raw_bwd))
    _report("full_c1 aligned colour branch: fraction of mask WORDS that differ between the bf16 and the fp32 forward", flips["colour"])
    for a, b, n in zip(got, ref, ["d_normals"] + names):
        if n == "colour table":
            chk("colour table worst level-norm rel", level_norm_err(a, b, offsets), 2e-2)
            _report("full_c1 aligned colour table relL2 (element-wise)", rl2(a, b))
        else:
            chk(f"colour {n} relL2", rl2(a, b), 1e-2)
    # ---------------------------------------------------------------- trunk
    net.set_mlp_precision("fp32")
    genc = net.encoding
    x_all = torch.cat([pts, (torch.rand(4096, 3, device=DEV) * 2 - 1)], 0).contiguous()
    n_main, Be, K = B, 4096, net.d_out
    l0, l1, l2 = net._lins()
    tparams = [genc.embeddings] + [p for l in (l0, l1, l2) for p in (l.weight_v, l.weight_g, l.bias)]
    tnames = ["geometry table"] + [f"lin{i}.{n}" for i in range(3) for n in ("v", "g", "bias")]
    g = torch.Generator(device=DEV).manual_seed(7)
    c_raw = torch.randn(n_main, K, device=DEV, generator=g) * 1e-3
    c_sdf = torch.randn(n_main, 1, device=DEV, generator=g) * 1e-2
    c_grad = torch.randn(n_main, 3, device=DEV, generator=g) * 1e-3
    c_yeik = torch.randn(Be, K, device=DEV, generator=g) * 1e-3
    c_theta = torch.randn(K * Be, 3, device=DEV, generator=g) * 1e-3          # the K per-object gradient rows (not the arg-min rows)
    # fp32: value + Jacobian rows through the product's fp32 path
    y, J = net.sdf_and_jacobian(x_all)
    y, J = y[:, :K], J[:, :K]
    sdf32, idx32 = y[:n_main].min(-1, keepdim=True)
    # bf16 forward first (for its arg-min), without building a graph
    net.set_mlp_precision("bf16")
    W0, W1, W2 = N.effective_weights([l0, l1, l2])
    args = (x_all, n_main, genc.embeddings, genc.offsets, float(np.log2(genc.per_level_scale)), int(genc.base_resolution), net.embedder.multires,
            float(net.divide_factor), W0, l0.bias, W1, l1.bias, W2, l2.bias)
    raw16, sdf16, idx16, grad16, yeik16, mineik16, theta16 = N.trunk_render(*args)
    same = (idx16[:n_main] == idx32).float()
    _report("full_c1 aligned trunk: fraction of samples whose arg-min differs between bf16 and fp32", 1.0 - float(same.mean()))

    def objective(raw, sdf, grad, yeik, theta_rows):
        return (raw * c_raw).sum() + (sdf * c_sdf * same).sum() + (grad * c_grad * same).sum() + (yeik * c_yeik).sum() + (theta_rows * c_theta).sum()
    grad32 = torch.gather(J[:n_main], 1, idx32.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
    theta32 = J[n_main:].transpose(0, 1).reshape(-1, 3)
    tref = [t.float() for t in torch.autograd.grad(objective(y[:n_main], sdf32, grad32, y[n_main:], theta32), tparams)]
    tgot = [t.float() for t in torch.autograd.grad(objective(raw16, sdf16, grad16, yeik16, theta16[:K * Be]), tparams)]
    for a, b, n in zip(tgot, tref, tnames):
        if n == "geometry table":
            chk("geometry table worst level-norm rel", level_norm_err(a, b, offsets), 2e-2)
            _report("full_c1 aligned geometry table relL2 (element-wise)", rl2(a, b))
        else:
            chk(f"trunk {n} relL2", rl2(a, b), 1.5e-2)       # measured 0.01-1.02 % (lin1.v, lin1.bias at 1.02 %; the other seven <= 0.92 %)
    chk.done()


def test_full_size_configs3_frame_584_by_876():
    """BASELINE configs[3]'s frame (confs/scannetpp/*.conf:41: img_res = [584, 876]; SURVEY 8(d) C4: R = 1 024, S = 128, K = 21 on the shared net) at
    size, by properties: (1) the on-device pixel draw + row gather over the 511 584-pixel non-square frame equals fancy indexing on an identically
    seeded scene, every uv inside the frame and the draw reaching both beyond the short side; (2) the fused ray set-up equals rend_util's lift at
    this frame's intrinsics; (3) the whole-iteration bf16 graph of a K = 21 model runs regular and background-patch iterations on it (the patch
    origin is drawn from the frame's own extent) with finite loss terms, its first objective within 2 % of the eager iteration on the same batch."""
    from holoscene_amd.training.synthetic import SyntheticScene
    from holoscene_amd.training.trainer import Stage1Trainer, benchmark_model_state, stock_conf
    from holoscene_amd.utils import rend_util
    H, W, R, K = 584, 876, 1024, 21
    a = SyntheticScene(R, K, img_res=(H, W), num_frames=2, seed=7, device=DEV)
    b = SyntheticScene(R, K, img_res=(H, W), num_frames=2, seed=7, device=DEV)
    _, mi, gt = a.next_batch()
    dst_i = {k: torch.full_like(v, -3) for k, v in mi.items()}
    dst_g = {k: torch.full_like(v, -3) for k, v in gt.items()}
    b.write_batch(dst_i, dst_g)
    seen_u = seen_v = 0.0
    for _ in range(3):
        for k in mi:
            assert torch.equal(mi[k], dst_i[k]), k
        for k in gt:
            assert torch.equal(gt[k], dst_g[k]), k
        uv = dst_i["uv"][0]
        assert float(uv[:, 0].min()) >= 0 and float(uv[:, 0].max()) <= W - 1 and float(uv[:, 1].min()) >= 0 and float(uv[:, 1].max()) <= H - 1
        assert len(torch.unique(gt["segs"])) == K                       # the instance-balanced draw (ns_dataset.py:409-430) sees every stripe
        seen_u, seen_v = max(seen_u, float(uv[:, 0].max())), max(seen_v, float(uv[:, 1].max()))
        _, mi, gt = a.next_batch()
        b.write_batch(dst_i, dst_g)
    assert seen_u > H and seen_v > 0.9 * (H - 1), (seen_u, seen_v)      # columns beyond the short side, rows down to the bottom
    conf = stock_conf(num_rays=R, S=128, d_out=K, beta=0.001, mlp_precision="bf16")
    tr = Stage1Trainer(conf, device=DEV, optimizer="flat", graph=True, freeze_parameters=True)
    benchmark_model_state(tr.model, 0.001)
    model = tr.model.train()
    # (2) rays of this frame: k_ray_setup against rend_util.get_camera_params (no pixel jitter: offsets None)
    rays = model._setup_rays_fused(mi["uv"], None, mi["pose"], mi["intrinsics"], None)
    dirs_ref, loc_ref = rend_util.get_camera_params(mi["uv"].clone(), mi["pose"], mi["intrinsics"])
    assert torch.allclose(rays["ray_dirs"], dirs_ref.reshape(-1, 3), atol=2e-6) and torch.allclose(rays["cam_loc"][0], loc_ref.reshape(-1)[:3], atol=1e-7)
    # (3) iterations 0 (background patch) .. 11 (the next one with a patch is 10) through the whole-iteration graph
    eager = Stage1Trainer(conf, device=DEV, optimizer="flat", graph=False, freeze_parameters=True)
    eager.model.load_state_dict(tr.model.state_dict())
    c = SyntheticScene(R, K, img_res=(H, W), num_frames=2, seed=11, device=DEV)
    d = SyntheticScene(R, K, img_res=(H, W), num_frames=2, seed=11, device=DEV)
    torch.manual_seed(5)
    _, l_e = eager.train_step(*c.next_batch())
    torch.manual_seed(5)
    _, l_g = tr.train_step_resident(d)
    le, lg = float(l_e["loss"]), float(l_g["loss"])
    print(f"PARITY configs[3] frame 584 x 876, K = 21, iteration 0 (background patch): objective eager {le:.6f} / whole-iteration graph {lg:.6f}")
    assert np.isfinite(le) and abs(le - lg) <= 2e-2 * abs(le), (le, lg)
    for i in range(1, 12):
        _, lo = tr.train_step_resident(d)
        vals = {k: float(v) for k, v in lo.items() if torch.is_tensor(v) and v.numel() == 1}
        assert all(np.isfinite(v) for v in vals.values()), (i, vals)
    assert len(tr._graphs) == 2 and int(torch.as_tensor(tr.model.ray_sampler._rounds).reshape(-1)[0]) >= 1
