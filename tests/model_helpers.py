import torch

from helpers import section


def conf_from_meta(rec, **over):
    from holoscene_amd.utils.conf import Conf
    m = {k[5:]: int(v) for k, v in rec.items() if k.startswith("meta.")}
    S = m["S"]
    beta = float(rec["state.density.beta"])
    c = Conf(
        feature_vector_size=m["feat"], scene_bounding_sphere=1.0, use_bg_reg=True, render_bg_iter=10,
        implicit_network=dict(d_in=3, d_out=m["K"], dims=[m["width"]] * 2, geometric_init=True, bias=0.9, skip_in=[4], weight_norm=True,
                              multires=6, inside_outside=True, use_grid_feature=True, divide_factor=1.0, sigmoid=10, color_grid_feature=True,
                              base_size=m["base"], end_size=m["end"], logmap=m["logmap"], num_levels=m["L"], level_dim=2),
        rendering_network=dict(mode="idr", d_in=9, d_out=3, dims=[m["width"]] * 2, weight_norm=True, multires_view=4, multires_point=4,
                               multires_normal=4),
        density=dict(params_init=dict(beta=beta), beta_min=0.0001),
        ray_sampler=dict(near=0.0, N_samples=S // 2, N_samples_eval=S, N_samples_extra=S // 4, eps=0.1, beta_iters=10, max_total_iters=5))
    c.update(over)
    return c


def build_model(rec, device="cpu"):
    from holoscene_amd.model.network import HoloSceneNetwork
    model = HoloSceneNetwork(conf=conf_from_meta(rec), graph_node_dict=None, num_images=4)
    model.load_state_dict(section(rec, "state."))
    return model.to(device)


def build_loss():
    from holoscene_amd.model.loss import HoloSceneLoss
    return HoloSceneLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, smooth_weight=0.005, depth_weight=0.5, normal_l1_weight=0.05,
                         normal_cos_weight=0.05, semantic_loss="torch.nn.MSELoss", use_obj_opacity=True, semantic_weight=5.0,
                         reg_vio_weight=0.01, bg_reg_weight=0.01, depth_type="marigold")


def close(a, b, rtol, atol, what=""):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {err.max():.3e}"


def z_close(z, ref, atol=1e-5, frac_loose=0.02):
    """Sample depths.  Inverse-CDF sampling is ill-conditioned where the pdf is ~0 (a sample may slide
    anywhere inside an almost-empty bin when the SDF changes in its last bits), so: nearly all entries
    must agree to `atol`; the few that do not must still lie within the neighbouring reference samples."""
    z, ref = z.detach().cpu().double(), ref.detach().cpu().double()
    assert z.shape == ref.shape
    err = (z - ref).abs()
    loose = err > atol + 1e-5 * ref.abs()
    assert loose.double().mean() <= frac_loose, f"{int(loose.sum())}/{loose.numel()} depths differ (max {err.max():.3e})"
    lo = torch.cat([ref[:, :1], ref[:, :-1]], 1)
    hi = torch.cat([ref[:, 1:], ref[:, -1:]], 1)
    inside = (z >= lo - 1e-6) & (z <= hi + 1e-6)
    assert bool(inside[loose].all()), "a deviating depth left its reference bracket"
