"""oracle/stage1_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-PyTorch CPU restatement of one HoloScene Stage-1 training iteration
(everything above the hash kernels, which live in oracle/hash_oracle.c).  It is
written functionally over a flat ``state_dict`` whose keys are the reference's
(checkpoint row) and takes every random draw as an explicit
tensor so that oracle, reference import and HIP product can be
driven by identical numbers.

It is pinned by tests/golden/*.npz, which tests/golden/make_golden.py produced by
importing the reference's own Python modules in the build container.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module; the product never does.

Reference map (file:line under /root/reference):
  posenc                 model/embedder.py:11-36
  implicit forward       model/network.py:169-210
  get_outputs / gradient model/network.py:273-301 / 212-254
  rendering network      model/network.py:585-614
  Laplace density        model/density.py:21-30
  volume rendering       model/network.py:1803-1824
  uniform sampler        model/ray_sampler.py:48-83
  error-bound sampler    model/ray_sampler.py:130-287, 450-458
  camera rays            utils/rend_util.py:56-98, 112-125
  network forward        model/network.py:778-971
  loss                   model/loss.py:181-346, 389-403, 487-547, 611-666
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import hash_oracle


class Cfg:
    """Plain attribute bag mirroring confs/replica/room_0/replica_room_0.conf."""

    def __init__(self, **kw):
        d = dict(
            feature_vector_size=256, scene_bounding_sphere=1.0, white_bkgd=False, use_bg_reg=True, render_bg_iter=10,
            d_out=32, dims=(256, 256), bias=0.9, multires=6, divide_factor=1.0, sigmoid=10.0,
            num_levels=16, level_dim=2, base_size=16, end_size=2048, logmap=19,
            multires_view=4, render_dims=(256, 256),
            beta_init=0.1, beta_min=1e-4,
            near=0.0, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10, max_total_iters=5,
            add_tiny=1e-6,
            # loss
            eikonal_weight=0.1, smooth_weight=0.005, depth_weight=0.5, normal_l1_weight=0.05, normal_cos_weight=0.05,
            semantic_weight=5.0, reg_vio_weight=0.01, bg_reg_weight=0.01,
        )
        d.update(kw)
        self.__dict__.update(d)

    @property
    def per_level_scale(self):
        return hash_oracle.per_level_scale_for(self.base_size, self.end_size, self.num_levels)

    @property
    def far(self):
        return 2.0 * self.scene_bounding_sphere * 1.75  # ray_sampler.py:19,110


# --------------------------------------------------------------------------- networks
def posenc(x, n_freq):
    out = [x]
    for k in range(n_freq):
        f = 2.0 ** k
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, -1)


def wn_linear(x, sd, prefix):
    v, g, b = sd[prefix + ".weight_v"], sd[prefix + ".weight_g"], sd[prefix + ".bias"]
    return F.linear(x, torch._weight_norm(v, g, 0), b)  # W = g * v / ||v||_row (nn.utils.weight_norm, dim=0)


class Stage1Oracle:
    def __init__(self, cfg, state_dict):
        self.cfg = cfg
        self.sd = state_dict
        self.training = True

    # ---- hash encoders (hashencoder/hashgrid.py:154-166)
    def _encode(self, key, x):
        c = self.cfg
        x01 = (x / c.divide_factor + 1) / 2
        return hash_oracle.hash_encode(x01, self.sd[f"implicit_network.{key}.embeddings"],
                                       self.sd[f"implicit_network.{key}.offsets"], c.per_level_scale, c.base_size,
                                       x01.requires_grad)

    # ---- ObjectImplicitNetworkGrid.forward (network.py:169-210)
    def implicit(self, x, with_color=True):
        c, sd = self.cfg, self.sd
        feat = self._encode("encoding", x)
        h = torch.cat([posenc(x, c.multires), feat], -1)
        n_lin = len(c.dims) + 1
        for l in range(n_lin):
            h = wn_linear(h, sd, f"implicit_network.lin{l}")
            if l < n_lin - 1:
                h = F.softplus(h, beta=100)
        if not with_color:
            return h
        cf = self._encode("color_encoding", x)
        cf = F.linear(cf, sd["implicit_network.color_grid_feature_map_mlp.0.weight"], sd["implicit_network.color_grid_feature_map_mlp.0.bias"])
        cf = F.linear(F.relu(cf), sd["implicit_network.color_grid_feature_map_mlp.2.weight"], sd["implicit_network.color_grid_feature_map_mlp.2.bias"])
        return torch.cat([h, cf], -1)

    def sdf_raw(self, x):
        return self.implicit(x)[:, : self.cfg.d_out]

    def sdf_vals(self, x):  # network.py:305-311 (min over objects)
        return self.sdf_raw(x).min(-1, keepdim=True)[0]

    def object_sdf_vals(self, x, idx):  # network.py:316-318
        return self.implicit(x)[:, idx]

    def get_outputs(self, x, idx=None):  # network.py:273-301 / 337-357
        c = self.cfg
        x.requires_grad_(True)
        out = self.implicit(x)
        raw = out[:, : c.d_out]
        semantic = c.sigmoid * torch.sigmoid(-c.sigmoid * raw)
        sdf = raw.min(-1, keepdim=True)[0]
        grads = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0]
        return sdf, out[:, c.d_out:], grads, semantic, (raw if idx is None else raw[:, idx])

    def gradient(self, x):  # network.py:212-254
        c = self.cfg
        x.requires_grad_(True)
        y = self.implicit(x)[:, : c.d_out]
        rows = []
        for k in range(c.d_out):
            sel = torch.zeros_like(y)
            sel[:, k] = 1
            rows.append(torch.autograd.grad(y, x, sel, create_graph=True, retain_graph=True)[0])
        sdf = y.min(-1, keepdim=True)[0]
        rows.append(torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0])
        return torch.cat(rows)

    def rendering(self, points, normals, dirs, feats):  # network.py:585-614
        m = self.cfg.multires_view
        h = torch.cat([posenc(points, m), posenc(dirs, m), posenc(normals, m), feats], -1)
        n_lin = len(self.cfg.render_dims) + 1
        for l in range(n_lin):
            h = wn_linear(h, self.sd, f"rendering_network.lin{l}")
            if l < n_lin - 1:
                h = F.relu(h)
        return torch.sigmoid(h[:, :3])

    # ---- density (density.py:21-30)
    def beta(self):
        return self.sd["density.beta"].abs() + self.cfg.beta_min

    def density(self, sdf, beta=None):
        if beta is None:
            beta = self.beta()
        return (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))

    # ---- compositing (network.py:1803-1824)
    def volume_rendering(self, z, sdf):
        sigma = self.density(sdf).reshape(-1, z.shape[1])
        dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((z.shape[0], 1), 1e10)], -1)
        fe = dists * sigma
        shifted = torch.cat([torch.zeros(z.shape[0], 1), fe[:, :-1]], -1)
        alpha = 1 - torch.exp(-fe)
        T = torch.exp(-torch.cumsum(shifted, -1))
        return alpha * T, T, dists

    def occlusion_opacity(self, T, dists, raw):
        sig = self.density(raw).transpose(0, 1).reshape(-1, dists.shape[0], dists.shape[1])
        return (1 - torch.exp(-dists * sig)) * T

    # ---- samplers
    def cube_far(self, o, d):  # ray_sampler.py:48-60
        b = self.cfg.scene_bounding_sphere
        tmin = (-b - o) / (d + 1e-15)
        tmax = (b - o) / (d + 1e-15)
        near = torch.where(tmin < tmax, tmin, tmax).max(-1, keepdim=True)[0]
        far = torch.where(tmin > tmax, tmin, tmax).min(-1, keepdim=True)[0]
        miss = far < near
        far = torch.where(miss, torch.full_like(far, 1e9), far)
        return far.clamp(max=self.cfg.far)

    def uniform_z(self, d, o, t_rand):  # ray_sampler.py:63-83
        c = self.cfg
        far = self.cube_far(o, d)
        near = torch.full_like(far, c.near)
        t = torch.linspace(0.0, 1.0, c.N_samples_eval)
        z = near * (1 - t) + far * t
        if self.training:
            mids = 0.5 * (z[:, 1:] + z[:, :-1])
            upper = torch.cat([mids, z[:, -1:]], -1)
            lower = torch.cat([z[:, :1], mids], -1)
            z = lower + (upper - lower) * t_rand
        return z

    def error_bound(self, beta, sdf, z, dists, d_star):  # ray_sampler.py:450-458
        sigma = self.density(sdf.reshape(z.shape), beta=beta)
        shifted = torch.cat([torch.zeros(z.shape[0], 1), dists * sigma[:, :-1]], -1)
        integral = torch.cumsum(shifted, -1)
        per_sec = torch.exp(-d_star / beta) * dists ** 2 / (4 * beta ** 2)
        e_int = torch.cumsum(per_sec, -1)
        bound = (torch.exp(e_int).clamp(max=1e6) - 1) * torch.exp(-integral[:, :-1])
        return bound.max(-1)[0]

    def sample_z(self, d, o, rand, idx=None):
        """ErrorBoundSampler.get_z_vals (ray_sampler.py:130-287).

        rand: dict with 't_rand' [R,S], 'u_final' [R,N_samples], 'perm' (long, >= N_extra entries,
        a permutation of range(final z count)), 'eik_idx' [R] long.  Returns (z, z_eik, rounds).
        """
        c = self.cfg
        with torch.no_grad():
            beta0 = self.beta().detach()
            z = self.uniform_z(d, o, rand.get("t_rand"))
            samples, order, sdf = z, None, None
            dists = z[:, 1:] - z[:, :-1]
            beta = torch.sqrt((1.0 / (4.0 * torch.log(torch.tensor(c.eps + 1.0)))) * (dists ** 2).sum(-1))  # fp32 log, as :138-140
            it, more = 0, True
            while more and it < c.max_total_iters:
                pts = (o[:, None] + samples[:, :, None] * d[:, None]).reshape(-1, 3)
                if idx is None:
                    new = self.sdf_vals(pts)
                else:
                    new = self.object_sdf_vals(pts, idx)[:, None]
                if order is not None:
                    merged = torch.cat([sdf.reshape(-1, z.shape[1] - samples.shape[1]), new.reshape(-1, samples.shape[1])], -1)
                    sdf = torch.gather(merged, 1, order).reshape(-1, 1)
                else:
                    sdf = new
                dd = sdf.reshape(z.shape)
                dists = z[:, 1:] - z[:, :-1]
                a, b, cc = dists, dd[:, :-1].abs(), dd[:, 1:].abs()
                first = a ** 2 + b ** 2 <= cc ** 2
                second = a ** 2 + cc ** 2 <= b ** 2
                s = (a + b + cc) / 2
                heron = 2.0 * torch.sqrt(s * (s - a) * (s - b) * (s - cc)) / a
                d_star = torch.zeros_like(a)
                d_star = torch.where(first, b, d_star)
                d_star = torch.where(second, cc, d_star)
                d_star = torch.where(~first & ~second & (b + cc - a > 0), heron, d_star)
                d_star = (dd[:, 1:].sign() * dd[:, :-1].sign() == 1) * d_star
                err = self.error_bound(beta0, sdf, z, dists, d_star)
                beta = torch.where(err <= c.eps, beta0.expand_as(beta), beta)
                lo, hi = beta0.expand_as(beta).clone(), beta.clone()
                for _ in range(c.beta_iters):
                    mid = (lo + hi) / 2
                    err = self.error_bound(mid[:, None], sdf, z, dists, d_star)
                    hi = torch.where(err <= c.eps, mid, hi)
                    lo = torch.where(err > c.eps, mid, lo)
                beta = hi
                sigma = self.density(sdf.reshape(z.shape), beta=beta[:, None])
                dists1 = torch.cat([dists, torch.full((z.shape[0], 1), 1e10)], -1)
                fe = dists1 * sigma
                shifted = torch.cat([torch.zeros(z.shape[0], 1), fe[:, :-1]], -1)
                T = torch.exp(-torch.cumsum(shifted, -1))
                w = (1 - torch.exp(-fe)) * T
                it += 1
                more = bool(beta.max() > beta0)
                upsample = more and it < c.max_total_iters
                if upsample:
                    N = c.N_samples_eval
                    per_sec = torch.exp(-d_star / beta[:, None]) * dists ** 2 / (4 * beta[:, None] ** 2)
                    pdf = (torch.exp(torch.cumsum(per_sec, -1)).clamp(max=1e6) - 1) * T[:, :-1] + c.add_tiny
                else:
                    N = c.N_samples
                    pdf = w[:, :-1] + 1e-5
                pdf = pdf / pdf.sum(-1, keepdim=True)
                cdf = torch.cat([torch.zeros(z.shape[0], 1), torch.cumsum(pdf, -1)], -1)
                if upsample or not self.training:
                    u = torch.linspace(0.0, 1.0, N)[None].repeat(z.shape[0], 1)
                else:
                    u = rand["u_final"]
                u = u.contiguous()
                inds = torch.searchsorted(cdf, u, right=True)
                below = (inds - 1).clamp(min=0)
                above = inds.clamp(max=cdf.shape[-1] - 1)
                c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
                b0, b1 = torch.gather(z, 1, below), torch.gather(z, 1, above)
                den = c1 - c0
                den = torch.where(den < 1e-5, torch.ones_like(den), den)
                samples = b0 + (u - c0) / den * (b1 - b0)
                if upsample:
                    z, order = torch.sort(torch.cat([z, samples], -1), dim=-1, stable=True)
            R = z.shape[0]
            near = torch.full((R, 1), c.near)
            far = torch.full((R, 1), c.far)
            if c.N_samples_extra > 0:
                if self.training:
                    perm = rand["perm"]
                    pick = perm[perm < z.shape[1]][: c.N_samples_extra]  # no-op filter when perm = randperm(len(z))
                else:
                    pick = torch.linspace(0, z.shape[1] - 1, c.N_samples_extra).long()
                extra = torch.cat([near, far, z[:, pick]], -1)
            else:
                extra = torch.cat([near, far], -1)
            z_out = torch.sort(torch.cat([samples, extra], -1), -1)[0]
            if "eik_idx" in rand:
                z_eik = torch.gather(z_out, 1, rand["eik_idx"][:, None])
            else:
                z_eik = None
        return z_out, z_eik, it

    # ---- camera (rend_util.py:56-125)
    @staticmethod
    def camera_rays(uv, pose, K):
        fx, fy, cx, cy, sk = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 0, 1]
        x, y = uv[..., 0], uv[..., 1]
        z = torch.ones_like(x)
        xl = (x - cx[:, None] + cy[:, None] * sk[:, None] / fy[:, None] - sk[:, None] * y / fy[:, None]) / fx[:, None] * z
        yl = (y - cy[:, None]) / fy[:, None] * z
        pc = torch.stack([xl, yl, z, torch.ones_like(z)], -1).permute(0, 2, 1)
        world = torch.bmm(pose, pc).permute(0, 2, 1)
        world = world[..., :3] / world[..., 3:4]
        loc = pose[:, :3, 3]
        return F.normalize(world - loc[:, None], dim=2), loc

    # ---- HoloSceneNetwork.forward (network.py:778-971), training mode
    def forward(self, uv, pose, K, rand, iter_step=-1):
        """rand: 'ray_offset' [1,R,2] (already minus 0.5), sampler draws (see sample_z),
        'eik_uniform' [R,3] in [-bs,bs], 'eik_jitter' [2R,3] in [0,1); for bg iterations
        'bg_xy0' (x0,y0) ints and 'bg' dict of sampler draws."""
        c = self.cfg
        off = rand["ray_offset"]
        dirs, loc = self.camera_rays(uv + off, pose, K)
        # reference quirk Q1: uv was modified in place, so the depth-scale rays see 2x the offset
        dtmp, _ = self.camera_rays(uv + 2 * off, torch.eye(4)[None], K)
        depth_scale = dtmp[0, :, 2:]
        R = dirs.shape[1]
        o = loc[:, None].repeat(1, R, 1).reshape(-1, 3)
        d = dirs.reshape(-1, 3)
        z, z_eik, rounds = self.sample_z(d, o, rand)
        N = z.shape[1]
        pts = (o[:, None] + z[:, :, None] * d[:, None]).reshape(-1, 3)
        dflat = d[:, None].repeat(1, N, 1).reshape(-1, 3)
        sdf, feats, grads, semantic, raw = self.get_outputs(pts)
        rgb = self.rendering(pts, grads, dflat, feats).reshape(-1, N, 3)
        semantic = semantic.reshape(-1, N, c.d_out)
        w, T, dists = self.volume_rendering(z, sdf)
        opacity = self.occlusion_opacity(T, dists, raw).sum(-1).transpose(0, 1)
        out = {
            "rgb": rgb,
            "semantic_values": (w[..., None] * semantic).sum(1),
            "object_opacity": opacity,
            "rgb_values": (w[..., None] * rgb).sum(1),
            "depth_values": depth_scale * ((w * z).sum(1, keepdim=True) / (w.sum(1, keepdim=True) + 1e-8)),
            "z_vals": z,
            "depth_vals": z * depth_scale,
            "sdf": sdf.reshape(z.shape),
            "weights": w,
            "sampler_rounds": rounds,
        }
        if self.training:
            near_pts = (o[:, None] + z_eik[:, :, None] * d[:, None]).reshape(-1, 3)
            e = torch.cat([rand["eik_uniform"], near_pts], 0)
            e = torch.cat([e, e + (rand["eik_jitter"] - 0.5) * 0.01], 0)
            g = self.gradient(e)
            out["sample_sdf"] = self.sdf_raw(e)
            out["sample_minsdf"] = self.sdf_vals(e)
            out["grad_theta"] = g[: g.shape[0] // 2]
            out["grad_theta_nei"] = g[g.shape[0] // 2:]
        normals = (grads / (grads.norm(2, -1, keepdim=True) + 1e-6)).reshape(-1, N, 3)
        nmap = (w[..., None] * normals).sum(1)
        rot = pose[0, :3, :3].t()
        out["normal_map"] = (rot @ nmap.t()).t().contiguous()
        if c.use_bg_reg and iter_step % c.render_bg_iter == 0:  # network.py:916-968
            x0, y0 = rand["bg_xy0"]
            ps = 32
            gx, gy = np.meshgrid(np.arange(ps), np.arange(ps), indexing="xy")
            uv0 = torch.from_numpy(np.stack([gx + x0, gy + y0], -1).reshape(1, -1, 2)).float()
            dirs0, loc0 = self.camera_rays(uv0, pose, K)
            dt0, _ = self.camera_rays(uv0, torch.eye(4)[None], K)
            ds0 = dt0[0, :, 2:]
            R0 = dirs0.shape[1]
            o0 = loc0[:, None].repeat(1, R0, 1).reshape(-1, 3)
            d0 = dirs0.reshape(-1, 3)
            bz, _, _ = self.sample_z(d0, o0, rand["bg"], idx=0)
            Nb = bz.shape[1]
            bpts = (o0[:, None] + bz[:, :, None] * d0[:, None]).reshape(-1, 3)
            ssdf, _, bgrads, ssem, bsdf = self.get_outputs(bpts, idx=0)
            bw, _, _ = self.volume_rendering(bz, bsdf)
            sw, _, _ = self.volume_rendering(bz, ssdf)
            sval = (sw[..., None] * ssem.reshape(-1, Nb, c.d_out)).sum(1)
            out["bg_mask"] = sval.argmax(-1, keepdim=True)
            out["bg_depth_values"] = ds0 * ((bw * bz).sum(1, keepdim=True) / (bw.sum(1, keepdim=True) + 1e-8))
            bn = (bgrads / (bgrads.norm(2, -1, keepdim=True) + 1e-6)).reshape(-1, Nb, 3)
            out["bg_normal_map"] = (rot @ (bw[..., None] * bn).sum(1).t()).t().contiguous()
        return out

    # ---- HoloSceneLoss.forward (loss.py:611-666 over :290-346)
    def loss(self, out, gt, call_reg=False):
        c = self.cfg
        rgb_loss = (out["rgb_values"] - gt["rgb"].reshape(-1, 3)).abs().mean()
        eik = ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).mean() if "grad_theta" in out else torch.tensor(0.0)
        fg = ((out["sdf"] > 0).any(-1) & (out["sdf"] < 0).any(-1))[None, :, None] & (gt["mask"] > 0.5)
        # depth: scale/shift-invariant LSQ (loss.py:181-193, 246-277)
        p = out["depth_values"].reshape(1, -1)
        t = gt["depth"].reshape(1, -1)
        A = torch.stack([p, torch.ones_like(p)], -1)  # [1,N,2]
        M = (A[..., :, None] * A[..., None, :]).sum(1)
        rhs = (A * t[..., None]).sum(1)[..., None]
        ws = (torch.inverse(M) @ rhs).reshape(1, 2)
        depth_loss = (((ws[:, :1] * p + ws[:, 1:]) - t) ** 2).clamp(max=1).mean()
        npred = F.normalize(out["normal_map"][None] * fg, p=2, dim=-1)
        ngt = F.normalize(gt["normal"], p=2, dim=-1)
        n_l1 = (npred - ngt).abs().sum(-1).mean()
        n_cos = (1.0 - (npred * ngt).sum(-1)).mean()
        g1, g2 = out["grad_theta"], out["grad_theta_nei"]
        n1 = g1 / (g1.norm(2, dim=1)[:, None] + 1e-5)
        n2 = g2 / (g2.norm(2, dim=1)[:, None] + 1e-5)
        smooth = (n1 - n2).norm(dim=-1).mean()
        total = rgb_loss + c.eikonal_weight * eik + c.smooth_weight * smooth + c.depth_weight * depth_loss \
            + c.normal_l1_weight * n_l1 + c.normal_cos_weight * n_cos
        # opacity BCE (loss.py:487-492)
        target = F.one_hot(gt["segs"].reshape(-1).long(), num_classes=out["object_opacity"].shape[1]).float()
        op = out["object_opacity"].clamp(1e-4, 1 - 1e-4)
        sem = F.binary_cross_entropy(op, target, reduction="none").mean(-1).mean()
        # collision (loss.py:389-403)
        if call_reg and "sample_sdf" in out:
            sv, mn = out["sample_sdf"], out["sample_minsdf"]
            arg = sv.argmin(1)
            v = torch.relu(-sv - mn.detach())
            keep = torch.ones_like(v, dtype=torch.bool)
            keep[torch.arange(v.shape[0]), arg] = False
            v = v[keep]
            cnt = (v > 0).sum()
            reg = v.sum() / cnt if cnt > 0 else torch.tensor(0.0)
        else:
            reg = torch.tensor(0.0)
        # bg smoothness (loss.py:495-547)
        if "bg_depth_values" in out:
            m = (out["bg_mask"] != 0).int().reshape(1, 32, 32)
            bd = out["bg_depth_values"].reshape(1, 32, 32)
            bn = out["bg_normal_map"].reshape(32, 32, 3).permute(2, 0, 1)
            bg = self._grad_err(bd, m) + self._grad_err(bn, m.repeat(3, 1, 1))
        else:
            bg = torch.tensor(0.0)
        total = total + c.semantic_weight * sem + c.reg_vio_weight * reg + c.bg_reg_weight * bg
        return {"loss": total, "rgb_loss": rgb_loss, "eikonal_loss": eik, "smooth_loss": smooth, "depth_loss": depth_loss,
                "normal_l1": n_l1, "normal_cos": n_cos, "semantic_loss": sem, "collision_reg_loss": reg,
                "background_reg_loss": bg}

    @staticmethod
    def _grad_err(x, mask):
        tot = torch.tensor(0.0)
        for i in range(4):
            st = 2 ** i
            m, xs = mask[:, ::st, ::st], x[:, ::st, ::st]
            div = m[:1].sum()
            v = m * xs
            gx = (m[:, :, 1:] * m[:, :, :-1]) * (v[:, :, 1:] - v[:, :, :-1]).abs()
            gy = (m[:, 1:, :] * m[:, :-1, :]) * (v[:, 1:, :] - v[:, :-1, :]).abs()
            if div != 0:
                tot = tot + (gx.sum() + gy.sum()) / div
        return tot


# --------------------------------------------------------------------------- initial state
def make_state(cfg, seed=42, perturb=0.0):
    """Reference initialisation (network.py:127-161, hashgrid.py:147-149), optional
    N(0,perturb) refill of the columns geometric init zeroes."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    offs = torch.from_numpy(hash_oracle.level_offsets(cfg.num_levels, cfg.base_size, cfg.per_level_scale, cfg.logmap))
    for key in ("encoding", "color_encoding"):
        sd[f"implicit_network.{key}.embeddings"] = (torch.rand(int(offs[-1]), cfg.level_dim, generator=g) * 2 - 1) * 1e-4
        sd[f"implicit_network.{key}.offsets"] = offs.clone()
    d_in0 = 3 + 6 * cfg.multires + cfg.num_levels * cfg.level_dim
    dims = [d_in0] + list(cfg.dims) + [cfg.d_out]
    for l in range(len(dims) - 1):
        w = torch.zeros(dims[l + 1], dims[l])
        b = torch.zeros(dims[l + 1])
        if l == len(dims) - 2:
            w[:1].normal_(-math.sqrt(math.pi) / math.sqrt(dims[l]), 1e-4, generator=g)
            w[1:].normal_(math.sqrt(math.pi) / math.sqrt(dims[l]), 1e-4, generator=g)
            b[:1] = cfg.bias
            b[1:] = -0.5 * cfg.bias
        elif l == 0:
            w[:, :3].normal_(0.0, math.sqrt(2) / math.sqrt(dims[l + 1]), generator=g)
            if perturb > 0:
                w[:, 3:].normal_(0.0, perturb, generator=g)
        else:
            w.normal_(0.0, math.sqrt(2) / math.sqrt(dims[l + 1]), generator=g)
        sd[f"implicit_network.lin{l}.weight_v"] = w
        sd[f"implicit_network.lin{l}.weight_g"] = w.norm(2, dim=1, keepdim=True)
        sd[f"implicit_network.lin{l}.bias"] = b

    def plain(prefix, n_out, n_in):
        k = 1.0 / math.sqrt(n_in)
        sd[prefix + ".weight"] = (torch.rand(n_out, n_in, generator=g) * 2 - 1) * k
        sd[prefix + ".bias"] = (torch.rand(n_out, generator=g) * 2 - 1) * k

    plain("implicit_network.color_grid_feature_map_mlp.0", 256, cfg.num_levels * cfg.level_dim)
    plain("implicit_network.color_grid_feature_map_mlp.2", cfg.feature_vector_size, 256)
    pe = 3 + 6 * cfg.multires_view
    rdims = [3 * pe + cfg.feature_vector_size] + list(cfg.render_dims) + [3]
    for l in range(len(rdims) - 1):
        k = 1.0 / math.sqrt(rdims[l])
        w = (torch.rand(rdims[l + 1], rdims[l], generator=g) * 2 - 1) * k
        sd[f"rendering_network.lin{l}.weight_v"] = w
        sd[f"rendering_network.lin{l}.weight_g"] = w.norm(2, dim=1, keepdim=True)
        sd[f"rendering_network.lin{l}.bias"] = (torch.rand(rdims[l + 1], generator=g) * 2 - 1) * k
    sd["density.beta"] = torch.tensor(cfg.beta_init)
    return sd
