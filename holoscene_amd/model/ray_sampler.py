"""Ray samplers of the Stage-1 path.

Same classes, constructor arguments and ``get_z_vals`` contract as the reference
(model/ray_sampler.py:16-102 ``UniformSampler``, :105-287 + :450-472 ``ErrorBoundSampler`` =
VolSDF Algorithm 1), restructured for the GPU:

  * no boolean-mask indexing (``d_star[mask] = ...`` in the reference forces a host sync per
    assignment, ray_sampler.py:171-177); everything is ``torch.where`` on whole tensors;
  * the only device->host sync per round is the convergence test the algorithm itself needs
    (``beta.max() > beta0``, ray_sampler.py:204);
  * SDF sweeps go through ``implicit_network.get_sdf_vals`` which, in this build, skips the
    colour branch the reference evaluates and throws away (network.py:177-179);
  * every random draw can be injected (``rng=`` dict) so CPU oracle, reference import and this
    code can be driven by identical numbers.  Without injection the draws
    come from the device generator (the reference draws on the CPU generator and copies,
    ray_sampler.py:79,238,269,279) unless ``cpu_rng=True``.
"""
import abc
import os

import torch

from ..hashencoder import backend as _be
from .density import laplace_density
from ..utils import rend_util as _rend_util

# "hip": fused per-ray kernels (csrc/sampler.hip) -- the product path; raises without CUDA tensors / the library.
# "torch": the pre-fusion whole-tensor formulation below, kept for A/B timing and for exercising the host
#          logic on CPU in `-m "not gpu"` tests.  Never selected implicitly.
SAMPLER_IMPL = os.environ.get("HOLOSCENE_SAMPLER_IMPL", "hip")
# "device": Algorithm 1's loop test runs on the GPU (hsSamplerCtl + gated kernels, loop unrolled max_total_iters times): no
#           host sync at all, so the sampler -- and with it the whole training iteration -- can live inside one HIP graph.
#           Needs the fused ray-mode SDF query (bf16 MLP mode, stock trunk shape); other configurations use "host".
# "host":   the host reads one convergence flag per round (the reference's control flow, ray_sampler.py:204).
CONTROL = os.environ.get("HOLOSCENE_SAMPLER_CONTROL", "device")
# device-controlled loop: the next round's draw fused into the update launch, rounds gated on the previous round's max beta ("1"),
# or one draw + control-step launch per round between control slots ("0")
FUSE_DRAW = True
# fp32 configuration: with the fused fp32 SDF sweep (csrc/sdf_mlp32.hip, gated launches) the loop control can live on the device there too
FP32_DEVICE_CONTROL = True
# ... and the tail -- final draw, extra-sample pick, merge / sort -- one launch instead of three (hs_sampler_tail); "0" restores the three
FUSE_TAIL = True


def _rand(shape, device, cpu_rng):
    if cpu_rng:
        return torch.rand(shape).to(device)
    return torch.rand(shape, device=device)


class RaySampler(metaclass=abc.ABCMeta):
    def __init__(self, near, far):
        self.near = near
        self.far = far

    @abc.abstractmethod
    def get_z_vals(self, ray_dirs, cam_loc, model):
        pass


class UniformSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, take_sphere_intersection=False, far=-1, cpu_rng=False):
        super().__init__(near, 2.0 * scene_bounding_sphere * 1.75 if far == -1 else far)  # ray_sampler.py:19
        self.N_samples = N_samples
        self.scene_bounding_sphere = scene_bounding_sphere
        self.take_sphere_intersection = take_sphere_intersection
        self.cpu_rng = cpu_rng

    def near_far_from_cube(self, rays_o, rays_d, bound):
        """Slab test against [-bound, bound]^3 (ray_sampler.py:48-60)."""
        tmin = (-bound - rays_o) / (rays_d + 1e-15)
        tmax = (bound - rays_o) / (rays_d + 1e-15)
        near = torch.minimum(tmin, tmax).max(dim=-1, keepdim=True)[0]
        far = torch.maximum(tmin, tmax).min(dim=-1, keepdim=True)[0]
        miss = far < near
        near = torch.where(miss, torch.full_like(near, 1e9), near)
        far = torch.where(miss, torch.full_like(far, 1e9), far)
        return near.clamp(min=self.near), far.clamp(max=self.far)

    def _stratify(self, near, far, training, t_rand=None):
        t = torch.linspace(0.0, 1.0, steps=self.N_samples, device=near.device)
        z = near * (1.0 - t) + far * t
        if training:
            mids = 0.5 * (z[..., 1:] + z[..., :-1])
            upper = torch.cat([mids, z[..., -1:]], -1)
            lower = torch.cat([z[..., :1], mids], -1)
            if t_rand is None:
                t_rand = _rand(z.shape, z.device, self.cpu_rng)
            z = lower + (upper - lower) * t_rand
        return z

    def get_z_vals(self, ray_dirs, cam_loc, model, t_rand=None):
        R = ray_dirs.shape[0]
        near = torch.full((R, 1), float(self.near), device=ray_dirs.device)
        if self.take_sphere_intersection:
            _, far = self.near_far_from_cube(cam_loc, ray_dirs, bound=self.scene_bounding_sphere)
        else:
            far = torch.full((R, 1), float(self.far), device=ray_dirs.device)
        return self._stratify(near, far, model.training, t_rand), near, far

    def get_z_vals_near_far(self, ray_dirs, cam_loc, model, near, far, t_rand=None):
        R = ray_dirs.shape[0]
        near = near * torch.ones(R, 1, device=ray_dirs.device)
        far = far * torch.ones(R, 1, device=ray_dirs.device)
        return self._stratify(near, far, model.training, t_rand), near, far


def opacity_error_bound(beta, sdf, dists, d_star):
    """Upper bound of the opacity approximation error per ray (ray_sampler.py:450-458).
    sdf [R,M]; dists, d_star [R,M-1]; beta scalar or [R,1]."""
    sigma = laplace_density(sdf, beta)
    free = torch.cumsum(dists * sigma[:, :-1], dim=-1)
    integral = torch.cat([torch.zeros_like(free[:, :1]), free[:, :-1]], -1)  # energy up to the start of each section
    err_int = torch.cumsum(torch.exp(-d_star / beta) * dists ** 2 / (4 * beta ** 2), dim=-1)
    return ((torch.exp(err_int).clamp(max=1.0e6) - 1.0) * torch.exp(-integral)).max(-1)[0]


def heron_distance_bound(sdf, dists):
    """d* of VolSDF Theorem 1 for every section (ray_sampler.py:165-178)."""
    a, b, c = dists, sdf[:, :-1].abs(), sdf[:, 1:].abs()
    first = a ** 2 + b ** 2 <= c ** 2
    second = a ** 2 + c ** 2 <= b ** 2
    s = (a + b + c) / 2.0
    height = 2.0 * torch.sqrt(s * (s - a) * (s - b) * (s - c)) / a
    d_star = torch.where(first, b, torch.zeros_like(a))
    d_star = torch.where(second, c, d_star)
    d_star = torch.where(~first & ~second & (b + c - a > 0), height, d_star)
    same_side = sdf[:, 1:].sign() * sdf[:, :-1].sign() == 1
    return torch.where(same_side, d_star, torch.zeros_like(d_star))


def invert_cdf(cdf, bins, u):
    """Inverse-transform sampling with linear interpolation inside a bin (ray_sampler.py:241-253)."""
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)


class ErrorBoundSampler(RaySampler):
    def __init__(self, scene_bounding_sphere, near, N_samples, N_samples_eval, N_samples_extra, eps, beta_iters, max_total_iters,
                 inverse_sphere_bg=False, N_samples_inverse_sphere=0, add_tiny=1.0e-6, cpu_rng=False):
        super().__init__(near, 2.0 * scene_bounding_sphere * 1.75)  # ray_sampler.py:110
        self.N_samples = N_samples
        self.N_samples_eval = N_samples_eval
        self.take_sphere_intersection = True
        self.uniform_sampler = UniformSampler(scene_bounding_sphere, near, N_samples_eval, take_sphere_intersection=True, cpu_rng=cpu_rng)
        self.N_samples_extra = N_samples_extra
        self.eps = eps
        self.beta_iters = beta_iters
        self.max_total_iters = max_total_iters
        self.scene_bounding_sphere = scene_bounding_sphere
        self.add_tiny = add_tiny
        self.cpu_rng = cpu_rng
        self.inverse_sphere_bg = inverse_sphere_bg
        if inverse_sphere_bg:     # ray_sampler.py:127-128 (no conf of the reference sets it; kept for the constructor's full argument list)
            self.inverse_sphere_sampler = UniformSampler(1.0, 0.0, N_samples_inverse_sphere, False, far=1.0, cpu_rng=cpu_rng)
        self._rounds = 0       # realised Algorithm-1 rounds of the latest call: an int, or a 1-element device tensor
        self._ctl_init = None

    @property
    def last_rounds(self):
        """Realised Algorithm-1 rounds of the latest call (reading it after a device-controlled call synchronises)."""
        return int(self._rounds)

    @last_rounds.setter
    def last_rounds(self, v):
        self._rounds = v

    def device_control_ok(self, model, idx=None):
        net = model.implicit_network
        ok_idx = idx is None or isinstance(idx, int) or (isinstance(idx, (list, tuple)) and len(idx) > 0 and all(isinstance(k, int) for k in idx))
        fused = hasattr(net, "_fused_trunk_supported") and (net._fused_trunk_supported(net.encoding.embeddings)
                                                              or (FP32_DEVICE_CONTROL and net._fused_sdf32_supported(net.encoding.embeddings)))
        return CONTROL == "device" and SAMPLER_IMPL == "hip" and ok_idx and getattr(net, "color_grid_feature", False) and fused

    def _query_sdf(self, model, points, idx):
        net = model.implicit_network
        if idx is None:
            return net.get_sdf_vals(points)
        if isinstance(idx, int):
            return net.get_object_sdf_vals(points, idx).unsqueeze(-1)
        return net.get_multi_object_sdf_vals(points, idx)

    @torch.no_grad()
    def get_z_vals(self, ray_dirs, cam_loc, model, idx=None, rng=None, z0=None, beta_init=None, x0=None, beta_work=None):
        """z0 / beta_init (/ x0 = positions (x, x01) of z0): optionally the first uniform depths and Lemma-2 beta already produced
        by the fused ray-setup kernel (HoloSceneNetwork._setup_rays_fused); otherwise they are computed here as in the reference.
        beta_work: a copy of beta_init this call may overwrite (its per-ray beta state), saving the clone."""
        rng = rng or {}
        bounds = None
        if self.inverse_sphere_bg:
            # ray_sampler.py:262-265: the far sample appended at the end is each ray's exit from the bounding sphere (the near one stays
            # self.near) -- the per-ray bounds the near/far entry point already hands to the final kernels
            far = _rend_util.get_sphere_intersections(cam_loc, ray_dirs, r=self.scene_bounding_sphere)[:, 1]
            bounds = (torch.full_like(far, float(self.near)), far.contiguous().float())
        if SAMPLER_IMPL == "hip":
            if ray_dirs.is_cuda and self.device_control_ok(model, idx):
                out = self._get_z_vals_device(ray_dirs, cam_loc, model, idx, rng, z0, beta_init, x0, bounds=bounds, beta_work=beta_work)
            else:
                out = self._get_z_vals_hip(ray_dirs, cam_loc, model, idx, rng, z0, beta_init, bounds=bounds)
        elif SAMPLER_IMPL == "torch":
            out = self._get_z_vals_torch(ray_dirs, cam_loc, model, idx, rng, bounds=bounds)
        else:
            raise RuntimeError(f"unknown HOLOSCENE_SAMPLER_IMPL={SAMPLER_IMPL!r}")
        if self.inverse_sphere_bg:
            # ray_sampler.py:282-285: uniform depths in (0, 1) of the inverted-sphere parametrisation, returned beside the foreground's
            z_inv, _, _ = self.inverse_sphere_sampler.get_z_vals(ray_dirs, cam_loc, model, t_rand=rng.get("t_rand_inverse"))
            return (out[0], z_inv * (1.0 / self.scene_bounding_sphere)), out[1]
        return out

    @torch.no_grad()
    def get_z_vals_near_far(self, ray_dirs, cam_loc, model, near, far, idx=None, rng=None):
        """Algorithm 1 between caller-supplied bounds (scalars or per-ray [R,1] tensors): the Stage-2/3 entry point
        (ray_sampler.py:290-447; callers network.py:1321, 1472, 1751).  Differs from get_z_vals only in the first uniform samples
        (ray_sampler.py:85-102) and in the near / far samples appended at the end (:434-436 commented out there)."""
        rng = rng or {}
        dev = ray_dirs.device
        z0, near_t, far_t = self.uniform_sampler.get_z_vals_near_far(ray_dirs, cam_loc, model, near, far, t_rand=rng.get("t_rand"))
        d0 = z0[:, 1:] - z0[:, :-1]
        beta_init = torch.sqrt((1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0, device=dev)))) * (d0 ** 2.0).sum(-1)).contiguous()
        bounds = (near_t.reshape(-1).contiguous().float(), far_t.reshape(-1).contiguous().float())
        if SAMPLER_IMPL == "hip":
            if ray_dirs.is_cuda and self.device_control_ok(model, idx):
                return self._get_z_vals_device(ray_dirs, cam_loc, model, idx, rng, z0.contiguous(), beta_init, bounds=bounds)
            return self._get_z_vals_hip(ray_dirs, cam_loc, model, idx, rng, z0.contiguous(), beta_init, bounds=bounds)
        if SAMPLER_IMPL != "torch":
            raise RuntimeError(f"unknown HOLOSCENE_SAMPLER_IMPL={SAMPLER_IMPL!r}")
        return self._get_z_vals_torch(ray_dirs, cam_loc, model, idx, rng, z0=z0, bounds=bounds)

    def _get_z_vals_hip(self, ray_dirs, cam_loc, model, idx, rng, z0=None, beta_init=None, bounds=None):
        """Algorithm 1 with the per-ray arithmetic in three fused kernels per round (update / draw / final).
        Per round: 1 SDF sweep, 2 kernel launches, one 4-byte device->host read for the convergence test."""
        if not ray_dirs.is_cuda:
            raise RuntimeError("ErrorBoundSampler: the fused sampler kernels need CUDA tensors "
                               "(set HOLOSCENE_SAMPLER_IMPL=torch explicitly for the whole-tensor formulation)")
        be = _be._backend
        dev = ray_dirs.device
        R = ray_dirs.shape[0]
        S = self.N_samples_eval
        ld = S * self.max_total_iters
        beta0 = model.density.get_beta().detach().reshape(1).contiguous()
        if hasattr(model.implicit_network, "invalidate_packed_weights"):
            model.implicit_network.invalidate_packed_weights()
        if z0 is None or beta_init is None:
            z0, _, _ = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model, t_rand=rng.get("t_rand"))
            d0 = z0[:, 1:] - z0[:, :-1]
            beta = torch.sqrt((1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0, device=dev)))) * (d0 ** 2.0).sum(-1)).contiguous()
        else:
            beta = beta_init.clone()
        z = torch.empty(R, ld, device=dev)
        sdf = torch.empty(R, ld, device=dev)
        beta_max_all = torch.zeros(self.max_total_iters + 1, device=dev)   # one slot per round: no re-zeroing inside the loop
        samples = z0.contiguous()
        m, rounds = 0, 0
        while True:
            points = (cam_loc.unsqueeze(1) + samples.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
            new_sdf = self._query_sdf(model, points, idx).reshape(R, -1).contiguous()
            beta_max = beta_max_all[rounds:rounds + 1]
            be.sampler_update(z, sdf, m, samples, new_sdf, beta, beta0, float(self.eps), int(self.beta_iters), beta_max)
            m += samples.shape[1]
            rounds += 1
            unconverged = bool(beta_max > beta0)  # the one host sync Algorithm 1 needs per round
            if not (unconverged and rounds < self.max_total_iters):
                break
            samples = torch.empty(R, S, device=dev)
            be.sampler_draw(z, sdf, m, beta, 0, float(self.add_tiny), None, S, samples)
        n = self.N_samples
        if model.training:
            u = (rng["u_final"].to(dev) if "u_final" in rng else _rand((R, n), dev, self.cpu_rng)).contiguous()
        else:
            u = None
        samples = torch.empty(R, n, device=dev)
        be.sampler_draw(z, sdf, m, beta, 1, float(self.add_tiny), u, n, samples)
        self.last_rounds = rounds
        if self.N_samples_extra > 0:
            if model.training:
                if "perm" in rng:
                    perm = rng["perm"]
                elif self.cpu_rng:
                    perm = torch.randperm(m)
                else:   # device draw: no pageable host->device copy in the launch-bound tail of the sampler
                    perm = torch.randperm(m, device=dev)
                pick = perm[: self.N_samples_extra].to(dev).long().contiguous()
            else:
                pick = torch.linspace(0, m - 1, self.N_samples_extra, device=dev).long()
        else:
            pick = None
        n_out = samples.shape[1] + 2 + self.N_samples_extra
        if "eik_idx" in rng:
            eik = rng["eik_idx"].to(dev).long().contiguous()
        elif self.cpu_rng:
            eik = torch.randint(n_out, (R,)).to(dev)
        else:
            eik = torch.randint(n_out, (R,), device=dev)
        z_out = torch.empty(R, n_out, device=dev)
        z_eik = torch.empty(R, 1, device=dev)
        be.sampler_final(samples, z, pick, float(self.near), float(self.far), eik, z_out, z_eik,
                         near_rays=None if bounds is None else bounds[0], far_rays=None if bounds is None else bounds[1])
        if hasattr(model.implicit_network, "invalidate_packed_weights"):
            model.implicit_network.invalidate_packed_weights()   # the images belong to this parameter state only
        return z_out, z_eik


    def _get_z_vals_device(self, ray_dirs, cam_loc, model, idx, rng, z0=None, beta_init=None, x0=None, bounds=None, beta_work=None):
        """Algorithm 1 with device-side loop control: max_total_iters unrolled rounds of gated kernels, zero host syncs."""
        be = _be._backend
        dev = ray_dirs.device
        R = ray_dirs.shape[0]
        S = self.N_samples_eval
        ld = S * self.max_total_iters
        net = model.implicit_network
        beta0 = model.density.get_beta().detach().reshape(1).contiguous()
        net.invalidate_packed_weights()
        if z0 is None or beta_init is None:
            z0, _, _ = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model, t_rand=rng.get("t_rand"))
            d0 = z0[:, 1:] - z0[:, :-1]
            beta = torch.sqrt((1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0, device=dev)))) * (d0 ** 2.0).sum(-1)).contiguous()
        else:
            beta = beta_work if beta_work is not None else beta_init.clone()
        nr = self.max_total_iters
        if self._ctl_init is None or self._ctl_init.device != dev or self._ctl_init.numel() != (nr + 1) * 4 + nr:
            init = torch.zeros((nr + 1) * 4 + nr)            # [nr+1] hsSamplerCtl slots (slot r+1 = state after round r) | [nr] per-round max beta
            init[:4] = torch.tensor([1.0, 0.5, 0.0, 0.0])     # slot 0: {running, half, m = 0, rounds = 0}
            self._ctl_init = init.to(dev)
        if FUSE_DRAW and _be.zero_pool_armed(dev, nr):
            # no control slots between the rounds in this form (below): slot 0 is only ever read -- the constant itself serves --, the end
            # state is written whole, and the per-round maxima come zeroed from the iteration's accumulator pool: no copy launch
            ctl = None
            ctl_first, ctl_last = self._ctl_init[:4], torch.empty(4, device=dev)
            beta_max_all = _be.zeros_small(nr, dev)
        else:
            state = self._ctl_init.clone()                    # one copy initialises the control slots and zeroes the max-beta cells
            ctl, beta_max_all = state[:(nr + 1) * 4].view(nr + 1, 4), state[(nr + 1) * 4:]
            ctl_first, ctl_last = ctl[0], ctl[nr]
        ci = None if ctl is None else ctl.view(torch.int32)
        z = torch.empty(R, ld, device=dev)
        sdf = torch.empty(R, ld, device=dev)
        cam = (cam_loc.expand(R, 3) if cam_loc.shape[0] != R else cam_loc).contiguous()
        dirs = ray_dirs.contiguous()
        sel = -1 if idx is None else (list(idx) if isinstance(idx, (list, tuple)) else idx)
        samples = z0.contiguous()
        df = float(net.divide_factor)
        if x0 is None:
            x, x01 = torch.empty(R * S, 3, device=dev), torch.empty(R * S, 3, device=dev)
            be.ray_points(cam, dirs, samples, x, x01, df)
        else:
            x, x01 = x0
        # per round: hash gather, fused trunk, update, then ONE launch that steps the loop control, draws the next depths and
        # writes their positions (csrc/sampler.hip: hs_sampler_draw_step); every kernel of round r is gated on slot r
        n = self.N_samples
        u = None
        if model.training:
            u = (rng["u_final"].to(dev) if "u_final" in rng else _rand((R, n), dev, self.cpu_rng)).contiguous()
        final = torch.empty(R, n, device=dev)
        if FUSE_DRAW:
            # three launches per round: gather, trunk, update + the next round's draw (hs_sampler_update_draw).  No control slots
            # between the rounds: round r is gated on round r - 1's max beta (a round that did not run leaves its maximum at zero, so
            # the gates chain by themselves) and merges into r * S entries; the final draw derives the realised state (merged count,
            # rounds) from the per-round maxima in one go
            for r in range(nr):
                gate = None if r == 0 else (beta_max_all[r - 1:r], beta0)
                new_sdf = net.sdf_at_points(x, x01, R, S, sel, gate=gate)
                if r + 1 < nr:
                    nxt = torch.empty(R, S, device=dev)
                    xn, xn01 = torch.empty(R * S, 3, device=dev), torch.empty(R * S, 3, device=dev)
                    be.sampler_update_draw(z, sdf, r * S, samples, new_sdf, beta, beta0, float(self.eps), int(self.beta_iters), beta_max_all[r:r + 1],
                                           gate, float(self.add_tiny), nxt, cam, dirs, df, xn, xn01)
                    samples, x, x01 = nxt, xn, xn01
                else:
                    be.sampler_update(z, sdf, r * S, samples, new_sdf, beta, beta0, float(self.eps), int(self.beta_iters), beta_max_all[r:r + 1],
                                      gate=gate)
            fuse_tail = FUSE_TAIL and n + 2 + self.N_samples_extra <= S
            if not fuse_tail:
                be.sampler_draw_steps(z, sdf, beta, 1, float(self.add_tiny), u, n, final, ctl_first, ctl_last, beta_max_all, beta0, S, nr, nr)
        else:
            fuse_tail = False
            for r in range(nr):
                gate, m_dev = (ctl[r, 0:1], ctl[r, 1:2]), ci[r, 2:3]
                new_sdf = net.sdf_at_points(x, x01, R, S, sel, gate=gate)
                be.sampler_update(z, sdf, 0, samples, new_sdf, beta, beta0, float(self.eps), int(self.beta_iters), beta_max_all[r:r + 1], gate=gate,
                                  m_dev=m_dev)
                if r + 1 < nr:
                    samples = torch.empty(R, S, device=dev)
                    x, x01 = torch.empty(R * S, 3, device=dev), torch.empty(R * S, 3, device=dev)
                    be.sampler_draw_step(z, sdf, beta, 0, float(self.add_tiny), None, S, samples, ctl[r], ctl[r + 1], beta_max_all[r:r + 1], beta0, S,
                                         nr, cam, dirs, df, x, x01)
            be.sampler_draw_step(z, sdf, beta, 1, float(self.add_tiny), u, n, final, ctl[nr - 1], ctl[nr], beta_max_all[nr - 1:nr], beta0, S, nr)
        ctl_end = ctl_last
        pick = up = None
        if self.N_samples_extra > 0:
            if "perm" in rng:   # explicit permutation of the (then host-known) merged set: parity tests
                pick = rng["perm"][: self.N_samples_extra].to(dev).long().contiguous()
            else:
                if model.training:
                    up = rng["u_pick"][: self.N_samples_extra].contiguous() if "u_pick" in rng else _rand((self.N_samples_extra,), dev, self.cpu_rng)
                if not fuse_tail:
                    pick = torch.empty(self.N_samples_extra, device=dev, dtype=torch.int64)
                    be.sampler_pick(ctl_end, up, self.N_samples_extra, pick)
        n_out = n + 2 + self.N_samples_extra
        eik_u = None
        if "eik_idx" in rng:
            eik = rng["eik_idx"].to(dev).long().contiguous()
        elif "eik_u" in rng:      # raw U[0,1) draws: the kernel quantises them to [0, n_out)
            eik, eik_u = None, rng["eik_u"].contiguous()
        elif self.cpu_rng:
            eik = torch.randint(n_out, (R,)).to(dev)
        else:
            eik = torch.randint(n_out, (R,), device=dev)
        z_out = torch.empty(R, n_out, device=dev)
        z_eik = torch.empty(R, 1, device=dev)
        nb, fb = (None, None) if bounds is None else (bounds[0], bounds[1])
        if fuse_tail:     # final draw + realised loop state + extra-sample pick + merge / sort: one launch (csrc/sampler.hip: hs_sampler_tail)
            be.sampler_tail(z, sdf, beta, float(self.add_tiny), u, n, final, ctl_first, ctl_last, beta_max_all, beta0, S, nr, nr, up, pick,
                            self.N_samples_extra, float(self.near), float(self.far), eik, z_out, z_eik, near_rays=nb, far_rays=fb, eik_u=eik_u)
        else:
            be.sampler_final(final, z, pick, float(self.near), float(self.far), eik, z_out, z_eik, eik_u=eik_u, near_rays=nb, far_rays=fb)
        net.invalidate_packed_weights()
        self._rounds = ctl_last.view(torch.int32)[3:4]
        return z_out, z_eik

    def _get_z_vals_torch(self, ray_dirs, cam_loc, model, idx, rng, z0=None, bounds=None):
        dev = ray_dirs.device
        R = ray_dirs.shape[0]
        beta0 = model.density.get_beta().detach()
        if z0 is None:
            z_vals, _, _ = self.uniform_sampler.get_z_vals(ray_dirs, cam_loc, model, t_rand=rng.get("t_rand"))
        else:
            z_vals = z0
        samples, order, sdf = z_vals, None, None
        dists = z_vals[:, 1:] - z_vals[:, :-1]
        # Lemma 2: beta that certainly satisfies the bound (fp32 log as the reference, :138-140)
        beta = torch.sqrt((1.0 / (4.0 * torch.log(torch.tensor(self.eps + 1.0, device=dev)))) * (dists ** 2.0).sum(-1))
        rounds, unconverged = 0, True
        while unconverged and rounds < self.max_total_iters:
            points = (cam_loc.unsqueeze(1) + samples.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
            new_sdf = self._query_sdf(model, points, idx).reshape(R, -1)
            if order is None:
                sdf = new_sdf
            else:  # bring old and new values into the merged sample order
                sdf = torch.gather(torch.cat([sdf, new_sdf], -1), 1, order)
            dists = z_vals[:, 1:] - z_vals[:, :-1]
            d_star = heron_distance_bound(sdf, dists)
            # line search for the smallest beta within the bound
            err0 = opacity_error_bound(beta0, sdf, dists, d_star)
            hi = torch.where(err0 <= self.eps, beta0.expand_as(beta), beta)
            lo = beta0.expand_as(beta)
            for _ in range(self.beta_iters):
                mid = (lo + hi) / 2.0
                ok = opacity_error_bound(mid.unsqueeze(-1), sdf, dists, d_star) <= self.eps
                hi = torch.where(ok, mid, hi)
                lo = torch.where(ok, lo, mid)
            beta = hi
            sigma = laplace_density(sdf, beta.unsqueeze(-1))
            dists_inf = torch.cat([dists, torch.full((R, 1), 1e10, device=dev)], -1)
            free_energy = dists_inf * sigma
            shifted = torch.cat([torch.zeros(R, 1, device=dev), free_energy[:, :-1]], -1)
            transmittance = torch.exp(-torch.cumsum(shifted, dim=-1))
            weights = (1 - torch.exp(-free_energy)) * transmittance
            rounds += 1
            unconverged = bool(beta.max() > beta0)  # the one host sync Algorithm 1 needs per round
            upsample = unconverged and rounds < self.max_total_iters
            if upsample:  # more samples where the error bound is large
                N = self.N_samples_eval
                per_section = torch.exp(-d_star / beta.unsqueeze(-1)) * (dists ** 2.0) / (4 * beta.unsqueeze(-1) ** 2)
                bound_opacity = (torch.exp(torch.cumsum(per_section, dim=-1)).clamp(max=1.0e6) - 1.0) * transmittance[:, :-1]
                pdf = bound_opacity + self.add_tiny
            else:  # final set, proportional to the rendering weights
                N = self.N_samples
                pdf = weights[..., :-1] + 1e-5
            pdf = pdf / pdf.sum(-1, keepdim=True)
            cdf = torch.cat([torch.zeros(R, 1, device=dev), torch.cumsum(pdf, -1)], -1)
            if upsample or not model.training:
                u = torch.linspace(0.0, 1.0, steps=N, device=dev).unsqueeze(0).repeat(R, 1)
            else:
                u = rng["u_final"] if "u_final" in rng else _rand((R, N), dev, self.cpu_rng)
            samples = invert_cdf(cdf, z_vals, u.contiguous())
            if upsample:
                z_vals, order = torch.sort(torch.cat([z_vals, samples], -1), -1)
        self.last_rounds = rounds
        z_samples = samples
        near = torch.full((R, 1), float(self.near), device=dev) if bounds is None else bounds[0].reshape(R, 1)
        far = torch.full((R, 1), float(self.far), device=dev) if bounds is None else bounds[1].reshape(R, 1)
        if self.N_samples_extra > 0:
            if model.training:
                perm = rng["perm"] if "perm" in rng else torch.randperm(z_vals.shape[1])
                pick = perm[: self.N_samples_extra].to(dev)
            else:
                pick = torch.linspace(0, z_vals.shape[1] - 1, self.N_samples_extra, device=dev).long()
            extra = torch.cat([near, far, z_vals[:, pick]], -1)
        else:
            extra = torch.cat([near, far], -1)
        z_vals, _ = torch.sort(torch.cat([z_samples, extra], -1), -1)
        if "eik_idx" in rng:
            eik = rng["eik_idx"].to(dev)
        elif self.cpu_rng:
            eik = torch.randint(z_vals.shape[-1], (R,)).to(dev)
        else:
            eik = torch.randint(z_vals.shape[-1], (R,), device=dev)
        z_samples_eik = torch.gather(z_vals, 1, eik.unsqueeze(-1))
        return z_vals, z_samples_eik
