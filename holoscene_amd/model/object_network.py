"""Per-object networks with their own hash grids: ``SingleObjectImplicitNetworkGrid``,
``SingleObjectRenderingNetwork`` and ``ObjectSDFNetwork`` -- same class names, constructor arguments, state-dict keys, methods
and return values as the reference (model/network.py:1835-2032, :2035-2109, :2111-2209).

Each object owns one hash grid looked up in the object's frame ((x - center) / scale, network.py:1947) and one trunk
71 -> 256 -> 256 -> 1 + 256 whose last layer also carries the colour feature vector.  As everywhere in this package the SDF gradient
comes from ONE value+Jacobian pass (three input tangents beside the value) instead of the reference's ``autograd.grad(...,
create_graph=True)`` -- the same function of the parameters, so plain first-order backward yields the reference's parameter gradients
(DESIGN V1) -- the hash lookups and their fused value+Jacobian scatter are the HIP kernels of csrc/hash_encode.hip, the per-ray
compositing is csrc/composite.hip with K = 1, and the sampler is the shared ``ErrorBoundSampler``.
"""
import numpy as np
import torch
import torch.nn as nn

from ..hashencoder.hashgrid import HashEncoder
from .density import LaplaceDensity
from .embedder import Embedder
from .ray_sampler import ErrorBoundSampler
from . import network as _net
from .network import RenderingNetwork
from .fused_ops import WNLinear, _composite, _trunk_input, linear_rows, softplus_tangent


class SingleObjectImplicitNetworkGrid(nn.Module):
    def __init__(self, feature_vector_size=256, d_in=3, d_out=1, dims=(256, 256), geometric_init=True, bias=0.9, skip_in=(4,), weight_norm=True,
                 multires=6, sphere_scale=1.0, base_size=16, end_size=2048, logmap=19, num_levels=16, level_dim=2, divide_factor=1.0,
                 use_grid_feature=True, sigmoid=10, object_center=None, object_scale=None, fg_bg=True):
        super().__init__()
        if not weight_norm or not use_grid_feature:
            raise NotImplementedError("the reference constructs this class with its defaults only (network.py:2120-2124)")
        self.d_out, self.sigmoid, self.sphere_scale = d_out, sigmoid, sphere_scale
        self.divide_factor, self.use_grid_feature, self.fg_bg = divide_factor, use_grid_feature, fg_bg
        self.feature_vector_size = feature_vector_size
        self.grid_feature_dim = num_levels * level_dim
        self.register_buffer("_center", torch.as_tensor(object_center if object_center is not None else [0.0, 0.0, 0.0], dtype=torch.float32).reshape(3),
                             persistent=False)
        self.object_center = object_center
        self.object_scale = 1.0 if object_scale is None else float(object_scale)
        dims = [d_in] + list(dims) + [d_out + feature_vector_size]
        dims[0] += self.grid_feature_dim
        print(f"[INFO]: using hash encoder with {num_levels} levels, each level with feature dim {level_dim}")
        print(f"[INFO]: resolution:{base_size} -> {end_size} with hash map size {logmap}")
        self.encoding = HashEncoder(input_dim=3, num_levels=num_levels, level_dim=level_dim, per_level_scale=2, base_resolution=base_size,
                                    log2_hashmap_size=logmap, desired_resolution=end_size)
        self.embedder = self.embed_fn = None
        if multires > 0:
            self.embedder = Embedder(multires, d_in)
            self.embed_fn = self.embedder.embed
            dims[0] += self.embedder.out_dim - 3
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        if any(l in self.skip_in for l in range(self.num_layers - 1)):
            raise NotImplementedError("skip connections never fire with the reference's three layers (skip_in = [4])")
        for l in range(self.num_layers - 1):
            lin = WNLinear(dims[l], dims[l + 1])
            if geometric_init:      # network.py:1906-1929
                with torch.no_grad():
                    v = lin.weight_v
                    if l == self.num_layers - 2:
                        # only the SDF row is shaped; the feature rows keep nn.Linear's default initialisation
                        sign = 1.0 if fg_bg else -1.0
                        v[:1].normal_(sign * np.sqrt(np.pi) / np.sqrt(dims[l]), 0.0001)
                        lin.bias[:1].fill_(-0.5 * bias if fg_bg else bias)
                    elif multires > 0 and l == 0:
                        lin.bias.zero_()
                        v[:, 3:].zero_()
                        v[:, :3].normal_(0.0, np.sqrt(2) / np.sqrt(dims[l + 1]))
                    else:
                        lin.bias.zero_()
                        v.normal_(0.0, np.sqrt(2) / np.sqrt(dims[l + 1]))
                lin.reset_g()
            setattr(self, "lin" + str(l), lin)
        self.softplus = nn.Softplus(beta=100)
        self.cache_sdf = None
        self.relu = nn.ReLU()

    def _lins(self):
        return [getattr(self, "lin" + str(l)) for l in range(self.num_layers - 1)]

    def _grid_coords(self, x):
        return (x - self._center.to(x.device)) / self.object_scale / self.divide_factor

    def forward(self, input):
        """[B,3] -> [B, d_out + feature_vector_size] (network.py:1945-1965)."""
        feature = self.encoding(self._grid_coords(input))
        inp = torch.cat((self.embed_fn(input) if self.embed_fn is not None else input, feature), dim=-1)
        h = inp
        lins = self._lins()
        for l, lin in enumerate(lins):
            if l < len(lins) - 1:
                h = softplus_tangent(linear_rows(h, lin.weight, None, False).unsqueeze(1), lin.bias).squeeze(1)
            else:
                h = linear_rows(h, lin.weight, None, False).float() + lin.bias
        return h

    def value_and_jacobian(self, x):
        """x [B,3] (constant) -> output [B, d_out + F] and J [B, d_out, 3] = d sdf / d x, differentiable w.r.t. every parameter."""
        x = x.detach()
        enc = self.encoding
        inp = _trunk_input.apply(x, enc.embeddings, enc.offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                                 self.embedder.multires if self.embedder is not None else 0, float(self.divide_factor), torch.float32,
                                 self._center.to(x.device), self.object_scale)                       # [B,4,71]
        return self.rows_to_value_and_jacobian(inp)

    def rows_to_value_and_jacobian(self, inp):
        """The MLP on assembled value + tangent rows [B, 4, 71] (from _trunk_input, or from ObjectSDFNetworkSet's batched lookup)."""
        h = inp
        lins = self._lins()
        for lin in lins[:-1]:
            h = softplus_tangent(linear_rows(h, lin.weight, None, False), lin.bias)
        W = lins[-1].weight
        y = linear_rows(h[:, 0], W, lins[-1].bias, False)                       # value row: every output
        J = linear_rows(h[:, 1:], W[:self.d_out], None, False)                    # tangent rows: only the SDF columns  [B,3,d_out]
        return y, J.transpose(1, 2)

    def gradient(self, x):
        """[d_out * B, 3]: rows k*B..(k+1)*B-1 = d sdf_k / dx (network.py:1967-1988)."""
        _, J = self.value_and_jacobian(x)
        return J.transpose(0, 1).reshape(-1, 3)

    def get_outputs(self, x, beta=None):
        y, J = self.value_and_jacobian(x)
        return y[:, :self.d_out], y[:, self.d_out:], J.sum(dim=1)      # gradient of sum_k sdf_k (grad_outputs = ones, network.py:1997-2004)

    def get_sdf_vals(self, x):
        return self.forward(x)[:, :self.d_out]

    def mlp_parameters(self):
        return [p for lin in self._lins() for p in lin.parameters()]

    def grid_parameters(self, verbose=False):
        return self.encoding.parameters()


class SingleObjectRenderingNetwork(RenderingNetwork):
    def __init__(self, feature_vector_size=256, mode="idr", d_in=9, d_out=3, dims=(256, 256), weight_norm=True, multires_view=4, multires_point=0,
                 multires_normal=0):
        super().__init__(feature_vector_size, mode, d_in, d_out, list(dims), weight_norm, multires_view, multires_point, multires_normal)
        self.set_mlp_precision("fp32")

    def forward(self, points, normals, view_dirs, feature_vectors):
        return super().forward(points, normals, view_dirs, feature_vectors, None)


class ObjectSDFNetwork(nn.Module):
    N_EIK_POINTS = 2048     # network.py:2187

    def __init__(self, center, scale, fg_bg, conf, implicit_kwargs=None, rendering_kwargs=None):
        """implicit_kwargs / rendering_kwargs: optional constructor overrides of the two networks (the reference hard-codes their defaults;
        the parity fixtures use a small hash table, the eight-object one also narrow layers)."""
        super().__init__()
        self.scene_bounding_sphere = 1.0
        self.implicit_network = SingleObjectImplicitNetworkGrid(object_center=center, object_scale=scale, fg_bg=fg_bg, **(implicit_kwargs or {}))
        self.rendering_network = SingleObjectRenderingNetwork(**(rendering_kwargs or {}))
        self.density = LaplaceDensity(**conf.get_config("density"))
        self.ray_sampler = ErrorBoundSampler(self.scene_bounding_sphere, **conf.get_config("ray_sampler"))

    def volume_rendering(self, z_vals, sdf):
        density = self.density(sdf).reshape(-1, z_vals.shape[1])
        dists = z_vals[:, 1:] - z_vals[:, :-1]
        dists = torch.cat([dists, torch.full_like(dists[:, :1], 1e10)], -1)
        free_energy = dists * density
        shifted = torch.cat([torch.zeros_like(free_energy[:, :1]), free_energy[:, :-1]], dim=-1)
        transmittance = torch.exp(-torch.cumsum(shifted, dim=-1))
        return (1 - torch.exp(-free_energy)) * transmittance, transmittance, dists

    def occlusion_opacity(self, z_vals, transmittance, dists, sdf_raw):
        obj_density = self.density(sdf_raw).transpose(0, 1).reshape(-1, dists.shape[0], dists.shape[1])
        return (1 - torch.exp(-dists * obj_density)) * transmittance

    def forward(self, ray_origins, ray_dirs, rng=None, _defer=False):
        """rng: optional explicit draws {'t_rand','u_final','perm','eik_idx' (sampler), 'eik_uniform' [2048,3], 'eik_jitter' [2048+R,3]}."""
        rng = rng or {}
        cam_loc = ray_origins.reshape(-1, 3).contiguous()
        ray_dirs = ray_dirs.reshape(-1, 3).contiguous()
        dev = cam_loc.device
        z_vals, z_samples_eik = self.ray_sampler.get_z_vals(ray_dirs, cam_loc, self, rng=rng)
        N = z_vals.shape[1]
        points_flat = (cam_loc.unsqueeze(1) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
        dirs_flat = ray_dirs.unsqueeze(1).expand(-1, N, -1).reshape(-1, 3)
        net = self.implicit_network
        # Eikonal set (network.py:2187-2199), evaluated in the same value+Jacobian pass as the rendered points
        b = self.scene_bounding_sphere
        e0 = rng["eik_uniform"].to(dev) if "eik_uniform" in rng else torch.empty(self.N_EIK_POINTS, 3, device=dev).uniform_(-b, b)
        near = (cam_loc.unsqueeze(1) + z_samples_eik.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
        eik = torch.cat([e0, near], 0)
        jitter = rng["eik_jitter"].to(dev) if "eik_jitter" in rng else torch.rand_like(eik)
        eik = torch.cat([eik, eik + (jitter - 0.5) * 0.01], 0)
        n_main = points_flat.shape[0]
        prep = (z_vals, points_flat, dirs_flat, torch.cat([points_flat, eik], 0), n_main)
        if _defer:          # ObjectSDFNetworkSet: the value+Jacobian pass of all its objects shares one hash lookup
            return prep
        y, J = net.value_and_jacobian(prep[3])
        return self._finish(prep, y, J)

    def _finish(self, prep, y, J):
        z_vals, points_flat, dirs_flat, _, n_main = prep
        net, dev, N = self.implicit_network, z_vals.device, z_vals.shape[1]
        sdf, feature_vectors, gradients = y[:n_main, :net.d_out], y[:n_main, net.d_out:], J[:n_main].sum(dim=1)
        rgb_flat = self.rendering_network(points_flat, gradients, dirs_flat, feature_vectors)
        if z_vals.is_cuda and _net.ops.COMPOSITE_IMPL == "hip":
            ones = torch.ones(z_vals.shape[0], 1, device=dev)
            weights, _, rgb_values, depth_values, normal_map, _, object_opacity = _composite.apply(
                z_vals, sdf, sdf, rgb_flat, gradients, self.density.get_beta(), ones, float(net.sigmoid))
        else:
            weights, transmittance, dists = self.volume_rendering(z_vals, sdf)
            object_opacity = self.occlusion_opacity(z_vals, transmittance, dists, sdf).sum(-1).transpose(0, 1)
            rgb_values = torch.sum(weights.unsqueeze(-1) * rgb_flat.reshape(-1, N, 3), 1)
            depth_values = torch.sum(weights * z_vals, 1, keepdims=True) / (weights.sum(dim=1, keepdims=True) + 1e-8)
            normals = (gradients / (gradients.norm(2, -1, keepdim=True) + 1e-6)).reshape(-1, N, 3)
            normal_map = torch.sum(weights.unsqueeze(-1) * normals, 1)
        grad_theta = J[n_main:].transpose(0, 1).reshape(-1, 3)
        half = grad_theta.shape[0] // 2
        return {"object_opacity": object_opacity, "rgb_values": rgb_values, "depth_values": depth_values, "normal_map": normal_map,
                "opacity": object_opacity, "grad_theta": grad_theta[:half], "grad_theta_nei": grad_theta[half:]}


class _trunk_input_grids(torch.autograd.Function):
    """_trunk_input for the points of SEVERAL objects at once: point b is looked up in table grid_id[b] of a stacked [G, T, C] table, in that
    object's frame -- ONE batched-over-grids hash launch (csrc/hash_encode.hip: hsHashLayout::grid_id) instead of one per object, and ONE
    fused value+Jacobian scatter on the way back."""

    @staticmethod
    def forward(ctx, x, grid_id, tables, offsets, S, H, nfreq, divide_factor, centers, scales):
        be = _net._be._backend
        x = x.contiguous()
        gid = grid_id.long()
        sc = scales.to(x.device)[gid].unsqueeze(1)                                  # per-point object scale
        x01 = ((((x - centers.to(x.device)[gid]) / sc) / divide_factor + 1.0) / 2.0).contiguous()
        B, D = x01.shape
        G, T, C = tables.shape
        L = offsets.shape[0] - 1
        feat = torch.empty(B, L * C, device=x.device)
        dydx = torch.empty(L, B, D * C, device=x.device)
        flat = tables.detach().reshape(G * T, C)
        be.fwd(x01, flat, offsets, feat, B, D, C, L, S, H, dydx, grids=(grid_id, T))
        jac = (0.5 / (divide_factor * sc)).reshape(1, B, 1)                         # d x01 / d x differs per object
        dydx.mul_(jac)
        out = torch.empty(B, 4, 3 + 6 * nfreq + L * C, device=x.device)
        be.trunk_input_fwd(x, feat, dydx, out, nfreq, L, C, 1.0)
        ctx.save_for_backward(x01, grid_id, tables, offsets, jac)
        ctx.cfg = (B, D, C, L, S, H, nfreq, G, T)
        return out

    @staticmethod
    def backward(ctx, Gr):
        be = _net._be._backend
        x01, grid_id, tables, offsets, jac = ctx.saved_tensors
        B, D, C, L, S, H, nfreq, G, T = ctx.cfg
        if not ctx.needs_input_grad[2]:
            return (None,) * 10
        g_feat = torch.empty(B, L * C, device=Gr.device)
        g_dydx = torch.empty(L, B, D * C, device=Gr.device)
        be.trunk_input_bwd(Gr.contiguous(), g_feat, g_dydx, nfreq, L, C, 1.0)
        g_dydx.mul_(jac)
        target = torch.zeros(G * T, C, device=Gr.device)
        be.bwd_jac(g_feat, g_dydx, x01, offsets, target, B, D, C, L, S, H, grids=(grid_id, T))
        return None, None, target.view(G, T, C), None, None, None, None, None, None, None


class ObjectSDFNetworkSet(nn.Module):
    """Several ObjectSDFNetworks of one grid geometry evaluated together ("many small hash grids"): their tables live in
    ONE stacked parameter [G, T, C] (every member's ``implicit_network.encoding.embeddings`` becomes a view of its slice, so the members keep
    working on their own -- the samplers' SDF sweeps do), and a training pass gathers the value + Jacobian features of ALL objects' rendered and
    Eikonal points with one batched-over-grids launch and scatters their table gradients with one (``_trunk_input_grids``).  Everything else
    of a member's forward -- its sampler, its 71 -> 256 -> 256 -> 257 trunk on library GEMMs, rendering network, compositing -- is the
    member's own code, so each object's outputs and gradients are those of ``ObjectSDFNetwork.forward`` on its rays.
    (The reference's Stage-2 loop imports ObjectSDFNetwork and never constructs one, training/holoscene_train_post.py:58: the per-object
    trunks stay on library GEMMs.)"""

    def __init__(self, nets):
        super().__init__()
        self.nets = nn.ModuleList(nets)
        encs = [n.implicit_network.encoding for n in nets]
        e0 = encs[0]
        for e in encs:
            if not torch.equal(e.offsets.cpu(), e0.offsets.cpu()) or e.embeddings.shape != e0.embeddings.shape:
                raise ValueError("the members' grids must share one level geometry")
        self.tables = nn.Parameter(torch.stack([e.embeddings.detach() for e in encs]))
        self._bind()

    def _bind(self):
        for g, n in enumerate(self.nets):       # the members' tables: views of the stacked parameter's storage (values shared, no copy)
            n.implicit_network.encoding.embeddings.data = self.tables.data[g]

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._bind()
        return out

    def table_grads(self):
        """Per member, the gradient of its table (a view of the stacked gradient)."""
        return [None if self.tables.grad is None else self.tables.grad[g] for g in range(len(self.nets))]

    def forward(self, ray_origins, ray_dirs, rngs=None):
        """ray_origins / ray_dirs / rngs: one entry per member -> list of ObjectSDFNetwork.forward's output dictionaries."""
        rngs = rngs or [None] * len(self.nets)
        preps = [n(o, d, rng=r, _defer=True) for n, o, d, r in zip(self.nets, ray_origins, ray_dirs, rngs)]
        xs = [p[3] for p in preps]
        dev = xs[0].device
        counts = [x.shape[0] for x in xs]
        grid_id = torch.cat([torch.full((c,), g, dtype=torch.int32, device=dev) for g, c in enumerate(counts)])
        nets = [n.implicit_network for n in self.nets]
        n0, e0 = nets[0], nets[0].encoding
        centers = torch.stack([n._center.to(dev) for n in nets])
        scales = torch.tensor([float(n.object_scale) for n in nets], device=dev)
        inp = _trunk_input_grids.apply(torch.cat(xs, 0).detach(), grid_id, self.tables, e0.offsets, float(np.log2(e0.per_level_scale)),
                                       int(e0.base_resolution), n0.embedder.multires if n0.embedder is not None else 0, float(n0.divide_factor),
                                       centers, scales)
        outs, at = [], 0
        for n, prep, c in zip(self.nets, preps, counts):
            y, J = n.implicit_network.rows_to_value_and_jacobian(inp[at:at + c])
            outs.append(n._finish(prep, y, J))
            at += c
        return outs
