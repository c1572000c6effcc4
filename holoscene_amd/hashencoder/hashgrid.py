"""HashEncoder -- same constructor, parameters, buffers and forward contract as the
reference module (hashencoder/hashgrid.py:107-166), backed by the gfx950 kernels in
holoscene_amd/csrc/hash_encode.hip.

Differences that do not change results:
  * features are produced point-major ([B, L*C]) by the kernel, so the reference's
    permute+reshape copy (hashgrid.py:44) and the [B,L*C]->[L,B,C] copy of the incoming
    gradient (hashgrid.py:61) disappear;
  * dy_dx is kept level-major ([L,B,D*C]) for coalesced access;
  * work whose result autograd does not ask for is skipped (``ctx.needs_input_grad``):
    e.g. ``autograd.grad(sdf, x, create_graph=True)`` no longer zero-fills and scatters a
    48.8 MB embedding gradient that nobody reads (reference: hashgrid.py:75-82);
  * the double-backward structure is unchanged: the first backward is itself a Function
    whose backward runs the second-backward kernels and, like the reference
    (hashgrid.py:101), returns no gradient for the inputs.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import backend as _be


class _hash_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False):
        embeddings_param = embeddings if isinstance(embeddings, torch.nn.Parameter) else None
        inputs = inputs.contiguous()
        embeddings = embeddings.contiguous()
        offsets = offsets.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = int(base_resolution)
        outputs = torch.empty(B, L * C, device=inputs.device, dtype=inputs.dtype)
        dy_dx = torch.empty(L, B, D * C, device=inputs.device, dtype=inputs.dtype) if calc_grad_inputs else None
        _be._backend.fwd(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, C, L, S, H)
        ctx.calc_grad_inputs = calc_grad_inputs
        ctx.table = embeddings_param if embeddings_param is not None else None
        if ctx.needs_input_grad[1]:
            _be.expect_scatter(ctx.table)
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        need_x = ctx.calc_grad_inputs and ctx.needs_input_grad[0]
        need_e = ctx.needs_input_grad[1]
        table = ctx.table
        if need_e and _be.accumulates_into_grad(table) and not torch.is_grad_enabled():
            B, D, C, L, S, H = ctx.dims
            gx = torch.empty_like(inputs) if need_x else None
            _be._backend.bwd(grad.contiguous(), inputs, offsets, table.grad, B, D, C, L, S, H, dy_dx, gx)
            _be.scatter_done(table)
            return gx, None, None, None, None, None
        grad_inputs, grad_embeddings = _hash_encode_backward.apply(grad.contiguous(), inputs, embeddings, offsets, dy_dx, ctx.dims,
                                                                   need_x, need_e)
        return grad_inputs, grad_embeddings, None, None, None, None


class _hash_encode_backward(Function):
    @staticmethod
    def forward(ctx, grad, inputs, embeddings, offsets, dy_dx, dims, need_x, need_e):
        B, D, C, L, S, H = dims
        grad_inputs = torch.empty_like(inputs) if need_x else None
        grad_embeddings = torch.zeros_like(embeddings) if need_e else None
        if need_x or need_e:
            _be._backend.bwd(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs)
        ctx.save_for_backward(grad, inputs, embeddings, offsets, dy_dx)
        ctx.dims = dims
        return grad_inputs, grad_embeddings

    @staticmethod
    def backward(ctx, grad_grad_inputs, _grad_grad_embeddings):
        grad, inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H = ctx.dims
        if grad_grad_inputs is None or dy_dx is None:
            return None, None, None, None, None, None, None, None
        need_gg = ctx.needs_input_grad[0]
        need_e2 = ctx.needs_input_grad[2]
        grad_grad = torch.empty_like(grad) if need_gg else None
        grad2_embeddings = torch.zeros_like(embeddings) if need_e2 else None
        if need_gg or need_e2:
            _be._backend.bwd2(grad, inputs, offsets, B, D, C, L, S, H, dy_dx, grad_grad_inputs.contiguous(), grad_grad, grad2_embeddings)
        return grad_grad, None, grad2_embeddings, None, None, None, None, None


class _hash_encode_dt(Function):
    """The reference's Function (hashgrid.py:14-101) for its OTHER scalar types -- double (a gradcheck) and half (a caller under autocast: the
    reference casts its inputs with custom_fwd(cast_inputs=torch.half)) -- on the *_dt entry points (csrc/hash_encode_dt.hip): the reference's
    layouts and its order of operations, nothing of the float path's machinery.  inputs and embeddings share the dtype."""

    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False):
        inputs, embeddings, offsets = inputs.contiguous(), embeddings.contiguous(), offsets.contiguous()
        B, D = inputs.shape
        L, C = offsets.shape[0] - 1, embeddings.shape[1]
        S, H = float(np.log2(per_level_scale)), int(base_resolution)
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=inputs.dtype)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=inputs.dtype) if calc_grad_inputs else None
        _be._backend.encode_forward_dt(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, C, L, S, H)
        return outputs.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H = ctx.dims
        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()
        gx, ge = _hash_encode_dt_backward.apply(grad, inputs, embeddings, offsets, dy_dx, ctx.dims)
        return (gx if dy_dx is not None else None), ge, None, None, None, None


class _hash_encode_dt_backward(Function):
    @staticmethod
    def forward(ctx, grad, inputs, embeddings, offsets, dy_dx, dims):
        B, D, C, L, S, H = dims
        grad_inputs = torch.zeros_like(inputs) if dy_dx is not None else None
        grad_embeddings = torch.zeros_like(embeddings)
        _be._backend.encode_backward_dt(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs)
        ctx.save_for_backward(grad, inputs, embeddings, offsets, dy_dx)
        ctx.dims = dims
        return grad_inputs, grad_embeddings

    @staticmethod
    def backward(ctx, grad_grad_inputs, _grad_grad_embeddings):
        grad, inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H = ctx.dims
        if grad_grad_inputs is None or dy_dx is None:
            return None, None, None, None, None, None
        grad_grad, grad2_embeddings = torch.empty_like(grad), torch.zeros_like(embeddings)
        _be._backend.encode_second_backward_dt(grad, inputs, embeddings, offsets, B, D, C, L, S, H, dy_dx, grad_grad_inputs.contiguous().to(grad.dtype),
                                               grad_grad, grad2_embeddings)
        return grad_grad, None, grad2_embeddings, None, None, None       # (no gradient for the inputs: hashgrid.py:101)


def hash_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False):
    """float32: the path everything in this package is built around; float64 / float16 (inputs and table alike): the plain kernels of the
    reference's other instantiations."""
    if inputs.dtype in (torch.float64, torch.float16) and inputs.is_cuda:
        return _hash_encode_dt.apply(inputs, embeddings.to(inputs.dtype), offsets, per_level_scale, base_resolution, calc_grad_inputs)
    return _hash_encode.apply(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs)


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size):
    """Entry offsets of each level (reference: hashgrid.py:127-138)."""
    cap = 2 ** log2_hashmap_size
    offs = [0]
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        offs.append(offs[-1] + min(cap, resolution ** input_dim))
    return np.asarray(offs, dtype=np.int32)


class HashEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None):
        super().__init__()
        if desired_resolution is not None:  # overrides per_level_scale (hashgrid.py:112-113)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.max_params = 2 ** log2_hashmap_size
        offsets = torch.from_numpy(level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size))
        self.register_buffer("offsets", offsets)
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    # ---- a grid of fewer than 16 levels as the fused kernels see it.  The matrix-core kernels of model/network.py address hash features as 16 levels
    # x 2 channels (no L argument: a lane half owns levels 8 h .. 8 h + 7).  A grid with L < 16 levels of 2 channels is handed to them with EMPTY
    # levels behind its own (offsets[l + 1] == offsets[l]): the hash kernels encode an empty level to zeros and scatter nothing into it
    # (csrc/hash_encode.hip), so every consumer sees the stock layout with zero features in the slots the conf does not have -- their weight
    # columns are zero-padded to match (network.py: fused_cols) and receive exactly-zero gradients.
    FUSED_LEVELS = 16

    @property
    def fused_pads(self):
        return self.level_dim == 2 and self.input_dim == 3 and self.num_levels < self.FUSED_LEVELS

    @property
    def fused_num_levels(self):
        return self.FUSED_LEVELS if self.fused_pads else self.num_levels

    @property
    def fused_offsets(self):
        if not self.fused_pads:
            return self.offsets
        pad = getattr(self, "_fused_offsets", None)
        if pad is None or pad.device != self.offsets.device:
            pad = torch.cat([self.offsets, self.offsets[-1:].expand(self.FUSED_LEVELS - self.num_levels)]).contiguous()
            self._fused_offsets = pad        # (not a registered buffer: state dicts keep the reference's keys)
        return pad

    def __repr__(self):
        return (f"HashEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"base_resolution={self.base_resolution} per_level_scale={self.per_level_scale} params={tuple(self.embeddings.shape)}")

    def forward(self, inputs, size=1):
        # inputs in [-size, size] -> [0, 1] (hashgrid.py:158)
        inputs = (inputs + size) / (2 * size)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = hash_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, inputs.requires_grad)
        return outputs.view(prefix_shape + [self.output_dim])


# ------------------------------------------------------------------------------------ many grids, one launch
class _hash_encode_grids(Function):
    """Points of N grids with the same level geometry in ONE launch (hsHashLayout::grid_id): point b looks up table
    ``embeddings[grid_id[b]]``.  Returns the features [B, L*C] and, with ``jacobian=True``, dy_dx [L, B, D*C] as a second
    DIFFERENTIABLE-IN-THE-TABLE output (the value+Jacobian formulation of this package: no gradient flows to the inputs, as in
    hashgrid.py:101).  backward: one fused value+Jacobian scatter into the stacked table gradient."""

    @staticmethod
    def forward(ctx, inputs, grid_id, embeddings, offsets, per_level_scale, base_resolution, jacobian=False):
        ctx.table = embeddings if isinstance(embeddings, torch.nn.Parameter) else None
        if ctx.needs_input_grad[2]:
            _be.expect_scatter(ctx.table)
        inputs = inputs.contiguous()
        grid_id = grid_id.contiguous()
        G, T, C = embeddings.shape
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        S, H = float(np.log2(per_level_scale)), int(base_resolution)
        outputs = torch.empty(B, L * C, device=inputs.device, dtype=inputs.dtype)
        dy_dx = torch.empty(L, B, D * C, device=inputs.device, dtype=inputs.dtype) if jacobian else None
        _be._backend.fwd(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, grids=(grid_id, T))
        ctx.save_for_backward(inputs, grid_id, embeddings, offsets)
        ctx.dims = (B, D, C, L, S, H, T)
        if not jacobian:
            return outputs
        return outputs, dy_dx

    @staticmethod
    def backward(ctx, g_feat, g_dydx=None):
        inputs, grid_id, embeddings, offsets = ctx.saved_tensors
        B, D, C, L, S, H, T = ctx.dims
        g_emb = None
        if ctx.needs_input_grad[2]:
            table = ctx.table
            inplace = _be.accumulates_into_grad(table)
            target = table.grad if inplace else torch.zeros_like(embeddings)
            _be._backend.bwd_jac(None if g_feat is None else g_feat.contiguous(), None if g_dydx is None else g_dydx.contiguous(), inputs, offsets,
                                 target, B, D, C, L, S, H, grids=(grid_id, T))
            g_emb = None if inplace else target
            if inplace:
                _be.scatter_done(table)
        return None, None, g_emb, None, None, None, None


hash_encode_grids = _hash_encode_grids.apply


class BatchedHashEncoder(nn.Module):
    """``num_grids`` HashEncoders of one geometry behind ONE stacked table [G, T, C] -- the per-object grids of
    SingleObjectImplicitNetworkGrid (model/network.py:1880-1883: every object constructs the same HashEncoder) evaluated together.
    ``forward(inputs, grid_id)`` = ``HashEncoder.forward`` of grid ``grid_id[b]`` on point b, bit for bit."""

    def __init__(self, num_grids, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None):
        super().__init__()
        one = HashEncoder(input_dim, num_levels, level_dim, per_level_scale, base_resolution, log2_hashmap_size, desired_resolution)
        self.num_grids, self.input_dim, self.num_levels, self.level_dim = num_grids, input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution, self.output_dim = one.per_level_scale, base_resolution, one.output_dim
        self.register_buffer("offsets", one.offsets.clone())
        self.embeddings = nn.Parameter(torch.empty(num_grids, one.embeddings.shape[0], level_dim).uniform_(-1e-4, 1e-4))

    @classmethod
    def from_encoders(cls, encoders):
        """Stack existing HashEncoders (copies their tables)."""
        e0 = encoders[0]
        self = cls(len(encoders), e0.input_dim, e0.num_levels, e0.level_dim, e0.per_level_scale, e0.base_resolution, e0.log2_hashmap_size)
        self.to(e0.embeddings.device)
        with torch.no_grad():
            for g, e in enumerate(encoders):
                if not torch.equal(e.offsets.cpu(), self.offsets.cpu()):
                    raise ValueError("grids of different level geometry cannot share a launch")
                self.embeddings[g].copy_(e.embeddings)
        return self

    def forward(self, inputs, grid_id, size=1, jacobian=False):
        """inputs [B, D] in [-size, size], grid_id int32 [B] -> features [B, L*C] (and dy_dx [L, B, D*C] w.r.t. the [0,1] coordinate)."""
        x01 = ((inputs + size) / (2 * size)).view(-1, self.input_dim)
        return hash_encode_grids(x01, grid_id.to(torch.int32), self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, jacobian)
