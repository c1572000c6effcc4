"""Stage-1 networks: ObjectImplicitNetworkGrid, RenderingNetwork, HoloSceneNetwork.

Same class names, constructor arguments, state-dict keys and public methods as the reference
(model/network.py:19-532, :535-614, :748-971, :1803-1824), re-designed around what the MI355X
kernels want:

  * value + Jacobian in ONE pass.  The reference obtains d sdf/dx by reverse-mode autograd with
    ``create_graph=True`` -- once for the rendered points (network.py:293-299) and K+1 times in a
    Python loop for the Eikonal set (:227-253) -- and then differentiates those graphs a second time
    in ``loss.backward()``.  Here the SDF trunk carries three input tangents next to the value
    (a 4-row GEMM per point), which yields the full Jacobian d sdf_k/dx for all K objects at once;
    the min-SDF gradient is a gather of the arg-min row.  The result is the same function of the
    parameters, so ordinary first-order backward gives the same parameter gradients.
  * the hash encoders run as ``hash_encode_jac``: features and dy_dx out, one fused scatter pass
    back (csrc/hash_encode.hip: k_hash_bwd_jac).
  * SDF-only queries (the sampler's sweeps) skip the colour grid + colour MLP that the reference
    evaluates and discards (network.py:177-179, 305-311).
  * no per-iteration device->host syncs besides the sampler's convergence test.
"""
import contextlib
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hashencoder.hashgrid import HashEncoder
from ..hashencoder import backend as _be
from ..utils import rend_util
from .density import LaplaceDensity, laplace_density
from .embedder import Embedder
from .ray_sampler import ErrorBoundSampler


# ------------------------------------------------------------------------------------------------
from . import fused_ops as ops
from .fused_ops import softplus_tangent      # noqa: F401  (re-exported)
from .fused_ops import (WNLinear, _Outputs, _composite, _fused_trunk, _render_input, _split_value_jacobian, _trunk_input, _trunk_rr32, _w2_planes, default_mlp_precision, effective_weights, fused_appearance, fused_cols, iteration_prologue, linear_rows, shared_effective_weights, trunk_render)      # noqa: F401
from .fused_ops import (_hash_encode_jac, hash_encode_jac, _xp_columns, _xp_columns32, _wgrad_rows, _wgrad_rows_many, _trunk_fwd_core, _trunk_bwd_core, _fused_trunk_render, _rr_slices, _pair_slices, _rows_slices, _tp_colsum, _trunk_render_rr, _posenc6, _fused_appearance, _xa_columns, _fused_appearance_wave, _split_rows, _linear_rows, _weight_norm_many, flat_grad_view, flat_grad_target, _DirectGrad, _iter_prologue, assert_relays_consumed, _softplus_tangent, softplus100, softplus100_grad)      # noqa: F401  (re-exported: callers reach them as network.<name>)


class ObjectImplicitNetworkGrid(nn.Module):
    def __init__(self, feature_vector_size, sdf_bounding_sphere, d_in, d_out, dims, geometric_init=True, bias=1.0, skip_in=(),
                 weight_norm=True, multires=0, sphere_scale=1.0, inside_outside=False, base_size=16, end_size=2048, logmap=19,
                 num_levels=16, level_dim=2, divide_factor=1.5, use_grid_feature=True, sigmoid=20, color_grid_feature=False):
        super().__init__()
        if not weight_norm or not use_grid_feature:
            raise NotImplementedError("Stage-1 configs always use weight_norm=True and use_grid_feature=True (confs/*/*.conf)")
        self.d_out = d_out
        self.sigmoid = sigmoid
        self.sdf_bounding_sphere = sdf_bounding_sphere
        self.sphere_scale = sphere_scale
        self.color_grid_feature = color_grid_feature
        self.divide_factor = divide_factor
        self.use_grid_feature = use_grid_feature
        self.feature_vector_size = feature_vector_size
        dims = [d_in] + list(dims) + [d_out if color_grid_feature else d_out + feature_vector_size]

        print(f"[INFO]: using hash encoder with {num_levels} levels, each level with feature dim {level_dim}")
        print(f"[INFO]: resolution:{base_size} -> {end_size} with hash map size {logmap}")
        self.encoding = HashEncoder(input_dim=3, num_levels=num_levels, level_dim=level_dim, per_level_scale=2,
                                    base_resolution=base_size, log2_hashmap_size=logmap, desired_resolution=end_size)
        self.grid_feature_dim = num_levels * level_dim
        dims[0] += self.grid_feature_dim
        if color_grid_feature:
            self.color_encoding = HashEncoder(input_dim=3, num_levels=num_levels, level_dim=level_dim, per_level_scale=2,
                                              base_resolution=base_size, log2_hashmap_size=logmap, desired_resolution=end_size)
            self.color_grid_feature_dim = num_levels * level_dim
            self.color_grid_feature_map_mlp = nn.Sequential(nn.Linear(self.color_grid_feature_dim, 256), nn.ReLU(),
                                                            nn.Linear(256, feature_vector_size))
        self.embedder = None
        self.embed_fn = None
        if multires > 0:
            self.embedder = Embedder(multires, d_in)
            self.embed_fn = self.embedder.embed
            dims[0] += self.embedder.out_dim - 3
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if l + 1 in self.skip_in else dims[l + 1]
            lin = WNLinear(dims[l], out_dim)
            if geometric_init:  # compositional-scene geometric init (network.py:135-156)
                with torch.no_grad():
                    v = lin.weight_v
                    if l == self.num_layers - 2:
                        # channel 0 = background (positive inside the room), others = objects of half radius
                        v[:1].normal_(-math.sqrt(math.pi) / math.sqrt(dims[l]), 0.0001)
                        v[1:].normal_(math.sqrt(math.pi) / math.sqrt(dims[l]), 0.0001)
                        lin.bias[:1].fill_(bias)
                        lin.bias[1:].fill_(-0.5 * bias)
                    elif multires > 0 and l == 0:
                        lin.bias.zero_()
                        v[:, 3:].zero_()
                        v[:, :3].normal_(0.0, math.sqrt(2) / math.sqrt(out_dim))
                    elif multires > 0 and l in self.skip_in:
                        lin.bias.zero_()
                        v.normal_(0.0, math.sqrt(2) / math.sqrt(out_dim))
                        v[:, -(dims[0] - 3):].zero_()
                    else:
                        lin.bias.zero_()
                        v.normal_(0.0, math.sqrt(2) / math.sqrt(out_dim))
                lin.reset_g()
            setattr(self, "lin" + str(l), lin)
        self.softplus = nn.Softplus(beta=100)
        self.cache_sdf = None
        # a grid of fewer than 16 levels (x 2 channels) on the fused 16-level kernels: empty levels behind its own (HashEncoder.fused_offsets), zero
        # columns behind the feature columns of the layers that read them (effective_weights / fused_cols).  Only when the features are the last
        # columns of the first layer's input, as the stock layout has them: [x | positional encoding | hash features]
        if self.encoding.fused_pads and multires == 6 and self.lin0.in_features == 39 + self.grid_feature_dim and 0 not in self.skip_in:
            self.lin0.fused_cols = 39 + 2 * HashEncoder.FUSED_LEVELS
            if color_grid_feature:
                self.color_grid_feature_map_mlp[0].fused_cols = 2 * HashEncoder.FUSED_LEVELS
        self.set_mlp_precision(default_mlp_precision())

    def set_mlp_precision(self, precision):
        """'fp32' (the reference's precision, parity mode) or 'bf16' (bf16 matrix cores, fp32 accumulate)."""
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"mlp precision must be 'fp32' or 'bf16', got {precision!r}")
        self.mlp_bf16 = precision == "bf16"
        for lin in self._lins():
            lin.bf16 = self.mlp_bf16

    # ---------------------------------------------------------------- building blocks
    def _lins(self):
        return [getattr(self, "lin" + str(l)) for l in range(self.num_layers - 1)]

    def _color_features(self, x):
        mlp = self.color_grid_feature_map_mlp
        h = self.color_encoding(x / self.divide_factor)
        h = torch.relu(linear_rows(h, mlp[0].weight, mlp[0].bias, self.mlp_bf16))
        return linear_rows(h, mlp[2].weight, mlp[2].bias, self.mlp_bf16)

    def _trunk(self, x):
        """SDF trunk only: x [B,3] -> [B, lin_last.out] f32 (no colour branch)."""
        feature = self.encoding(x / self.divide_factor)
        inp = torch.cat((self.embed_fn(x) if self.embed_fn is not None else x, feature), dim=-1)
        h = inp
        lins = self._lins()
        for l, lin in enumerate(lins):
            if l in self.skip_in:
                h = torch.cat([h, inp.to(h.dtype)], 1) / np.sqrt(2)
            if l < len(lins) - 1:
                h = ops.softplus_tangent(linear_rows(h, lin.weight, None, self.mlp_bf16).unsqueeze(1), lin.bias).squeeze(1)
            else:
                h = linear_rows(h, lin.weight, None, self.mlp_bf16).float() + lin.bias
        return h

    # ---------------------------------------------------------------- fused matrix-core inference (bf16 mode)
    def fused_trunk_blockers(self):
        """Why the hand-written matrix-core kernels cannot take this trunk (empty list: they can) -- the shape is whatever the conf says
        (reference: model/network.py:19-167), the kernels are built for the stock one."""
        lins = self._lins()
        why = []
        if len(lins) != 3:
            why.append(f"{len(lins)} linear layers (kernels: 3)")
            return why
        if self.embedder is None or self.embedder.multires != 6:
            why.append(f"multires = {getattr(self.embedder, 'multires', 0)} (kernels: 6 frequencies)")
        if not self._stock_grid(padded=True) or not self._stock_inputs(padded=True):
            enc = self.encoding
            why.append(f"hash grid {getattr(enc, 'num_levels', '?')} levels x {getattr(enc, 'level_dim', '?')} channels (kernels: <= 16 levels x 2, features last in the first layer's input)")
        if not (lins[0].out_features == 256 and lins[1].in_features == 256 and lins[1].out_features == 256):
            why.append(f"hidden widths {lins[0].out_features}, {lins[1].out_features} (kernels: 256, 256)")
        if lins[2].out_features > 64:
            why.append(f"d_out = {lins[2].out_features} (kernels: <= 64)")
        if any(l in self.skip_in for l in range(3)):
            why.append(f"skip connection into layer {sorted(l for l in self.skip_in if l < 3)}")
        return why

    def _fused_trunk_supported(self, x):
        return self.mlp_bf16 and x.is_cuda and not self.fused_trunk_blockers()

    def _rr32_supported(self, x):
        """fp32 trunk of the stock shape (71 -> 256 -> 256 -> K, softplus 100, no skip, 16 x 2 hash features, 6 frequencies) on the device"""
        lins = self._lins()
        return (not self.mlp_bf16 and x.is_cuda and len(lins) == 3 and self.embedder is not None and self.embedder.multires == 6
                and self.grid_feature_dim == 32 and self._stock_grid() and lins[0].in_features == 71 and lins[2].out_features == self.d_out
                and not any(l in self.skip_in for l in range(3)))

    def _stock_grid(self, padded=False):
        """The fused MLP kernels address hash features as 16 levels x 2 channels (csrc/sdf_mlp2.hip, trunk_mlp2.hip: no L / C arguments);
        any other split of the 32 features (8 x 4, 32 x 1) takes the library-GEMM path.  padded: also a grid of FEWER levels x 2 channels whose
        first layer pads its columns (lin0.fused_cols) -- the bf16 kernels then see empty levels behind the conf's own."""
        enc = self.encoding
        if padded and getattr(enc, "fused_pads", False) and getattr(self.lin0, "fused_cols", None) is not None:
            return True
        return getattr(enc, "num_levels", None) == 16 and getattr(enc, "level_dim", None) == 2 and getattr(enc, "input_dim", 3) == 3

    def _stock_inputs(self, padded=False):
        """first layer = [x | 6 octaves | 16 x 2 hash features] = 71 columns (padded: or fewer feature columns, zero-padded to that)"""
        l0 = self._lins()[0]
        if padded and getattr(l0, "fused_cols", None) == 71:
            return self.grid_feature_dim == 2 * self.encoding.num_levels and l0.in_features == 39 + self.grid_feature_dim
        return self.grid_feature_dim == 32 and l0.in_features == 71

    def _fused_sdf_supported(self, x):
        return not torch.is_grad_enabled() and (self._fused_trunk_supported(x) or self._fused_sdf32_supported(x))

    def _fused_sdf32_supported(self, x):
        """The no-grad SDF queries of the fp32 configuration through csrc/sdf_mlp32.hip (fp32 operands on the fp32 matrix cores): the stock
        trunk shape (or a grid of fewer levels padded to it: _stock_grid), d_out <= 32.  HOLOSCENE_FP32_SDF=gemm keeps the library GEMMs."""
        lins = self._lins()
        return (ops.FP32_SDF == "mfma" and not self.mlp_bf16 and x.is_cuda and len(lins) == 3 and self.embedder is not None and self.embedder.multires == 6
                and self._stock_grid(padded=True) and self._stock_inputs(padded=True) and lins[0].out_features == 256
                and lins[1].in_features == 256 and lins[1].out_features == 256 and lins[2].out_features == self.d_out <= 32
                and not any(l in self.skip_in for l in range(3)))

    def _packed_weights(self):
        """bf16 images of the three weight-normalised matrices in the layout csrc/sdf_mlp.hip reads (rebuilt per call:
        three tiny cast kernels; the weights change every optimiser step)."""
        if getattr(self, "_packed_cache", None) is not None:
            return self._packed_cache
        l0, l1, l2 = self._lins()
        dev, bf = l0.weight_v.device, torch.bfloat16
        n2 = 32 * ((l2.out_features + 31) // 32)
        w0, w1 = torch.empty(256, 96, device=dev, dtype=bf), torch.empty(256, 256, device=dev, dtype=bf)
        with torch.no_grad():
            f0, f1, f2 = effective_weights([l0, l1, l2])
        # k_sdf_mlp works in the scaled activation domain t = 100*log2(e)*v (csrc/sdf_mlp.hip, softplus_scaled): the factor goes into
        # W0 (the kernel scales the biases), its inverse ln2/100 into W2 (two bf16 planes: _w2_planes); W1 needs none
        act = 100.0 * 1.4426950408889634
        with torch.no_grad():
            w2 = _w2_planes(f2.detach().float(), l2.out_features, n2, 1.0 / act)
        _be._backend.pack_bf16([(f0, w0, 0, 0, 256, l0.in_features, False, act), (f1, w1, 0, 0, 256, 256, False)])
        self._packed_cache = (w0, l0.bias.detach().float().contiguous(), w1, l1.bias.detach().float().contiguous(), w2,
                              l2.bias.detach().float().contiguous())
        return self._packed_cache

    def _packed_weights2(self):
        """Fragment-order images for the wave-tile kernel (csrc/sdf_mlp2.hip), one pack launch per parameter state."""
        if getattr(self, "_packed_cache2", None) is not None:
            return self._packed_cache2
        if ops._SCOPE.packs is not None and ops._SCOPE.packs.get("net") == id(self) and ops._SCOPE.packs["sdf"] is not None:
            return ops._SCOPE.packs["sdf"]            # (not cached on the module: valid inside this iteration_prologue() only)
        l0, l1, l2 = self._lins()
        with torch.no_grad():
            f0, f1, f2 = effective_weights([l0, l1, l2])
        self._packed_cache2 = _be._backend.sdf_mlp2_pack(f0.detach().float().contiguous(), l0.bias.detach().float().contiguous(),
                                                         f1.detach().float().contiguous(), l1.bias.detach().float().contiguous(),
                                                         f2.detach().float().contiguous(), l2.bias.detach().float().contiguous(), l2.out_features)
        return self._packed_cache2

    def _packed_weights2_wide(self):
        """33 <= d_out <= 64: two packs of the wave-tile kernel's images, the last layer's rows 0..31 and 32.. (hs_sdf_mlp2_fwd_wide)."""
        if getattr(self, "_packed_cache2w", None) is not None:
            return self._packed_cache2w
        l0, l1, l2 = self._lins()
        with torch.no_grad():
            f0, f1, f2 = effective_weights([l0, l1, l2])
        c = lambda t: t.detach().float().contiguous()  # noqa: E731
        K = l2.out_features
        pk = _be._backend.sdf_mlp2_pack
        self._packed_cache2w = (pk(c(f0), c(l0.bias), c(f1), c(l1.bias), c(f2[:32]), c(l2.bias[:32]), 32),
                                pk(c(f0), c(l0.bias), c(f1), c(l1.bias), c(f2[32:]), c(l2.bias[32:]), K - 32))
        return self._packed_cache2w

    def _packed_weights32(self):
        """fp32 operand images of csrc/sdf_mlp32.hip, one pack launch per parameter state."""
        if getattr(self, "_packed_cache32", None) is not None:
            return self._packed_cache32
        l0, l1, l2 = self._lins()
        with torch.no_grad():
            f0, f1, f2 = effective_weights([l0, l1, l2])
        c = lambda t: t.detach().float().contiguous()  # noqa: E731
        self._packed_cache32 = _be._backend.sdf_mlp32_pack(c(f0), c(l0.bias), c(f1), c(l1.bias), c(f2), c(l2.bias), l2.out_features)
        return self._packed_cache32

    def _sdf_mlp(self, x, feat, d_out, select, out, raw, gate, lm):
        # lm: False = fp32 [B, 32], True = fp32 [16, B, 2], 2 = int32 [16, B] bf16 words (wave-tile kernel only)
        """The fused SDF trunk on already gathered hash features: wave-tile kernel for d_out <= 32, workgroup-tile kernel otherwise."""
        be = _be._backend
        if not self.mlp_bf16:      # fp32 operands on the fp32 matrix cores (csrc/sdf_mlp32.hip)
            be.sdf_mlp32_fwd(x, feat, self._packed_weights32(), d_out, select, out, raw, gate=gate, feat_level_major=bool(lm))
        elif ops.SDF_MLP_IMPL == "wave" and d_out <= 32:
            be.sdf_mlp2_fwd(x, feat, self._packed_weights2(), d_out, select, out, raw, gate=gate, feat_level_major=lm)
        elif ops.SDF_MLP_IMPL == "wave" and d_out <= 64 and ops.SDF_WIDE:
            # 33..64 objects: the wave-tile kernel with the last layer's second 32-row tile after the first (36 us per sweep where the
            # workgroup-tile kernel below takes 65)
            pa, pb = self._packed_weights2_wide()
            be.sdf_mlp2_fwd_wide(x, feat, pa, pb, d_out, select, out, raw, gate=gate, feat_level_major=lm)
        elif ops.SDF_MLP_IMPL in ("wave", "tile"):
            w0, b0, w1, b1, w2, b2 = self._packed_weights()
            be.sdf_mlp_fwd(x, feat, w0, b0, w1, b1, w2, b2, d_out, select, out, raw, gate=gate, feat_level_major=lm)
        else:
            raise RuntimeError(f"unknown HOLOSCENE_SDF_MLP_IMPL={ops.SDF_MLP_IMPL!r}")

    def invalidate_packed_weights(self):
        """The packed bf16 images are valid for one parameter state; the sampler drops them at the start of every call."""
        self._packed_cache = None
        self._packed_cache2 = None
        self._packed_cache2w = None
        self._packed_cache32 = None

    def _sdf_fused(self, x, select=-1, want_raw=False):
        """min_k sdf_k (select -1), sdf_select (int) or the minimum over an object list [B,1] (and raw [B, d_out]) through csrc/sdf_mlp.hip."""
        x = x.contiguous().float()
        B = x.shape[0]
        enc = self.encoding
        L, C = enc.fused_num_levels, enc.level_dim        # (a grid of fewer levels: empty ones behind its own, HashEncoder.fused_offsets)
        lm = L == 16 and C == 2        # level-major features: coalesced stores in the gather kernel (see sdf_along_rays)
        x01 = ((x / self.divide_factor + 1.0) / 2.0).contiguous()
        feat = torch.empty((L, B, C) if lm else (B, L * C), device=x.device)
        be = _be._backend
        be.fwd(x01, enc.embeddings, enc.fused_offsets, feat, B, 3, C, L, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), None,
               level_major=lm)
        d_out = self._lins()[2].out_features
        out = torch.empty(B, 1, device=x.device)
        raw = torch.empty(B, d_out, device=x.device) if want_raw else None
        self._sdf_mlp(x, feat, d_out, select, out, raw, None, lm)
        return out, raw

    def sdf_along_rays(self, cam_loc, ray_dirs, z, select=-1, gate=None):
        """Scene SDF (select < 0: min over objects; int: that object's; list: min over those objects) at cam_loc + z*ray_dirs, [R,S] -> [R,S]: the sampler's
        per-round query (ray_sampler.py:151-157) in three launches -- positions, hash encode, fused matrix-core trunk.
        gate: optional device-side launch gate (backend.hsGate) shared by the three kernels."""
        R, S = z.shape
        dev = z.device
        x = torch.empty(R * S, 3, device=dev)
        x01 = torch.empty(R * S, 3, device=dev)
        _be._backend.ray_points(cam_loc.contiguous(), ray_dirs.contiguous(), z.contiguous(), x, x01, float(self.divide_factor), gate=gate)
        return self.sdf_at_points(x, x01, R, S, select, gate=gate)

    def sdf_at_points(self, x, x01, R, S, select=-1, gate=None):
        """sdf_along_rays after its positions launch: x [R*S,3] world positions, x01 their hash-grid coordinates (as hs_ray_points,
        hs_ray_setup or hs_sampler_draw_step wrote them) -> [R,S]."""
        dev = x.device
        be = _be._backend
        enc = self.encoding
        L, C = enc.fused_num_levels, enc.level_dim
        # level-major features [L, R*S, C]: the gather kernel's stores become fully coalesced (point-major 8-byte pieces at a
        # 128-byte stride were written 4x, PMC WRITE_SIZE 66 MB for 17 MB), and the MFMA kernel reads 8-byte runs per level
        lm = L == 16 and C == 2
        d_out = self._lins()[2].out_features
        # ... and as bf16 words [L, R*S] when the wave-tile trunk kernel follows: it rounds the features to bf16 anyway (same rounding:
        # identical results), so the gather writes and the trunk reads half the bytes
        words = lm and ops.SDF_FEAT_BF16 and ops.SDF_MLP_IMPL == "wave" and d_out <= (64 if ops.SDF_WIDE else 32) and enc.embeddings.shape[1] == 2 and self.mlp_bf16
        if words and ops.SDF_SWEEP_FUSED and enc.level_dim == 2 and enc.fused_offsets.numel() == 17 and (d_out <= 32 or ops.SDF_WIDE):
            # ONE launch per sweep: every lane of the trunk's wave tile gathers the eight levels of its own point (csrc/sdf_mlp2.hip: k_sdf_mlp2<., true>);
            # bit-identical to the gather + trunk pair below
            out = torch.empty(R, S, device=dev)
            if d_out <= 32:
                pa, pb = self._packed_weights2(), None
            else:
                pa, pb = self._packed_weights2_wide()
            be.sdf_sweep_fwd(x, x01, enc.embeddings, enc.fused_offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), pa, pb, d_out, select,
                             out, None, gate=gate)
            return out
        if words:
            feat = torch.empty(L, R * S, device=dev, dtype=torch.int32)
            be.fwd(x01, enc.embeddings, enc.fused_offsets, feat, R * S, 3, C, L, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), None,
                   gate=gate, level_major=True, out_bf16=True)
        else:
            feat = torch.empty((L, R * S, C) if lm else (R * S, L * C), device=dev)
            be.fwd(x01, enc.embeddings, enc.fused_offsets, feat, R * S, 3, C, L, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), None,
                   gate=gate, level_major=lm)
        out = torch.empty(R, S, device=dev)
        self._sdf_mlp(x, feat, d_out, select, out, None, gate, 2 if words else lm)
        return out

    def sdf_and_jacobian(self, x):
        """x [B,3] (treated as constant) -> y [B,K'], J [B,K',3] with J[b,k,:] = d y_k / d x.
        Differentiable w.r.t. every parameter by plain first-order autograd.

        Each point carries 4 rows through the trunk: the value and its three input tangents.  A Linear acts
        on all rows alike (one GEMM with M = 4B); Softplus becomes the fused `ops.softplus_tangent` stage."""
        x = x.detach()
        enc = self.encoding
        if ops.TRUNK_IMPL == "mfma" and self._fused_trunk_supported(x):
            l0, l1, l2 = self._lins()
            return _fused_trunk.apply(x, enc.embeddings, enc.fused_offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                                      self.embedder.multires, float(self.divide_factor), fused_cols(l0.weight, l0), l0.bias, l1.weight, l1.bias,
                                      l2.weight, l2.bias)
        inp = _trunk_input.apply(x, enc.embeddings, enc.offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                                 self.embedder.multires if self.embedder is not None else 0, float(self.divide_factor),
                                 torch.bfloat16 if self.mlp_bf16 else torch.float32)                      # [B,4,F]
        h = inp
        lins = self._lins()
        for l, lin in enumerate(lins):
            if l in self.skip_in:
                h = torch.cat([h, inp.to(h.dtype)], 2) / np.sqrt(2)
            out = linear_rows(h, lin.weight, None, self.mlp_bf16)        # one GEMM, M = 4B
            if l < len(lins) - 1:
                h = ops.softplus_tangent(out, lin.bias)
            else:
                return _split_value_jacobian.apply(out, lin.bias)

    # ---------------------------------------------------------------- reference API
    def forward(self, input):
        x = self._trunk(input)
        if self.color_grid_feature:
            x = torch.cat([x, self._color_features(input).float()], dim=-1)
        return x

    def _min_sdf(self, sdf_raw):
        sdf, indices = sdf_raw.min(dim=-1, keepdim=True)
        return sdf, indices

    def gradient(self, x):
        """[(K+1)*B, 3]: rows k*B..(k+1)*B-1 = d sdf_k/dx, last B rows = d min_k sdf_k/dx (network.py:212-254)."""
        y, J = self.sdf_and_jacobian(x)
        y, J = y[:, :self.d_out], J[:, :self.d_out]
        _, idx = self._min_sdf(y)
        g_min = torch.gather(J, 1, idx.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
        return torch.cat([J.transpose(0, 1).reshape(-1, 3), g_min], 0)

    def gradient_obj_i(self, x, obj_i):
        _, J = self.sdf_and_jacobian(x)
        return J[:, obj_i]

    def _outputs(self, x):
        y, J = self.sdf_and_jacobian(x)
        if self.color_grid_feature:
            sdf_raw, feature_vectors = y, self._color_features(x)
        else:
            sdf_raw, feature_vectors = y[:, :self.d_out], y[:, self.d_out:]
            J = J[:, :self.d_out]
        sdf, idx = self._min_sdf(sdf_raw)
        gradients = torch.gather(J, 1, idx.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
        return sdf, feature_vectors, gradients, sdf_raw, idx, J

    def get_outputs(self, x, beta=None):
        sdf, feature_vectors, gradients, sdf_raw, _, _ = self._outputs(x)
        if beta is None:
            semantic = self.sigmoid * torch.sigmoid(-self.sigmoid * sdf_raw)
        else:
            semantic = laplace_density(sdf_raw, beta)
        return sdf, feature_vectors, gradients, semantic, sdf_raw

    def get_outputs_and_indices(self, x):
        sdf, feature_vectors, gradients, sdf_raw, idx, _ = self._outputs(x)
        semantic = self.sigmoid * torch.sigmoid(-self.sigmoid * sdf_raw)
        return sdf, feature_vectors, gradients, semantic, sdf_raw, idx

    def get_specific_outputs(self, x, idx):
        sdf, feature_vectors, gradients, sdf_raw, _, _ = self._outputs(x)
        semantic = self.sigmoid * torch.sigmoid(-self.sigmoid * sdf_raw)
        return sdf, feature_vectors, gradients, semantic, sdf_raw[:, idx]

    # ---- object-subset variants used by the Stage-2/3 entry points of HoloSceneNetwork (network.py:359-459).  One value+Jacobian
    # pass gives every per-object SDF and gradient; a variant only chooses which columns the minimum runs over.
    def _subset_min(self, sdf_raw, J, cols):
        """min over the object columns `cols` (None = all): value [B,1], the gradient of that minimum [B,3]."""
        y = sdf_raw if cols is None else sdf_raw[:, cols]
        Jc = J if cols is None else J[:, cols]
        sdf, k = y.min(dim=-1, keepdim=True)
        return sdf, torch.gather(Jc, 1, k.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)

    def _value_jacobian_features(self, x, want_features=True):
        y, J = self.sdf_and_jacobian(x)
        if self.color_grid_feature:
            return y[:, :self.d_out], J[:, :self.d_out], (self._color_features(x) if want_features else None)
        return y[:, :self.d_out], J[:, :self.d_out], y[:, self.d_out:]

    def get_multi_specific_outputs(self, x, idxs):
        """Scene SDF / gradient / semantics over ALL objects + the minimum over the objects `idxs` (network.py:359-385)."""
        sdf_raw, J, fv = self._value_jacobian_features(x)
        sdf, gradients = self._subset_min(sdf_raw, J, None)
        semantic = self.sigmoid * torch.sigmoid(-self.sigmoid * sdf_raw)
        return sdf, fv, gradients, semantic, sdf_raw[:, idxs].min(dim=-1, keepdim=True)[0]

    def get_only_multi_specific_outputs(self, x, idxs):
        """Everything restricted to the objects `idxs` (network.py:387-406)."""
        sdf_raw, J, fv = self._value_jacobian_features(x)
        sdf, gradients = self._subset_min(sdf_raw, J, idxs)
        semantic = self.sigmoid * torch.sigmoid(-self.sigmoid * sdf_raw[:, idxs])
        return sdf, fv, gradients, semantic

    def get_multi_specific_outputs_subset_objs(self, x, idxs, subset_idxs):
        """SDF / gradient / semantics over `subset_idxs` + the minimum over `idxs` (network.py:408-435)."""
        sdf_raw, J, fv = self._value_jacobian_features(x)
        sdf, gradients = self._subset_min(sdf_raw, J, subset_idxs)
        semantic = self.sigmoid * torch.sigmoid(-self.sigmoid * sdf_raw[:, subset_idxs])
        return sdf, fv, gradients, semantic, sdf_raw[:, idxs].min(dim=-1, keepdim=True)[0]

    def get_specific_outputs_nm(self, x, idx):
        """One object's own SDF and gradient, no minimum (network.py:437-458); semantics over all objects."""
        sdf_raw, J, fv = self._value_jacobian_features(x)
        semantic = self.sigmoid * torch.sigmoid(-self.sigmoid * sdf_raw)
        return sdf_raw[:, idx], fv, J[:, idx], semantic

    def get_sdf_raw(self, x):
        if self.color_grid_feature and self._fused_sdf_supported(x):
            return self._sdf_fused(x, want_raw=True)[1]
        return self._trunk(x)[:, :self.d_out]

    def get_sdf_vals(self, x):
        if self.color_grid_feature and self._fused_sdf_supported(x):
            return self._sdf_fused(x)[0]
        return self._min_sdf(self.get_sdf_raw(x))[0]

    def get_object_sdf_vals(self, x, idx):
        if self.color_grid_feature and isinstance(idx, int) and self._fused_sdf_supported(x):
            return self._sdf_fused(x, select=idx)[0].squeeze(-1)
        return self._trunk(x)[:, idx]

    def get_multi_object_sdf_vals(self, x, idxs):
        if self.color_grid_feature and self._fused_sdf_supported(x):    # the subset minimum inside the fused matrix-core sweep (object bit mask)
            return self._sdf_fused(x, select=list(idxs))[0]
        return self._trunk(x)[:, idxs].min(dim=-1, keepdim=True)[0]

    def get_sdf_vals_and_sdfs(self, x):
        sdf_raw = self.get_sdf_raw(x)
        return self._min_sdf(sdf_raw)[0], sdf_raw

    def get_shift_sdf_raw(self, x):
        """Raw SDFs with every non-minimal object pushed outside the minimal one where the scene SDF
        is negative (used by the mesh extraction, network.py:460-479)."""
        sdf_raw = self.get_sdf_raw(x)
        sdf, idx = self._min_sdf(sdf_raw)
        shifted = torch.where((sdf < 0).expand_as(sdf_raw), torch.max(sdf_raw, (-sdf).expand_as(sdf_raw)), sdf_raw)
        return shifted.scatter(1, idx, sdf)

    def mlp_parameters(self):
        parameters = []
        for lin in self._lins():
            parameters += list(lin.parameters())
        if self.color_grid_feature:
            parameters += list(self.color_grid_feature_map_mlp.parameters())
        return parameters

    def grid_parameters(self, verbose=False):
        if self.color_grid_feature:
            return list(self.encoding.parameters()) + list(self.color_encoding.parameters())
        return self.encoding.parameters()


class RenderingNetwork(nn.Module):
    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_view=0, multires_point=0,
                 multires_normal=0, num_images=1024):
        super().__init__()
        if not weight_norm:
            raise NotImplementedError("Stage-1 configs always use weight_norm=True")
        self.mode = mode
        dims = [d_in + feature_vector_size] + list(dims) + [d_out]
        self.embedview_fn = None
        self.multires_view = multires_view
        self.multires_point = multires_point
        self.multires_normal = multires_normal
        if multires_view > 0 or multires_point > 0 or multires_normal > 0:
            emb = Embedder(multires_view, 3)  # one embedder (of multires_view) serves all three (network.py:559)
            self.embedview_fn = emb.embed
            if multires_view > 0:
                dims[0] += emb.out_dim - 3
            if multires_point > 0 and mode == "idr":
                dims[0] += emb.out_dim - 3
            if multires_normal > 0 and mode == "idr":
                dims[0] += emb.out_dim - 3
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            setattr(self, "lin" + str(l), WNLinear(dims[l], dims[l + 1]))
        self.relu = nn.ReLU()
        self.sigmoid = nn.Sigmoid()
        self.set_mlp_precision(default_mlp_precision())

    def set_mlp_precision(self, precision):
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"mlp precision must be 'fp32' or 'bf16', got {precision!r}")
        self.mlp_bf16 = precision == "bf16"
        for l in range(self.num_layers - 1):
            getattr(self, "lin" + str(l)).bf16 = self.mlp_bf16

    def forward(self, points, normals, view_dirs, feature_vectors, indices=None):
        fused = (self.mode == "idr" and self.multires_view > 0 and self.multires_point > 0 and self.multires_normal > 0)
        if fused:
            x = _render_input.apply(points, view_dirs, normals, feature_vectors, self.multires_view)
        else:
            if self.multires_view > 0:
                view_dirs = self.embedview_fn(view_dirs)
            if self.multires_point > 0:
                points = self.embedview_fn(points)
            if self.multires_normal > 0:
                normals = self.embedview_fn(normals)
            dt = feature_vectors.dtype
            if self.mode == "idr":
                x = torch.cat([points.to(dt), view_dirs.to(dt), normals.to(dt), feature_vectors], dim=-1)
            elif self.mode == "nerf":
                x = torch.cat([view_dirs.to(dt), feature_vectors], dim=-1)
            else:
                raise NotImplementedError
        for l in range(self.num_layers - 1):
            x = getattr(self, "lin" + str(l))(x)
            if l < self.num_layers - 2:
                x = self.relu(x)
        return self.sigmoid(x[:, :3].float())


DEFAULT_FP32_STAGES = ""


class HoloSceneNetwork(nn.Module):
    def __init__(self, conf, plots_dir=None, graph_node_dict=None, ft_folder=None, num_images=1024):
        super().__init__()
        self.feature_vector_size = conf.get_int("feature_vector_size")
        self.scene_bounding_sphere = conf.get_float("scene_bounding_sphere", default=1.0)
        self.white_bkgd = conf.get_bool("white_bkgd", default=False)
        self.register_buffer("bg_color", torch.tensor(conf.get_list("bg_color", default=[1.0, 1.0, 1.0])).float(), persistent=False)
        self.use_bg_reg = conf.get_bool("use_bg_reg", default=False)
        self.render_bg_iter = conf.get_int("render_bg_iter", default=10)
        # Eikonal / smoothness gradients of the 4R-point regulariser set: "analytic" = d sdf_k / dx from the value+Jacobian pass, which is
        # what the reference computes (autograd, network.py:212-254) and the parity path; "fd" = the opt-in 4-tap tetrahedral finite
        # difference BASELINE.json's configs[4] words ("4-tap Eikonal finite-difference"), reported as an extra
        self.eikonal_mode = conf.get_string("eikonal_mode", default="analytic")
        self.eikonal_fd_h = conf.get_float("eikonal_fd_h", default=1.0e-3)
        if self.eikonal_mode not in ("analytic", "fd"):
            raise ValueError(f"eikonal_mode must be 'analytic' or 'fd', got {self.eikonal_mode!r}")
        self.graph_node_dict = graph_node_dict
        self.implicit_network = ObjectImplicitNetworkGrid(self.feature_vector_size, 0.0 if self.white_bkgd else self.scene_bounding_sphere,
                                                          **conf.get_config("implicit_network"))
        self.num_semantic = conf.get_int("implicit_network.d_out")
        self.rendering_network = RenderingNetwork(self.feature_vector_size, num_images=num_images, **conf.get_config("rendering_network"))
        self.density = LaplaceDensity(**conf.get_config("density"))
        self.ray_sampler = ErrorBoundSampler(self.scene_bounding_sphere, **conf.get_config("ray_sampler"))
        precision = conf.get_string("mlp_precision", default=default_mlp_precision())
        self.implicit_network.set_mlp_precision(precision)
        self.rendering_network.set_mlp_precision(precision)
        self.mlp_precision = precision
        # bf16 mode only: stages that nevertheless run in the reference's fp32 arithmetic (library GEMMs): any of "sampler" (the no-grad SDF
        # sweeps), "trunk" (value + gradient of the rendered samples), "eikonal" (value + all K gradients of the 4R regulariser points),
        # "colour" (feature MLP + rendering network).  Conf key `fp32_stages`, env HOLOSCENE_FP32_STAGES (comma separated).  The default is
        # decided by what 300-iteration trainings showed (tools/exp/conv_hybrid.py).
        st = conf.get_list("fp32_stages", default=None)
        if st is None:
            st = [t for t in os.environ.get("HOLOSCENE_FP32_STAGES", DEFAULT_FP32_STAGES).split(",") if t]
        self.fp32_stages = frozenset(st)
        if self.fp32_stages - {"sampler", "trunk", "eikonal", "colour"}:
            raise ValueError(f"fp32_stages: unknown stage in {sorted(self.fp32_stages)}")
        self.plots_dir = plots_dir
        self.ft_folder = ft_folder
        self.all_mesh_bbox_dict = None  # only ever set by the Stage-2 trainer (holoscene_train_post.py:715-731)

    # ---------------------------------------------------------------- compositing (network.py:1803-1824)
    def fused_path_report(self):
        """Which parts of a bf16 training iteration leave the benchmarked kernels, and why: [] = none (the stock path: wave-tile kernels, ONE
        weight-pack launch, whole-iteration graph).  Training warns once per model when the list is not empty (`_warn_off_fused_path`)."""
        net, rn = self.implicit_network, self.rendering_network
        if not net.mlp_bf16:
            return []           # the fp32 configuration is a choice, not a fallback
        out = []
        blockers = net.fused_trunk_blockers()
        if blockers:
            out.append("SDF trunk on library GEMMs + elementwise launches, sampler rounds host-controlled, no whole-iteration graph: " + "; ".join(blockers))
        else:
            K = net._lins()[2].out_features
            if K != net.d_out:
                out.append(f"trunk output width {K} != d_out {net.d_out}: rendered samples on the library-GEMM value+Jacobian path")
            elif K > 32:
                if ops.RR_WIDE and ops.TRUNK_MODE == "rr" and ops.RR_FORWARD == "fused":
                    out.append(f"d_out = {K} > 32: the benchmarked kernels with the last layer as two 32-row tiles (k_sdf_mlp2<true>, k_rr_fwd<true>, "
                               "k_rr_bwd_value<., true>, two 32-row weight-gradient jobs); three weight-pack launches per parameter state instead of the "
                               "iteration's one, the Eikonal points' backward on the 128-point workgroup-tile kernel (k_trunk_bwd<64>)")
                else:
                    out.append(f"d_out = {K} > 32 with HOLOSCENE_RR_WIDE=0 / ops.TRUNK_MODE != rr: rendered samples on the four-row value+Jacobian kernels "
                               "(k_trunk_fwd2<true, true>, k_trunk_bwd<64>, library weight-gradient GEMMs); per-call weight packing")
        probe = self.density.beta
        if probe.is_cuda and not self._fused_appearance_supported(probe):
            out.append("colour branch on library GEMMs: it is not the stock one (idr mode, four layers of 256, 4 frequencies each, 16 x 2 colour grid, "
                       "256-wide feature MLP)")
        return out

    def _warn_off_fused_path(self):
        if getattr(self, "_off_fused_warned", False):
            return
        self._off_fused_warned = True
        report = self.fused_path_report()
        if report:
            import warnings
            warnings.warn("holoscene_amd: this conf does not run on the benchmarked bf16 kernels end to end -- " + " | ".join(report)
                          + " (DESIGN.md section 5; expect a slower iteration, results unchanged)")

    def _fused_appearance_supported(self, x):
        net, rn = self.implicit_network, self.rendering_network
        if not (x.is_cuda and net.mlp_bf16 and rn.mlp_bf16 and net.color_grid_feature and rn.mode == "idr" and rn.num_layers == 4):
            return False
        mlp = net.color_grid_feature_map_mlp
        enc = net.color_encoding
        padded = getattr(mlp[0], "fused_cols", None) == 32 and enc.fused_pads       # fewer levels: empty ones behind them, zero columns in mlp[0]
        return (rn.multires_view == 4 and rn.multires_point == 4 and rn.multires_normal == 4 and enc.level_dim == 2
                and ((enc.num_levels == 16 and tuple(mlp[0].weight.shape) == (256, 32)) or (padded and tuple(mlp[0].weight.shape) == (256, 2 * enc.num_levels)))
                and tuple(mlp[2].weight.shape) == (256, 256)
                and tuple(rn.lin0.weight_v.shape) == (256, 337) and tuple(rn.lin1.weight_v.shape) == (256, 256)
                and tuple(rn.lin2.weight_v.shape) == (3, 256))

    def volume_rendering(self, z_vals, sdf):
        density = self.density(sdf).reshape(-1, z_vals.shape[1])
        dists = z_vals[:, 1:] - z_vals[:, :-1]
        dists = torch.cat([dists, torch.full_like(dists[:, :1], 1e10)], -1)
        free_energy = dists * density
        shifted = torch.cat([torch.zeros_like(free_energy[:, :1]), free_energy[:, :-1]], dim=-1)
        alpha = 1 - torch.exp(-free_energy)
        transmittance = torch.exp(-torch.cumsum(shifted, dim=-1))
        return alpha * transmittance, transmittance, dists

    def occlusion_opacity(self, z_vals, transmittance, dists, sdf_raw):
        obj_density = self.density(sdf_raw).transpose(0, 1).reshape(-1, dists.shape[0], dists.shape[1])  # [K, R, N]
        return (1 - torch.exp(-dists * obj_density)) * transmittance

    def _rgb_at(self, points_flat, dirs_flat, gradients, indices=None, x01=None):
        """Colour of every sample [B,3]: colour hash grid -> feature MLP -> rendering network, through the fused matrix-core kernels
        (csrc/appearance_mlp.hip) when the shapes are the stock ones in bf16 mode, else library GEMMs."""
        net = self.implicit_network
        with self._stage_fp32("colour") as forced:
            if forced:
                return self.rendering_network(points_flat, gradients, dirs_flat, net._color_features(points_flat), indices)
        if ops.APPEARANCE_IMPL == "mfma" and self._fused_appearance_supported(points_flat):
            enc, mlp, rn = net.color_encoding, net.color_grid_feature_map_mlp, self.rendering_network
            R0, R1, R2 = effective_weights([rn.lin0, rn.lin1, rn.lin2])
            return fused_appearance(points_flat, dirs_flat, gradients, enc.embeddings, enc.fused_offsets, float(np.log2(enc.per_level_scale)),
                                           int(enc.base_resolution), float(net.divide_factor), self._color_w0(), mlp[0].bias, mlp[2].weight,
                                           mlp[2].bias, R0, rn.lin0.bias, R1, rn.lin1.bias, R2, rn.lin2.bias, x01)
        return self.rendering_network(points_flat, gradients, dirs_flat, net._color_features(points_flat), indices)

    # ---------------------------------------------------------------- Stage-2/3 entry points
    # network.py:1016-1801 is sixteen near-copies of one routine: sample the rays against an object subset, evaluate value +
    # gradient + colour at the samples, composite with the weights of one SDF (`weights`) and of another (`bg_weights`).
    # _render_object_rays is that routine; the public methods below keep the reference's names, signatures and return values
    # (plus an optional `rng=` dict of injected draws, as everywhere in this package).
    def _render_object_rays(self, cam_loc, ray_dirs, rot, depth_scale, kind, obj_idxs, subset=None, near_far=None, sem_bg_weights=False,
                            detach_rgb=False, nf_outputs=False, indices=0, rng=None):
        """kind: "multi" = scene outputs + min over obj_idxs (get_multi_specific_outputs), "only" = everything restricted to
        obj_idxs, "subset" = outputs over `subset` + min over obj_idxs."""
        sm, net = self.ray_sampler, self.implicit_network
        if near_far is None:
            z_vals, _ = sm.get_z_vals(ray_dirs, cam_loc, self, idx=obj_idxs, rng=rng)
        else:
            z_vals, _ = sm.get_z_vals_near_far(ray_dirs, cam_loc, self, near_far[0], near_far[1], idx=obj_idxs, rng=rng)
        N = z_vals.shape[1]
        points_flat = (cam_loc.unsqueeze(1) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
        dirs_flat = ray_dirs.unsqueeze(1).expand(-1, N, -1).reshape(-1, 3)
        # one value+Jacobian pass; the variants of network.py:359-435 only choose the columns each minimum runs over (the net-level
        # get_*_outputs methods do the same and also return the colour features; here the colour goes through _rgb_at instead)
        sdf_raw, J, fv = net._value_jacobian_features(points_flat, want_features=not net.color_grid_feature)
        cols = {"multi": None, "only": obj_idxs, "subset": subset}[kind]
        if kind not in ("multi", "only", "subset"):
            raise ValueError(kind)
        sdf, gradients = net._subset_min(sdf_raw, J, cols)
        semantic = net.sigmoid * torch.sigmoid(-net.sigmoid * (sdf_raw if cols is None else sdf_raw[:, cols]))
        sdf_obj = None if kind == "only" else sdf_raw[:, obj_idxs].min(dim=-1, keepdim=True)[0]
        g_rgb = gradients.detach() if detach_rgb else gradients
        if fv is None:
            rgb = self._rgb_at(points_flat, dirs_flat, g_rgb, indices).reshape(-1, N, 3)
        else:
            rgb = self.rendering_network(points_flat, g_rgb, dirs_flat, fv, indices).reshape(-1, N, 3)
        base = {"rgb": rgb, "z_vals": z_vals, "depth_vals": z_vals * depth_scale, "sdf": sdf.reshape(z_vals.shape)}
        if ops.COMPOSITE_IMPL == "hip" and z_vals.is_cuda and not nf_outputs and not detach_rgb:
            # the fused compositing kernel (csrc/composite.hip) once per weight set.  Pass A runs on the SDF the semantics belong to,
            # with one extra "object" column appended to the per-object SDFs so that its occlusion-aware opacity comes out of the
            # same launch; pass B runs on the object-subset minimum and delivers colour, depth and the (rotated) normal map.
            beta, ds, rgb_flat = self.density.get_beta(), depth_scale.contiguous(), rgb.reshape(-1, 3)
            cols_raw = sdf_raw if cols is None else sdf_raw[:, cols]
            n_sem = cols_raw.shape[1]
            extra = sdf if kind in ("only", "subset") else sdf_obj          # whose density the opacity uses (network.py:1131, 1201, 1270)
            a = _composite.apply(z_vals, sdf, torch.cat([cols_raw, extra], 1), rgb_flat, gradients, beta, ds, net.sigmoid, rot)
            weights, sem_w, opacity = a[0], a[5][:, :n_sem], a[6][:, n_sem:]
            if kind == "only":
                base.update({"semantic_values": sem_w, "object_opacity": opacity, "rgb_values": a[2], "depth_values": a[3], "weights": weights,
                             "normal_map": a[4]})
                return base
            b = _composite.apply(z_vals, sdf_obj, cols_raw if sem_bg_weights else sdf_obj, rgb_flat, gradients, beta, ds, net.sigmoid, rot)
            base.update({"semantic_values": b[5] if sem_bg_weights else sem_w, ("object_opacity" if kind == "multi" else "opacity"): opacity,
                         "rgb_values": b[2], "depth_values": b[3], "weights": weights, "bg_weights": b[0], "normal_map": b[4]})
            return base
        semantic = semantic.reshape(-1, N, semantic.shape[-1])
        weights, transmittance, dists = self.volume_rendering(z_vals, sdf)
        if kind == "only":
            w_out = weights
            opacity_key, opacity = "object_opacity", self.occlusion_opacity(z_vals, transmittance, dists, sdf).sum(-1).transpose(0, 1)
        else:
            w_out, _, _ = self.volume_rendering(z_vals, sdf_obj)
            if kind == "multi":
                opacity_key, opacity = "object_opacity", self.occlusion_opacity(z_vals, transmittance, dists, sdf_obj).sum(-1).transpose(0, 1)
            elif nf_outputs:
                opacity_key, opacity = "opacity", torch.sum(w_out, dim=-1).reshape(-1)
            else:
                opacity_key, opacity = "opacity", self.occlusion_opacity(z_vals, transmittance, dists, sdf).sum(-1).transpose(0, 1)
        rgb_values = torch.sum((w_out.detach() if detach_rgb else w_out).unsqueeze(-1) * rgb, 1)
        semantic_values = torch.sum((w_out if sem_bg_weights else weights).unsqueeze(-1) * semantic, 1)
        depth_values = torch.sum(w_out * z_vals, 1, keepdims=True)
        if not nf_outputs:
            depth_values = depth_values / (w_out.sum(dim=1, keepdims=True) + 1e-8)
        output = dict(base)
        output.update({"semantic_values": semantic_values, opacity_key: opacity, "rgb_values": rgb_values,
                       "depth_values": depth_scale * depth_values, "weights": weights})
        if kind != "only":
            output["bg_weights"] = w_out
        normals = (gradients / (gradients.norm(2, -1, keepdim=True) + 1e-6)).reshape(-1, N, 3)
        normal_map = torch.sum(w_out.unsqueeze(-1) * normals, 1)
        output["normal_map"] = (rot @ normal_map.permute(1, 0)).permute(1, 0).contiguous()
        return output

    @staticmethod
    def _world_rays(ray_origins, ray_dirs, pose):
        """Rays given in world space (network.py:1095-1104): normalised directions, world->camera rotation, z of the camera-frame
        direction as the depth scale."""
        cam_loc = ray_origins.reshape(-1, 3)
        ray_dirs = F.normalize(ray_dirs.reshape(-1, 3), dim=-1)
        rot = pose[..., :3, :3].reshape(3, 3).permute(1, 0).contiguous()
        depth_scale = (rot @ ray_dirs.permute(1, 0)).permute(1, 0)[:, 2:]
        return cam_loc, ray_dirs, rot, depth_scale

    def forward_multi_obj(self, input, indices, obj_idxs, iter_step=-1, rng=None):
        """network.py:1016-1090: pixel rays of one frame rendered against the objects `obj_idxs`."""
        intrinsics, uv, pose = input["intrinsics"], input["uv"], input["pose"]
        ray_dirs, cam_loc = rend_util.get_camera_params(uv, pose, intrinsics)
        ray_dirs_tmp, _ = rend_util.get_camera_params(uv, torch.eye(4, device=pose.device)[None], intrinsics)
        num_pixels = ray_dirs.shape[1]
        cam_loc = cam_loc.unsqueeze(1).repeat(1, num_pixels, 1).reshape(-1, 3)
        rot = pose[0, :3, :3].permute(1, 0).contiguous()
        return self._render_object_rays(cam_loc, ray_dirs.reshape(-1, 3), rot, ray_dirs_tmp[0, :, 2:], "multi", obj_idxs, indices=indices, rng=rng)

    def forward_multi_obj_rays(self, ray_origins, ray_dirs, pose, obj_idxs, iter_step=-1, sem_bg_weights=False, rng=None):
        """network.py:1092-1164."""
        return self._render_object_rays(*self._world_rays(ray_origins, ray_dirs, pose), "multi", obj_idxs, sem_bg_weights=sem_bg_weights, rng=rng)

    def forward_only_multi_obj_rays(self, ray_origins, ray_dirs, pose, obj_idxs, iter_step=-1, rng=None):
        """network.py:1166-1233."""
        return self._render_object_rays(*self._world_rays(ray_origins, ray_dirs, pose), "only", obj_idxs, rng=rng)

    def forward_multi_obj_rays_subset_all_sdf(self, ray_origins, ray_dirs, pose, obj_idxs, subset_obj_idxs, iter_step=-1, rng=None):
        """network.py:1235-1305."""
        return self._render_object_rays(*self._world_rays(ray_origins, ray_dirs, pose), "subset", obj_idxs, subset=subset_obj_idxs, rng=rng)

    def forward_multi_obj_rays_subset_all_sdf_near_far(self, ray_origins, ray_dirs, pose, obj_idxs, subset_obj_idxs, near, far, iter_step=-1,
                                                       rng=None):
        """network.py:1307-1382 (un-normalised depth sum; 'opacity' = sum of the object weights)."""
        return self._render_object_rays(*self._world_rays(ray_origins, ray_dirs, pose), "subset", obj_idxs, subset=subset_obj_idxs,
                                        near_far=(near, far), nf_outputs=True, rng=rng)

    def forward_multi_obj_rays_subset_all_sdf_detach_rgb_for_geometry(self, ray_origins, ray_dirs, pose, obj_idxs, subset_obj_idxs, iter_step=-1,
                                                                      rng=None):
        """network.py:1384-1456: the colour term sees detached normals and detached object weights."""
        return self._render_object_rays(*self._world_rays(ray_origins, ray_dirs, pose), "subset", obj_idxs, subset=subset_obj_idxs,
                                        detach_rgb=True, rng=rng)

    def forward_multi_obj_rays_subset_all_sdf_detach_rgb_for_geometry_near_far(self, ray_origins, ray_dirs, pose, obj_idxs, subset_obj_idxs, near,
                                                                               far, iter_step=-1, rng=None):
        """network.py:1458-1530."""
        return self._render_object_rays(*self._world_rays(ray_origins, ray_dirs, pose), "subset", obj_idxs, subset=subset_obj_idxs,
                                        near_far=(near, far), detach_rgb=True, rng=rng)

    def _colors_along_rays(self, points, rays, obj=None, near_far=None, nm=False, rng=None):
        """Shared body of the get_colors_* family (network.py:1532-1800): rays leaving `points` along `rays`, composited with the
        weights of the scene SDF (obj None), of one object's own SDF (nm: get_specific_outputs_nm) or of the min over [obj].
        Returns weights [R,N], colour [R,3], world-frame normal map [R,3], composited semantics [R,K']."""
        cam_loc = points.reshape(-1, 3)
        ray_dirs = F.normalize(rays.reshape(-1, 3), dim=-1)
        sm, net = self.ray_sampler, self.implicit_network
        idx = None if obj is None else (obj if nm else [obj])
        if near_far is None:
            z_vals, _ = sm.get_z_vals(ray_dirs, cam_loc, self, idx=idx, rng=rng)
        else:
            z_vals, _ = sm.get_z_vals_near_far(ray_dirs, cam_loc, self, near_far[0], near_far[1], idx=idx, rng=rng)
        N = z_vals.shape[1]
        points_flat = (cam_loc.unsqueeze(1) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
        dirs_flat = ray_dirs.unsqueeze(1).expand(-1, N, -1).reshape(-1, 3)
        sdf_raw, J, fv = net._value_jacobian_features(points_flat, want_features=not net.color_grid_feature)
        if nm:        # one object's own SDF and gradient, no minimum (get_specific_outputs_nm)
            sdf, gradients = sdf_raw[:, obj], J[:, obj]
        else:         # scene minimum (get_outputs) or the minimum over [obj] (get_multi_specific_outputs_subset_objs)
            sdf, gradients = net._subset_min(sdf_raw, J, None if obj is None else [obj])
        semantic = net.sigmoid * torch.sigmoid(-net.sigmoid * (sdf_raw if (obj is None or nm) else sdf_raw[:, [obj]]))
        if fv is None:
            rgb = self._rgb_at(points_flat, dirs_flat, gradients, 0).reshape(-1, N, 3)
        else:
            rgb = self.rendering_network(points_flat, gradients, dirs_flat, fv, 0).reshape(-1, N, 3)
        if ops.COMPOSITE_IMPL == "hip" and z_vals.is_cuda:      # one launch: weights, colour, world-frame normal map, composited semantics
            ones = torch.ones(z_vals.shape[0], 1, device=z_vals.device)
            raw_cols = sdf_raw if (obj is None or nm) else sdf_raw[:, [obj]]
            a = _composite.apply(z_vals, sdf.reshape(-1, 1), raw_cols, rgb.reshape(-1, 3), gradients, self.density.get_beta(), ones, net.sigmoid)
            return a[0], a[2], a[4], a[5]
        weights, _, _ = self.volume_rendering(z_vals, sdf)
        normals = (gradients / (gradients.norm(2, -1, keepdim=True) + 1e-6)).reshape(-1, N, 3)
        sem = torch.sum(weights.unsqueeze(-1) * semantic.reshape(-1, N, semantic.shape[-1]), 1)
        return weights, torch.sum(weights.unsqueeze(-1) * rgb, 1).reshape(-1, 3), torch.sum(weights.unsqueeze(-1) * normals, 1), sem

    @staticmethod
    def _to_camera(normal_world, pose):
        rot = pose.reshape(-1, 4)[:3, :3].permute(1, 0).contiguous()
        return (rot @ normal_world.permute(1, 0)).permute(1, 0).contiguous()

    def get_colors_normals_from_point_rays(self, points, rays, pose, rng=None):
        """network.py:1532-1569."""
        _, rgb_values, normal_world, _ = self._colors_along_rays(points, rays, rng=rng)
        return rgb_values, self._to_camera(normal_world, pose)

    def get_colors_normals_from_point_rays_obj(self, points, rays, pose, obj_idx, rng=None):
        """network.py:1571-1612: one object's own SDF; also the arg-max label of the composited semantics."""
        _, rgb_values, normal_world, sem = self._colors_along_rays(points, rays, obj=obj_idx, nm=True, rng=rng)
        return rgb_values, self._to_camera(normal_world, pose), torch.argmax(sem, dim=-1)

    def get_colors_normals_from_point_rays_obj_f(self, points, rays, pose, obj_idx, rng=None):
        """network.py:1614-1654: as above, the composited semantics themselves."""
        _, rgb_values, normal_world, sem = self._colors_along_rays(points, rays, obj=obj_idx, nm=True, rng=rng)
        return rgb_values, self._to_camera(normal_world, pose), sem

    def get_colors_from_point_rays(self, points, rays, rng=None):
        """network.py:1656-1683."""
        return self._colors_along_rays(points, rays, rng=rng)[1]

    def get_colors_from_point_rays_obj(self, points, rays, obj_i, rng=None):
        """network.py:1685-1712."""
        return self._colors_along_rays(points, rays, obj=obj_i, rng=rng)[1]

    def get_colors_from_point_rays_obj_offset(self, points, rays, obj_i, rng=None):
        """network.py:1714-1741 (identical to get_colors_from_point_rays_obj in the reference)."""
        return self._colors_along_rays(points, rays, obj=obj_i, rng=rng)[1]

    def get_colors_from_point_rays_obj_offset_near_far(self, points, rays, obj_i, near, far, rng=None):
        """network.py:1743-1770."""
        return self._colors_along_rays(points, rays, obj=obj_i, near_far=(near, far), rng=rng)[1]

    def get_colors_from_point_rays_obj_debug(self, points, rays, obj_i, rng=None):
        """network.py:1772-1801: colour and the summed weights per ray."""
        weights, rgb_values, _, _ = self._colors_along_rays(points, rays, obj=obj_i, rng=rng)
        return rgb_values, torch.sum(weights, 1)

    # ---------------------------------------------------------------- forward (network.py:778-971), in stages
    # forward() = prepare_rays -> sample -> (prepare_background) -> render.  The stages exist so the trainer can
    # run the data-dependent part (rays + Algorithm-1 sampler, which needs a host decision per round) eagerly and
    # replay everything after it -- render, loss, backward, Adam -- as one captured HIP graph.
    _FD_TAPS = ((1.0, -1.0, -1.0), (-1.0, -1.0, 1.0), (-1.0, 1.0, -1.0), (1.0, 1.0, 1.0))
    _fd_taps_dev = {}

    def _eikonal_gradients_fd(self, x, y_centre):
        """Opt-in replacement of the analytic Eikonal-set gradients: grad f(x) ~ sum_i k_i f(x + h k_i) / (4 h) over the four tetrahedral
        taps k_i (truncation error of order h^2 times the third derivatives), for every object SDF and -- with the centre point's arg-min
        object -- for the scene SDF; same stacking as gradient() ([(K + 1) * B, 3], network.py:212-254).  The taps go through the
        differentiable value-only trunk, so the parameters receive the finite-difference form's own gradient.  4 x the Eikonal set's
        points (16 R) instead of its 3 tangent rows."""
        net, h = self.implicit_network, self.eikonal_fd_h
        if net.mlp_bf16:
            # measured: relative L2 distance 1.4 from the analytic rows -- a bf16 trunk rounds its INPUTS to 8 bits (spacing 2e-3 .. 4e-3 at
            # |x| ~ 0.5), so taps 1e-3 apart collapse onto the same operand, and its output noise (3e-3 relative) exceeds the differences
            raise ValueError("eikonal_mode='fd' needs mlp_precision='fp32': finite differences at h = 1e-3 are below the resolution of bf16 operands")
        B = x.shape[0]
        key = (str(x.device), x.dtype)          # built once per device: a host-to-device copy is not allowed inside a graph capture
        if key not in self._fd_taps_dev:       # (the trainer's eager warm-up passes fill the cache before it captures)
            self._fd_taps_dev[key] = torch.tensor(self._FD_TAPS, device=x.device, dtype=x.dtype)
        taps = self._fd_taps_dev[key]                                                       # [4,3]
        pts = (x.detach().unsqueeze(0) + h * taps.unsqueeze(1)).reshape(-1, 3)              # [4B,3]
        y = net._trunk(pts)[:, :net.d_out].float().reshape(4, B, net.d_out)                 # [4,B,K]
        J = torch.einsum("tbk,td->bkd", y, taps) / (4.0 * h)                                # [B,K,3]
        idx = y_centre.detach().argmin(dim=-1, keepdim=True)
        g_min = torch.gather(J, 1, idx.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
        return torch.cat([J.transpose(0, 1).reshape(-1, 3), g_min], 0)

    def weight_norm_layers(self):
        """Every weight-normalised layer a Stage-1 iteration evaluates (for shared_effective_weights)."""
        rn = self.rendering_network
        return list(self.implicit_network._lins()) + [l for l in (getattr(rn, "lin0", None), getattr(rn, "lin1", None), getattr(rn, "lin2", None))
                                                       if l is not None]

    def _color_w0(self):
        """The colour feature MLP's first matrix as the fused colour kernels read it: 32 feature columns (zero ones behind those of a grid with
        fewer than 16 levels).  ONE tensor per iteration_prologue() block -- the packed images are keyed by its identity."""
        lin = self.implicit_network.color_grid_feature_map_mlp[0]
        if getattr(lin, "fused_cols", None) is None or not lin.weight.is_cuda:
            return lin.weight
        packs = ops._SCOPE.packs
        if packs is not None and packs.get("net") == id(self.implicit_network) and packs.get("color_w0") is not None:
            return packs["color_w0"]
        return fused_cols(lin.weight, lin)

    def _pack_iteration(self):
        """Inside iteration_prologue(): the fragment images of this iteration's weights for the sampler sweeps, the training trunk and the
        colour branch from one launch -- None when the model does not run on the fused bf16 kernels (the pack sites then pack for themselves)."""
        net, rn = self.implicit_network, self.rendering_network
        probe = self.density.beta
        if not (probe.is_cuda and ops.TRUNK_IMPL == "mfma" and ops.TRUNK_MODE == "rr" and ops.RR_FORWARD == "fused" and ops.SDF_MLP_IMPL == "wave" and ops.APPEARANCE_IMPL == "mfma"
                and ops.APPEARANCE_FORM == "wave" and net._fused_trunk_supported(probe) and self._fused_appearance_supported(probe)):
            return None
        l0, l1, l2 = net._lins()
        if l2.out_features != net.d_out or net.d_out > 32:
            return None
        grad = torch.is_grad_enabled()          # (the colour branch's backward image only when a backward pass can follow)
        mlp = net.color_grid_feature_map_mlp
        c0 = fused_cols(mlp[0].weight, mlp[0])  # (differentiable: made outside the no_grad block; == mlp[0].weight on the stock grid)
        with torch.no_grad():
            W0, W1, W2 = effective_weights([l0, l1, l2])
            R0, R1, R2 = effective_weights([rn.lin0, rn.lin1, rn.lin2])
            f = lambda t: t.detach().float().contiguous()  # noqa: E731
            out = _be._backend.pack_iteration(
                (f(W0), f(l0.bias), f(W1), f(l1.bias), f(W2), f(l2.bias), net.d_out, True, True, self.training),
                ((c0, mlp[2].weight, R0, R1, R2), (mlp[0].bias, mlp[2].bias, rn.lin0.bias, rn.lin1.bias, rn.lin2.bias), grad))
        out["net"] = id(net)
        out["trunk_key"] = (id(W0), id(W1), id(W2))
        out["appear_key"] = (id(c0), id(mlp[2].weight), id(R0), id(R1), id(R2))
        out["color_w0"] = c0 if c0 is not mlp[0].weight else None
        return out

    BG_PATCH = 32       # side of the background patch (network.py:919-925)

    def uniform_sizes(self, num_rays, with_bg=False):
        """Names and shapes of the U[0, 1) draws of one training iteration (draw_uniforms / iteration_prologue).  with_bg: also the background
        patch's -- its origin and its own sampler's draws ("bg.<name>": nested under "bg" in the rng dictionary, nest_draws)."""
        sm = self.ray_sampler
        R = num_rays
        sizes = {"ray_offset_u": (1, R, 2), "t_rand": (R, sm.N_samples_eval), "u_final": (R, sm.N_samples), "u_pick": (max(sm.N_samples_extra, 1),),
                 "eik_u": (R,), "eik_uniform_u": (R, 3), "eik_jitter": (2 * R, 3)}
        if with_bg:
            P = self.BG_PATCH ** 2
            sizes.update({"bg_xy_u": (2,), "bg.t_rand": (P, sm.N_samples_eval), "bg.u_final": (P, sm.N_samples), "bg.u_pick": (max(sm.N_samples_extra, 1),),
                          "bg.eik_u": (P,)})
        return sizes

    @staticmethod
    def nest_draws(rng):
        """{"bg.t_rand": t, ...} -> {"bg": {"t_rand": t, ...}, ...} (the second sampler dictionary of forward()'s rng)."""
        out = {}
        for k, v in rng.items():
            if k.startswith("bg."):
                out.setdefault("bg", {})[k[3:]] = v
            else:
                out[k] = v
        return out

    def rng_state(self, device):
        """Device-resident (seed, counter, scratch) of this model's Philox stream (hs_iter_prologue): seeded once from torch's default CPU
        generator -- torch.manual_seed decides it, data-parallel ranks that reseed after the common initialisation get their own --, advanced
        by every launch that draws from it."""
        st = getattr(self, "_rng_state", None)
        dev = torch.device(device)
        if st is None or st.device.type != dev.type or (dev.index is not None and st.device.index != dev.index):
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            st = self._rng_state = torch.tensor([seed, 0, 0], dtype=torch.int64).to(device)
        return st

    def draw_uniforms(self, num_rays, device, with_bg=False):
        """Every U[0,1) draw of one training iteration from ONE generator launch (the reference draws them where it needs them:
        network.py:773 ray offsets, ray_sampler.py:77 stratified jitter, :238 inverse-CDF draws, :269 extra samples, :279
        Eikonal pick, network.py:846-853 Eikonal points): an `rng` dict for prepare_rays / sample / render whose *_u entries are
        raw draws that the consuming kernels shift / scale / quantise themselves."""
        sizes = self.uniform_sizes(num_rays, with_bg)
        total = sum(int(np.prod(v)) for v in sizes.values())
        pool = torch.rand(total, device=device)
        rng, off = {}, 0
        for k, shp in sizes.items():
            n = int(np.prod(shp))
            rng[k] = pool[off:off + n].view(shp)
            off += n
        return self.nest_draws(rng)

    def prepare_rays(self, input, rng=None):
        rng = rng or {}
        intrinsics, uv, pose = input["intrinsics"], input["uv"], input["pose"]
        dev = uv.device
        from . import ray_sampler as _rs
        fused = _rs.SAMPLER_IMPL == "hip" and uv.is_cuda and pose.shape[1] != 7 and uv.shape[0] == 1
        shift = 0.0
        if not self.training:
            ray_offset = None
        elif "ray_offset" in rng:
            ray_offset = rng["ray_offset"]
        elif "ray_offset_u" in rng and fused:
            ray_offset, shift = rng["ray_offset_u"], -0.5     # the kernel subtracts the 0.5
        else:
            ray_offset = (rng["ray_offset_u"] if "ray_offset_u" in rng else torch.rand_like(uv)) - 0.5
        if fused:
            return self._setup_rays_fused(uv, ray_offset, pose, intrinsics, rng.get("t_rand"), offset_shift=shift)
        if self.training:
            ray_dirs, cam_loc = rend_util.get_camera_params(uv, pose, intrinsics, ray_offset=ray_offset)
            # quirk Q1: the reference's first call shifted uv in place, so its depth-scale rays carry 2x the offset
            ray_dirs_tmp, _ = rend_util.get_camera_params(uv, torch.eye(4, device=dev)[None], intrinsics, ray_offset=2 * ray_offset)
        else:
            ray_dirs, cam_loc = rend_util.get_camera_params(uv, pose, intrinsics)
            ray_dirs_tmp, _ = rend_util.get_camera_params(uv, torch.eye(4, device=dev)[None], intrinsics)
        num_pixels = ray_dirs.shape[1]
        return {"ray_dirs": ray_dirs.reshape(-1, 3).contiguous(),
                "cam_loc": cam_loc.unsqueeze(1).repeat(1, num_pixels, 1).reshape(-1, 3).contiguous(),
                "depth_scale": ray_dirs_tmp[0, :, 2:].contiguous(),
                "rot": pose[0, :3, :3].permute(1, 0).contiguous()}

    def _setup_rays_fused(self, uv, ray_offset, pose, intrinsics, t_rand=None, offset_shift=0.0, patch_u=None):
        """Rays, depth scale, the sampler's first (stratified uniform) depths and its Lemma-2 beta from one kernel
        (csrc/sampler.hip: k_ray_setup).  patch_u (uv = None): the rays of the BG_PATCH x BG_PATCH pixel block that two U[0, 1) draws place."""
        dev = pose.device
        R = uv.shape[1] if uv is not None else self.BG_PATCH ** 2
        sm = self.ray_sampler
        S = sm.N_samples_eval
        if self.training and t_rand is None:
            t_rand = torch.rand(R, S, device=dev)
        out = {"ray_dirs": torch.empty(R, 3, device=dev), "cam_loc": torch.empty(R, 3, device=dev), "depth_scale": torch.empty(R, 1, device=dev),
               "z0": torch.empty(R, S, device=dev), "beta_init": torch.empty(R, device=dev),
               "beta_work": torch.empty(R, device=dev),      # a second copy for the sampler to iterate on (taken by the first sample() call)
               "x0": torch.empty(R * S, 3, device=dev), "x0_grid": torch.empty(R * S, 3, device=dev),   # positions of z0: the sampler's first sweep
               "rot": torch.empty(3, 3, device=dev)}          # pose[0, :3, :3]^T, written by the kernel
        _be._backend.ray_setup(None if uv is None else uv[0].contiguous().float(), None if ray_offset is None else ray_offset[0].contiguous().float(),
                               pose[0].contiguous().float(), intrinsics[0].contiguous().float(),
                               None if t_rand is None else t_rand.to(dev).contiguous(), S, float(sm.uniform_sampler.near),
                               float(sm.uniform_sampler.far), float(self.scene_bounding_sphere), float(sm.eps), out["ray_dirs"], out["cam_loc"],
                               out["depth_scale"], out["z0"], out["beta_init"], float(self.implicit_network.divide_factor), out["x0"], out["x0_grid"], offset_shift, out["rot"],
                               beta_work=out["beta_work"], patch_u=None if patch_u is None else patch_u.contiguous().float(), patch=self.BG_PATCH)
        return out

    @contextlib.contextmanager
    def _stage_fp32(self, stage):
        """Run the enclosed stage of a bf16 model in fp32 when `stage` is listed in fp32_stages (autograd Functions record the precision
        they ran in, so the backward pass follows without the switch)."""
        net, rn = self.implicit_network, self.rendering_network
        if not (stage in self.fp32_stages and net.mlp_bf16):
            yield False
            return
        net.set_mlp_precision("fp32")
        rn.set_mlp_precision("fp32")
        try:
            yield True
        finally:
            net.set_mlp_precision("bf16")
            rn.set_mlp_precision("bf16")

    def sample(self, rays, rng=None, idx=None):
        with self._stage_fp32("sampler"):
            return self._sample(rays, rng, idx)

    def _sample(self, rays, rng=None, idx=None):
        return self.ray_sampler.get_z_vals(rays["ray_dirs"], rays["cam_loc"], self, idx=idx, rng=rng, z0=rays.get("z0"),
                                           beta_init=rays.get("beta_init"), x0=(rays["x0"], rays["x0_grid"]) if "x0" in rays else None,
                                           beta_work=rays.pop("beta_work", None))

    def wants_background(self, iter_step):
        return bool(self.use_bg_reg and iter_step % self.render_bg_iter == 0)

    def prepare_background(self, input, rng=None):
        """Rays and depths of the 32x32 background patch (network.py:919-942)."""
        rng = rng or {}
        intrinsics, pose = input["intrinsics"], input["pose"]
        dev = pose.device
        patch = self.BG_PATCH
        from . import ray_sampler as _rs
        if "bg_xy_u" in rng and "bg_xy0" not in rng and _rs.SAMPLER_IMPL == "hip" and pose.is_cuda and pose.shape[1] != 7:
            # raw draws from the iteration's pool: the ray kernel places the patch itself (no origin / pixel-grid tensors: ~13 launches less)
            bg = self._setup_rays_fused(None, None, pose, intrinsics, (rng.get("bg") or {}).get("t_rand"), patch_u=rng["bg_xy_u"])
            x0 = (bg.pop("x0"), bg.pop("x0_grid"))
            bg["z_vals"], _ = self.ray_sampler.get_z_vals(bg["ray_dirs"], bg["cam_loc"], self, idx=0, rng=rng.get("bg"), z0=bg.pop("z0", None),
                                                          beta_init=bg.pop("beta_init", None), x0=x0)
            bg.pop("rot", None)
            return bg
        if "bg_xy0" in rng and torch.is_tensor(rng["bg_xy0"]) and rng["bg_xy0"].device == dev:
            xy0 = rng["bg_xy0"].float()       # already on the device (graph capture: no host->device copy here)
        elif "bg_xy0" in rng:
            xy0 = torch.tensor([int(v) for v in rng["bg_xy0"]], device=dev, dtype=torch.float32)
        else:  # device-side draw of the patch origin: no host read of the intrinsics
            span = (intrinsics[0, :2, 2] * 2.0).floor() - patch + 1
            xy0 = torch.floor(torch.rand(2, device=dev) * span)
        gy, gx = torch.meshgrid(torch.arange(patch, device=dev), torch.arange(patch, device=dev), indexing="ij")
        uv0 = (torch.stack([gx, gy], -1).reshape(1, -1, 2).float() + xy0)
        if _rs.SAMPLER_IMPL == "hip" and uv0.is_cuda and pose.shape[1] != 7:
            bg = self._setup_rays_fused(uv0, None, pose, intrinsics, (rng.get("bg") or {}).get("t_rand"))
        else:
            ray_dirs0, cam_loc0 = rend_util.get_camera_params(uv0, pose, intrinsics)
            tmp0, _ = rend_util.get_camera_params(uv0, torch.eye(4, device=dev)[None], intrinsics)
            n0 = ray_dirs0.shape[1]
            bg = {"ray_dirs": ray_dirs0.reshape(-1, 3).contiguous(),
                  "cam_loc": cam_loc0.unsqueeze(1).repeat(1, n0, 1).reshape(-1, 3).contiguous(),
                  "depth_scale": tmp0[0, :, 2:].contiguous()}
        x0 = (bg.pop("x0"), bg.pop("x0_grid")) if "x0" in bg else None
        bg["z_vals"], _ = self.ray_sampler.get_z_vals(bg["ray_dirs"], bg["cam_loc"], self, idx=0, rng=rng.get("bg"), z0=bg.pop("z0", None),
                                                      beta_init=bg.pop("beta_init", None), x0=x0)
        bg.pop("rot", None)
        return bg

    def forward(self, input, indices, iter_step=-1, rng=None):
        """rng: optional dict of explicit random draws: 'ray_offset' [1,R,2] (already
        minus 0.5), sampler draws 't_rand','u_final','perm','eik_idx', 'eik_uniform' [R,3], 'eik_jitter' [2R,3],
        and for background iterations 'bg_xy0' + 'bg' (a second sampler dict)."""
        rng = rng or {}
        rays = self.prepare_rays(input, rng)
        z_vals, z_samples_eik = self.sample(rays, rng)
        bg = self.prepare_background(input, rng) if self.wants_background(iter_step) else None
        return self.render(rays, z_vals, z_samples_eik, indices, rng=rng, bg=bg)

    def render(self, rays, z_vals, z_samples_eik, indices=None, rng=None, bg=None):
        rng = rng or {}
        ray_dirs, cam_loc, depth_scale, rot = rays["ray_dirs"], rays["cam_loc"], rays["depth_scale"], rays["rot"]
        dev = ray_dirs.device
        num_rays = ray_dirs.shape[0]
        N_samples = z_vals.shape[1]
        net = self.implicit_network
        if self.training and ray_dirs.is_cuda:
            self._warn_off_fused_path()
        if self.training and self.all_mesh_bbox_dict is not None:
            raise NotImplementedError("collision-driven Eikonal sampling belongs to Stage 2 (network.py:868-902)")
        n_main = num_rays * N_samples
        # Eikonal set (network.py:843-854), drawn up front so that ONE value+Jacobian pass serves the rendered points
        # and the Eikonal points together (half the trunk launches; the GEMMs simply get 4 % more rows)
        e0 = jitter = None
        eik_scale, eik_shift = 1.0, 0.0
        fused_points = ray_dirs.is_cuda and ops.COMPOSITE_IMPL == "hip"
        if self.training:
            b = float(self.scene_bounding_sphere)
            if "eik_uniform" in rng:
                e0 = rng["eik_uniform"].to(dev)
            elif "eik_uniform_u" in rng:     # raw U[0,1): uniform_(-b, b) = u * 2b - b, applied by the positions kernel when there is one
                e0 = rng["eik_uniform_u"]
                if fused_points:
                    eik_scale, eik_shift = 2.0 * b, -b
                else:
                    e0 = e0 * (2.0 * b) - b
            else:
                e0 = torch.empty(num_rays, 3, device=dev).uniform_(-b, b)
            jitter = rng["eik_jitter"].to(dev) if "eik_jitter" in rng else torch.rand(2 * num_rays, 3, device=dev)
        x01_all = None
        if fused_points:   # all positions, their grid coordinates and the view directions: one kernel
            n_all = n_main + (4 * num_rays if self.training else 0)
            x_all, x01_all = torch.empty(n_all, 3, device=dev), torch.empty(n_all, 3, device=dev)
            dirs_flat = torch.empty(n_main, 3, device=dev)
            _be._backend.render_points(cam_loc.contiguous(), ray_dirs.contiguous(), z_vals.contiguous(),
                                       z_samples_eik.reshape(-1).contiguous() if self.training else None,
                                       None if e0 is None else e0.contiguous().float(), None if jitter is None else jitter.contiguous().float(),
                                       float(net.divide_factor), x_all, x01_all, dirs_flat, eik_scale, eik_shift)
            points_flat = x_all[:n_main]
        else:
            points_flat = (cam_loc.unsqueeze(1) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
            dirs_flat = ray_dirs.unsqueeze(1).expand(-1, N_samples, -1).reshape(-1, 3)
            x_all = points_flat
            if self.training:
                near_surface = (cam_loc.unsqueeze(1) + z_samples_eik.unsqueeze(2) * ray_dirs.unsqueeze(1)).reshape(-1, 3)
                eik = torch.cat([e0, near_surface], 0)
                x_all = torch.cat([points_flat, eik, eik + (jitter - 0.5) * 0.01], 0)
        trunk_W = None
        hybrid = net.mlp_bf16 and bool(self.fp32_stages & {"trunk", "eikonal"})
        if hybrid:
            # per-stage precision (fp32_stages): the two point families of the trunk separately, each by its own precision's path
            def generic(xs):
                y_, J_ = net.sdf_and_jacobian(xs)
                return y_[:, :net.d_out], J_[:, :net.d_out]
            fused_ok = ops.TRUNK_IMPL == "mfma" and net._fused_trunk_supported(x_all) and net._lins()[2].out_features == net.d_out

            def fused(xs, n_m, x01s):
                enc = net.encoding
                l0, l1, l2 = net._lins()
                W0, W1, W2 = effective_weights([l0, l1, l2])
                return trunk_render(xs.detach(), n_m, enc.embeddings, enc.fused_offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                                    net.embedder.multires, float(net.divide_factor), W0, l0.bias, W1, l1.bias, W2, l2.bias, x01s)
            x01m = None if x01_all is None else x01_all[:n_main]
            x01e = None if x01_all is None else x01_all[n_main:]
            with self._stage_fp32("trunk") as forced:
                if forced or not fused_ok:
                    sdf_raw, J_main = generic(x_all[:n_main])
                    sdf, idx_min = sdf_raw.min(dim=-1, keepdim=True)
                    gradients = torch.gather(J_main, 1, idx_min.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
                else:
                    sdf_raw, sdf, idx_min, gradients = fused(x_all[:n_main], n_main, x01m)[:4]
            min_eik = gtheta = None
            y_eik, J_eik = sdf_raw[:0], sdf_raw.new_zeros(0, net.d_out, 3)
            if x_all.shape[0] > n_main:
                with self._stage_fp32("eikonal") as forced:
                    if forced or not fused_ok:
                        y_eik, J_eik = generic(x_all[n_main:])
                    else:
                        y_eik, min_eik, gtheta = fused(x_all[n_main:], 0, x01e)[4:]
        elif ops.TRUNK_IMPL == "mfma" and net._fused_trunk_supported(x_all) and net._lins()[2].out_features == net.d_out:
            enc = net.encoding
            l0, l1, l2 = net._lins()
            trunk_W = W0, W1, W2 = effective_weights([l0, l1, l2])
            sdf_raw, sdf, idx_min, gradients, y_eik, min_eik, gtheta = trunk_render(
                x_all.detach(), n_main, enc.embeddings, enc.fused_offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                net.embedder.multires, float(net.divide_factor), W0, l0.bias, W1, l1.bias, W2, l2.bias, x01_all)
        elif ops.FP32_TRUNK == "rr" and net._rr32_supported(x_all) and n_main > 0:
            # fp32: the rendered samples by the reverse-over-reverse closed form (rows = samples), the Eikonal points by value+Jacobian rows
            enc = net.encoding
            l0, l1, l2 = net._lins()
            W0, W1, W2 = effective_weights([l0, l1, l2])
            sdf_raw, sdf, idx_min, gradients = _trunk_rr32.apply(
                x_all[:n_main].detach(), None if x01_all is None else x01_all[:n_main], enc.embeddings, enc.offsets,
                float(np.log2(enc.per_level_scale)), int(enc.base_resolution), float(net.divide_factor), W0, l0.bias, W1, l1.bias, W2, l2.bias)
            if x_all.shape[0] > n_main:
                y_eik, J_eik = net.sdf_and_jacobian(x_all[n_main:])
                y_eik, J_eik = y_eik[:, :net.d_out], J_eik[:, :net.d_out]
            else:
                y_eik, J_eik = sdf_raw[:0], sdf_raw.new_zeros(0, net.d_out, 3)
            min_eik = gtheta = None
        else:
            y_all, J_all = net.sdf_and_jacobian(x_all)
            y_all, J_all = y_all[:, :net.d_out], J_all[:, :net.d_out]
            sdf_raw, J_main = y_all[:n_main], J_all[:n_main]
            sdf, idx_min = sdf_raw.min(dim=-1, keepdim=True)
            gradients = torch.gather(J_main, 1, idx_min.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
            y_eik, J_eik = y_all[n_main:], J_all[n_main:]
            min_eik = gtheta = None
        if not net.color_grid_feature:
            raise NotImplementedError("Stage-1 configs use color_grid_feature=True (confs/*/*.conf)")
        rgb = self._rgb_at(points_flat, dirs_flat, gradients, indices, None if x01_all is None else x01_all[:n_main]).reshape(-1, N_samples, 3)
        if ops.COMPOSITE_IMPL == "hip":
            if not z_vals.is_cuda:
                raise RuntimeError("fused compositing needs CUDA tensors (set HOLOSCENE_COMPOSITE_IMPL=torch explicitly for the "
                                   "whole-tensor formulation)")
            weights, _, rgb_values, depth_values, normal_cam, semantic_values, object_opacity = _composite.apply(
                z_vals, sdf, sdf_raw, rgb.reshape(-1, 3), gradients, self.density.get_beta(), depth_scale, self.implicit_network.sigmoid, rot)
            normal_world = None     # the kernel rotated the normal map into the camera frame
        elif ops.COMPOSITE_IMPL == "torch":
            semantic = (net.sigmoid * torch.sigmoid(-net.sigmoid * sdf_raw)).reshape(-1, N_samples, self.num_semantic)
            weights, transmittance, dists = self.volume_rendering(z_vals, sdf)
            object_opacity = self.occlusion_opacity(z_vals, transmittance, dists, sdf_raw).sum(-1).transpose(0, 1)
            rgb_values = torch.sum(weights.unsqueeze(-1) * rgb, 1)
            semantic_values = torch.sum(weights.unsqueeze(-1) * semantic, 1)
            depth_values = depth_scale * (torch.sum(weights * z_vals, 1, keepdims=True) / (weights.sum(dim=1, keepdims=True) + 1e-8))
            normals = (gradients / (gradients.norm(2, -1, keepdim=True) + 1e-6)).reshape(-1, N_samples, 3)
            normal_world = torch.sum(weights.unsqueeze(-1) * normals, 1)
        else:
            raise RuntimeError(f"unknown HOLOSCENE_COMPOSITE_IMPL={ops.COMPOSITE_IMPL!r}")
        if self.white_bkgd:
            rgb_values = rgb_values + (1.0 - torch.sum(weights, -1)[..., None]) * self.bg_color.unsqueeze(0)

        output = _Outputs({
            "rgb": rgb,
            "semantic_values": semantic_values,
            "object_opacity": object_opacity,
            "rgb_values": rgb_values,
            "depth_values": depth_values,
            "z_vals": z_vals,
            "sdf": sdf.reshape(z_vals.shape),
            "weights": weights,
        })

        output.defer("depth_vals", lambda: z_vals * depth_scale)     # nothing in training reads it: evaluated on first access
        if self.training:
            # replaces gradient() + get_sdf_raw() + get_sdf_vals() on the Eikonal set (network.py:856-863)
            if gtheta is not None:      # already stacked by the split kernel
                y, min_sdf, grad_theta = y_eik, min_eik, gtheta
                output["grad_theta_all"] = grad_theta    # (the fused objective takes the two halves from this one tensor)
            else:
                y, J = y_eik, J_eik
                min_sdf, idx = y.min(dim=-1, keepdim=True)
                g_min = torch.gather(J, 1, idx.unsqueeze(-1).expand(-1, 1, 3)).squeeze(1)
                grad_theta = torch.cat([J.transpose(0, 1).reshape(-1, 3), g_min], 0)
            if self.eikonal_mode == "fd":
                grad_theta = self._eikonal_gradients_fd(x_all[n_main:], y)
                output.pop("grad_theta_all", None)
            output["sample_sdf"] = y
            output["sample_minsdf"] = min_sdf
            half = grad_theta.shape[0] // 2  # quirk Q2: halves of the stacked rows, not original/jittered
            output["grad_theta"] = grad_theta[:half]
            output["grad_theta_nei"] = grad_theta[half:]

        output["normal_map"] = normal_cam if normal_world is None else (rot @ normal_world.permute(1, 0)).permute(1, 0).contiguous()

        if bg is not None:  # background-surface pass (network.py:943-968)
            bg_z, ray_dirs0, cam_loc0 = bg["z_vals"], bg["ray_dirs"], bg["cam_loc"]
            n_bg = bg_z.shape[1]
            fused_bg = (ops.BG_IMPL == "hip" and fused_points and ops.COMPOSITE_IMPL == "hip" and ops.TRUNK_IMPL == "mfma"
                        and net._fused_trunk_supported(bg_z) and net._lins()[2].out_features == net.d_out)
            if fused_bg:
                # the same kernels as the main pass: positions, value+Jacobian trunk + split (no Eikonal rows), compositing -- once
                # with the scene SDF for the label map (forward only), once with object 0's SDF for depth and normals
                B0 = bg_z.numel()
                xb, xb01 = torch.empty(B0, 3, device=dev), torch.empty(B0, 3, device=dev)
                _be._backend.ray_points(cam_loc0.contiguous(), ray_dirs0.contiguous(), bg_z.contiguous(), xb, xb01, float(net.divide_factor))
                enc = net.encoding
                l0, l1, l2 = net._lins()
                W0, W1, W2 = trunk_W if trunk_W is not None else effective_weights([l0, l1, l2])   # the main pass's normalised weights
                raw_b, sdf_b, _, grad_b, _, _, _ = trunk_render(
                    xb, B0, enc.embeddings, enc.fused_offsets, float(np.log2(enc.per_level_scale)), int(enc.base_resolution),
                    net.embedder.multires, float(net.divide_factor), W0, l0.bias, W1, l1.bias, W2, l2.bias, xb01)
                beta, unused_rgb = self.density.get_beta(), grad_b.detach()     # (the colour slot of the kernel is not needed here)
                with torch.no_grad():
                    bg_semantic = _composite.apply(bg_z, sdf_b, raw_b, unused_rgb, grad_b, beta, bg["depth_scale"], net.sigmoid)[5]
                output["bg_mask"] = torch.argmax(bg_semantic, dim=-1, keepdim=True)
                comp = _composite.apply(bg_z, raw_b[:, 0:1], raw_b, unused_rgb, grad_b, beta, bg["depth_scale"], net.sigmoid, rot)
                output["bg_depth_values"], output["bg_normal_map"] = comp[3], comp[4]      # (camera frame: rotated by the kernel)
            elif ops.BG_IMPL not in ("hip", "torch"):
                raise RuntimeError(f"unknown HOLOSCENE_BG_IMPL={ops.BG_IMPL!r}")
            else:
                bg_points = (cam_loc0.unsqueeze(1) + bg_z.unsqueeze(2) * ray_dirs0.unsqueeze(1)).reshape(-1, 3)
                scene_sdf, _, bg_gradients, scene_semantic, bg_sdf = self.implicit_network.get_specific_outputs(bg_points, 0)
                bg_weight, _, _ = self.volume_rendering(bg_z, bg_sdf)
                scene_weight, _, _ = self.volume_rendering(bg_z, scene_sdf)  # semantics use the scene SDF
                bg_semantic = torch.sum(scene_weight.unsqueeze(-1) * scene_semantic.reshape(-1, n_bg, self.num_semantic), 1)
                output["bg_mask"] = torch.argmax(bg_semantic, dim=-1, keepdim=True)
                output["bg_depth_values"] = bg["depth_scale"] * (torch.sum(bg_weight * bg_z, 1, keepdims=True) / (bg_weight.sum(dim=1, keepdims=True) + 1e-8))
                bg_normals = (bg_gradients / (bg_gradients.norm(2, -1, keepdim=True) + 1e-6)).reshape(-1, n_bg, 3)
                bg_normal_map = torch.sum(bg_weight.unsqueeze(-1) * bg_normals, 1)
                output["bg_normal_map"] = (rot @ bg_normal_map.permute(1, 0)).permute(1, 0).contiguous()
        return output
