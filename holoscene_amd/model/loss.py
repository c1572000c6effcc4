"""Stage-1 objective: MonoSDFLoss / HoloSceneLoss with the reference's constructor arguments,
``forward`` contract and returned keys (model/loss.py:196-346, :349-666).

Written without data-dependent Python branches: where the reference tests a device scalar in an
``if`` (``collision_cnt > 0`` loss.py:399, ``divisor == 0`` :533, ...) -- one host sync each -- this
version divides by ``clamp(count, 1)`` so the value is identical and the stream never stalls.
"""
import math
import os

import torch
import torch.nn.functional as F
from torch import nn

from ..hashencoder import backend as _be
from ..utils.conf import get_class

# "hip": value + analytic gradient of the per-ray and Eikonal/smoothness terms in two fused kernels (csrc/loss.hip);
# "torch": the whole-tensor formulation below (A/B reference, and what CPU host-logic tests select explicitly).
LOSS_IMPL = "hip"


_ZERO = {}
_ONE = {}


def unit_cotangent(dev):
    """The shared 0-d tensor 1.0 of a device.  `loss.backward(gradient=unit_cotangent(dev))` lets the fused objective recognise the
    plain d loss / d loss = 1 root by its storage and hand out its stored gradients as they are (no fill for the root, no multiply by
    one); any other cotangent takes the general path.  Never written to."""
    o = _ONE.get(str(dev))
    if o is None:
        o = _ONE[str(dev)] = torch.ones((), device=dev)
    return o


def _zero_scalar(dev):
    """A shared 0-d zero per device for the terms that are switched off (never written to): no fill launch per iteration."""
    z = _ZERO.get(str(dev))
    if z is None:
        z = _ZERO[str(dev)] = torch.zeros((), device=dev)
    return z


class _fused_core_loss(torch.autograd.Function):
    """(rgb_values, depth_values, normal_map, object_opacity, grad_theta, grad_theta_nei) -> weighted sum of the rgb,
    depth, normal, opacity, Eikonal and smoothness terms + the 7 unweighted terms (for logging, not differentiable)."""

    @staticmethod
    def forward(ctx, rgb, depth, nmap, opac, g1, g2, sdf, rgb_gt, depth_gt, n_gt, gt_mask, segs, weights):
        ctx.set_materialize_grads(False)     # (the seven logging terms would otherwise reach backward as a zero-FILLED tensor: one launch)
        w_rgb, w_depth, w_l1, w_cos, w_opac, w_eik, w_smooth = weights
        dev = rgb.device
        R = rgb.shape[0]
        c = lambda t: t.detach().contiguous().float()  # noqa: E731
        rgb_, depth_, nmap_, opac_ = c(rgb), c(depth).reshape(-1), c(nmap), c(opac)
        ctx.stacked = g2 is None
        if ctx.stacked:     # g1 = ALL stacked gradient rows (render's "grad_theta_all"): the halves are views, and so are their cotangents
            g_all = c(g1)
            half = g_all.shape[0] // 2
            g1_, g2_ = g_all[:half], g_all[half:]
            d_all = torch.empty_like(g_all)
            d_g1, d_g2 = d_all[:half], d_all[half:]
        else:
            g1_, g2_ = c(g1), c(g2)
            d_g1, d_g2 = torch.empty_like(g1_), torch.empty_like(g2_)
        out8 = torch.empty(8, device=dev)   # the seven unweighted terms (rgb, depth, n_l1, n_cos, opacity, eikonal, smooth) + weighted total
        g_rgb, g_depth, g_nmap, g_opac = torch.empty_like(rgb_), torch.empty_like(depth_), torch.empty_like(nmap_), torch.empty_like(opac_)
        _be._backend.loss_stage1(rgb_, c(rgb_gt).reshape(-1, 3), depth_, c(depth_gt).reshape(-1), nmap_, c(n_gt).reshape(-1, 3), c(gt_mask).reshape(-1),
                                 c(sdf), opac_, segs.reshape(-1).long().contiguous(), g1_, g2_, weights, out8, g_rgb, g_depth, g_nmap, g_opac, d_g1, d_g2)
        terms, total = out8[:7], out8[7]
        if ctx.stacked:
            ctx.save_for_backward(g_rgb, g_depth.reshape(depth.shape), g_nmap, g_opac, d_all)
        else:
            ctx.save_for_backward(g_rgb, g_depth.reshape(depth.shape), g_nmap, g_opac, d_g1, d_g2)
        ctx.mark_non_differentiable(terms)
        return total, terms

    @staticmethod
    def backward(ctx, g, _g_terms):
        if g is None:
            return (None,) * 13
        one = _ONE.get(str(g.device))
        if one is not None and g.data_ptr() == one.data_ptr():    # the root cotangent 1.0 (unit_cotangent): the stored gradients are the answer
            grads = tuple(ctx.saved_tensors)
        else:
            grads = tuple(torch._foreach_mul(list(ctx.saved_tensors), g))   # one launch for all cotangents
        if ctx.stacked:
            grads = grads + (None,)
        return grads + (None,) * 7


class _fused_bg_smooth(torch.autograd.Function):
    """HoloSceneLoss.get_bg_render_loss in one launch (csrc/loss.hip: k_bg_smooth): value and analytic gradient."""

    @staticmethod
    def forward(ctx, bg_depth, bg_normal, labels, side):
        d = bg_depth.detach().reshape(-1).contiguous().float()
        n = bg_normal.detach().reshape(-1, 3).contiguous().float()
        out = torch.empty(1, device=d.device)
        g_d, g_n = torch.empty_like(d), torch.empty_like(n)
        _be._backend.bg_smooth_loss(d, n, labels.reshape(-1).long().contiguous(), side, out, g_d, g_n)
        ctx.save_for_backward(g_d.reshape(bg_depth.shape), g_n.reshape(bg_normal.shape))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        g_d, g_n = torch._foreach_mul(list(ctx.saved_tensors), g)
        return g_d, g_n, None, None


def compute_scale_and_shift_batch(prediction, target):
    """Least-squares (scale, shift) aligning prediction [B,N] to target [B,N] (loss.py:181-193)."""
    ones = torch.ones_like(prediction)
    a00 = (prediction * prediction).sum(1)
    a01 = prediction.sum(1)
    a11 = ones.sum(1)
    b0 = (prediction * target).sum(1)
    b1 = target.sum(1)
    # closed-form inverse of the 2x2 normal matrix (the reference calls torch.inverse, loss.py:190; an LU routine
    # with a host-side status check is neither needed for 2x2 nor capturable in a HIP graph)
    det = a00 * a11 - a01 * a01
    return (a11 * b0 - a01 * b1) / det, (a00 * b1 - a01 * b0) / det


def _resolve(cls_or_name):
    if isinstance(cls_or_name, str):
        if cls_or_name.startswith("torch.nn."):
            return getattr(nn, cls_or_name.split(".")[-1])
        return get_class(cls_or_name)
    return cls_or_name


class MonoSDFLoss(nn.Module):
    def __init__(self, rgb_loss, eikonal_weight, smooth_weight=0.005, depth_weight=0.1, normal_l1_weight=0.05, normal_cos_weight=0.05,
                 uncertainty_begin_iter=20000000, depth_type="marigold", phy_un_weight=50, end_step=-1):
        super().__init__()
        self.eikonal_weight = eikonal_weight
        self.smooth_weight = smooth_weight
        self.depth_weight = depth_weight
        self.normal_l1_weight = normal_l1_weight
        self.normal_cos_weight = normal_cos_weight
        self.uncertainty_begin_iter = uncertainty_begin_iter
        self.depth_type = depth_type
        self.phy_un_weight = phy_un_weight
        self.rgb_loss = _resolve(rgb_loss)(reduction="mean")
        self.step = 0
        self.end_step = end_step

    def get_rgb_loss(self, rgb_values, rgb_gt):
        return self.rgb_loss(rgb_values, rgb_gt.reshape(-1, 3))

    def get_eikonal_loss(self, grad_theta):
        return ((grad_theta.norm(2, dim=1) - 1) ** 2).mean()

    def get_smooth_loss(self, model_outputs):
        g1, g2 = model_outputs["grad_theta"], model_outputs["grad_theta_nei"]
        n1 = g1 / (g1.norm(2, dim=1).unsqueeze(-1) + 1e-5)
        n2 = g2 / (g2.norm(2, dim=1).unsqueeze(-1) + 1e-5)
        return torch.norm(n1 - n2, dim=-1).mean()

    def get_depth_loss(self, depth_pred, depth_gt):
        depth_pred = depth_pred.reshape(1, -1)
        depth_gt = depth_gt.reshape(1, -1)
        w, q = compute_scale_and_shift_batch(depth_pred, depth_gt)
        diff = ((w.reshape(-1, 1) * depth_pred + q.reshape(-1, 1)) - depth_gt) ** 2
        return torch.clip(diff, max=1).mean()

    def get_normal_loss(self, normal_pred, normal_gt):
        normal_gt = F.normalize(normal_gt, p=2, dim=-1)
        normal_pred = F.normalize(normal_pred, p=2, dim=-1)
        l1 = torch.abs(normal_pred - normal_gt).sum(dim=-1).mean()
        cos = (1.0 - torch.sum(normal_pred * normal_gt, dim=-1)).mean()
        return l1, cos

    def forward(self, model_outputs, ground_truth):
        dev = model_outputs["rgb_values"].device
        rgb_gt = ground_truth["rgb"].to(dev)
        depth_gt = ground_truth["depth"].to(dev)
        normal_gt = ground_truth["normal"].to(dev)
        zero = torch.zeros((), device=dev)
        rgb_loss = self.get_rgb_loss(model_outputs["rgb_values"], rgb_gt)
        eikonal_loss = self.get_eikonal_loss(model_outputs["grad_theta"]) if "grad_theta" in model_outputs else zero
        # supervise normals only on rays that cross a surface (sign change of the SDF along the ray)
        sdf = model_outputs["sdf"]
        mask = ((sdf > 0.0).any(dim=-1) & (sdf < 0.0).any(dim=-1))[None, :, None] & (ground_truth["mask"].to(dev) > 0.5)
        depth_loss = self.get_depth_loss(model_outputs["depth_values"], depth_gt) if self.depth_weight > 0 else zero
        normal_l1, normal_cos = self.get_normal_loss(model_outputs["normal_map"][None] * mask, normal_gt)
        smooth_loss = self.get_smooth_loss(model_outputs)
        decay = math.exp(-self.step / self.end_step * 10.0) if self.end_step > 0 else 1.0
        self.step += 1
        loss = rgb_loss + self.eikonal_weight * eikonal_loss + self.smooth_weight * smooth_loss + decay * self.depth_weight * depth_loss \
            + decay * self.normal_l1_weight * normal_l1 + decay * self.normal_cos_weight * normal_cos
        return {"loss": loss, "rgb_loss": rgb_loss, "eikonal_loss": eikonal_loss, "smooth_loss": smooth_loss, "depth_loss": depth_loss,
                "normal_l1": normal_l1, "normal_cos": normal_cos}


class HoloSceneLoss(MonoSDFLoss):
    def __init__(self, rgb_loss, eikonal_weight, semantic_weight=0.04, smooth_weight=0.005,
                 semantic_loss=torch.nn.CrossEntropyLoss(ignore_index=-1), depth_weight=0.1, normal_l1_weight=0.05,
                 normal_cos_weight=0.05, reg_vio_weight=0.1, use_obj_opacity=True, bg_reg_weight=0.1, depth_type="marigold", end_step=-1):
        super().__init__(rgb_loss=rgb_loss, eikonal_weight=eikonal_weight, smooth_weight=smooth_weight, depth_weight=depth_weight,
                         normal_l1_weight=normal_l1_weight, normal_cos_weight=normal_cos_weight, depth_type=depth_type, end_step=end_step)
        self.semantic_weight = semantic_weight
        self.bg_reg_weight = bg_reg_weight
        if isinstance(semantic_loss, nn.Module):
            self.semantic_loss = torch.nn.CrossEntropyLoss(ignore_index=-1, reduction="none")
        else:
            self.semantic_loss = _resolve(semantic_loss)(reduction="none")
        self.reg_vio_weight = reg_vio_weight
        self.use_obj_opacity = use_obj_opacity

    def get_semantic_loss(self, semantic_value, semantic_gt):
        return self.semantic_loss(semantic_value, semantic_gt.squeeze()).mean()

    def object_distinct_loss(self, sdf_value, min_sdf):
        """Penalise any non-minimal object that is also 'inside' where the scene SDF is negative (loss.py:389-403)."""
        arg = sdf_value.argmin(dim=1, keepdim=True)
        viol = torch.relu(-sdf_value - min_sdf.detach())
        viol = viol.scatter(1, arg, 0.0)  # the minimal object itself is exempt
        count = (viol > 0).sum()
        return viol.sum() / count.clamp(min=1)

    def object_opacity_loss(self, predict_opacity, gt_opacity, weight=None):
        target = F.one_hot(gt_opacity.reshape(-1), num_classes=predict_opacity.shape[1]).float()
        predict_opacity = torch.clip(predict_opacity, 1e-4, 1 - (1e-4))
        return F.binary_cross_entropy(predict_opacity, target, reduction="none").mean(dim=-1).mean()

    def compute_grad_error(self, x, mask):
        """Multi-scale masked first-difference magnitude (loss.py:519-547)."""
        total = torch.zeros((), device=x.device)
        for i in range(4):
            step = 2 ** i
            m = mask[:, ::step, ::step]
            v = m * x[:, ::step, ::step]
            gx = (m[:, :, 1:] * m[:, :, :-1]) * torch.abs(v[:, :, 1:] - v[:, :, :-1])
            gy = (m[:, 1:, :] * m[:, :-1, :]) * torch.abs(v[:, 1:, :] - v[:, :-1, :])
            divisor = m[:1].sum()
            total = total + torch.where(divisor > 0, (gx.sum() + gy.sum()) / divisor.clamp(min=1), torch.zeros_like(total))
        return total

    def get_bg_render_loss(self, bg_depth, bg_normal, mask, labels=None):
        """labels: the integer map the mask came from (mask = labels != 0); given on CUDA, the fused kernel is used."""
        if labels is not None and LOSS_IMPL == "hip" and bg_depth.is_cuda:
            return _fused_bg_smooth.apply(bg_depth, bg_normal, labels, 32)
        bg_depth = bg_depth.reshape(1, 32, 32)
        bg_normal = bg_normal.reshape(32, 32, 3).permute(2, 0, 1)
        mask = mask.reshape(1, 32, 32)
        return self.compute_grad_error(bg_depth, mask) + self.compute_grad_error(bg_normal, mask.repeat(3, 1, 1))

    def _forward_fused(self, model_outputs, ground_truth):
        """MonoSDFLoss.forward + the opacity term through csrc/loss.hip (same keys, same values)."""
        decay = math.exp(-self.step / self.end_step * 10.0) if self.end_step > 0 else 1.0
        self.step += 1
        weights = (1.0, decay * self.depth_weight, decay * self.normal_l1_weight, decay * self.normal_cos_weight,
                   self.semantic_weight, self.eikonal_weight, self.smooth_weight)
        stacked = model_outputs.get("grad_theta_all")
        total, t = _fused_core_loss.apply(model_outputs["rgb_values"], model_outputs["depth_values"], model_outputs["normal_map"],
                                          model_outputs["object_opacity"], model_outputs["grad_theta"] if stacked is None else stacked,
                                          model_outputs["grad_theta_nei"] if stacked is None else None,
                                          model_outputs["sdf"], ground_truth["rgb"], ground_truth["depth"], ground_truth["normal"],
                                          ground_truth["mask"], ground_truth["segs"], weights)
        return {"loss": total, "rgb_loss": t[0], "depth_loss": t[1], "normal_l1": t[2], "normal_cos": t[3], "eikonal_loss": t[5],
                "smooth_loss": t[6]}, t[4]

    def forward(self, model_outputs, ground_truth, call_reg=False, call_bg_reg=False):
        fused = (LOSS_IMPL == "hip" and self.use_obj_opacity and "object_opacity" in model_outputs and "grad_theta" in model_outputs
                 and self.depth_weight > 0 and isinstance(self.rgb_loss, nn.L1Loss) and "rgb_offset" not in model_outputs)
        if fused:
            if not model_outputs["rgb_values"].is_cuda:
                raise RuntimeError("fused loss needs CUDA tensors (set HOLOSCENE_LOSS_IMPL=torch explicitly for the whole-tensor formulation)")
            output, fused_semantic = self._forward_fused(model_outputs, ground_truth)
        else:
            output, fused_semantic = super().forward(model_outputs, ground_truth), None
        dev = output["loss"].device
        zero = _zero_scalar(dev)
        if fused_semantic is not None:
            semantic_loss = fused_semantic
        elif "semantic_values" in model_outputs and not self.use_obj_opacity:
            semantic_loss = self.get_semantic_loss(model_outputs["semantic_values"], ground_truth["segs"].to(dev).long())
        elif "object_opacity" in model_outputs and self.use_obj_opacity:
            semantic_loss = self.object_opacity_loss(model_outputs["object_opacity"], ground_truth["segs"].to(dev).long())
        else:
            semantic_loss = zero
        if "sample_sdf" in model_outputs and call_reg:
            if "collision_relations" in model_outputs:
                raise NotImplementedError("scene-graph collision loss is driven by the Stage-2 trainer (loss.py:405-484)")
            sample_sdf_loss = self.object_distinct_loss(model_outputs["sample_sdf"], model_outputs["sample_minsdf"])
        else:
            sample_sdf_loss = zero
        if "bg_depth_values" in model_outputs:
            labels = model_outputs["bg_mask"] if "bg_mask" in model_outputs else ground_truth["segs"].to(dev)
            if LOSS_IMPL == "hip" and labels.is_cuda:
                bg_mask = None                                   # the fused kernel tests labels != 0 itself
            elif "bg_mask" in model_outputs:
                bg_mask = (labels != 0).int()  # smooth only where something occludes the background
            else:
                bg_mask = labels != 0
            background_reg_loss = self.get_bg_render_loss(model_outputs["bg_depth_values"], model_outputs["bg_normal_map"], bg_mask, labels=labels)
        else:
            background_reg_loss = zero
        if "rgb_offset" in model_outputs:
            rgb_offset_loss = torch.mean(model_outputs["rgb_offset"] ** 2)
            output["loss"] = output["loss"] + rgb_offset_loss
            output["rgb_offset_loss"] = rgb_offset_loss
        output["semantic_loss"] = semantic_loss
        output["collision_reg_loss"] = sample_sdf_loss
        output["background_reg_loss"] = background_reg_loss
        if fused_semantic is None:   # (the fused core already contains semantic_weight * semantic_loss)
            output["loss"] = output["loss"] + self.semantic_weight * semantic_loss
        if sample_sdf_loss is not zero:     # switched-off terms add nothing: skip their launches
            output["loss"] = output["loss"] + self.reg_vio_weight * sample_sdf_loss
        if background_reg_loss is not zero:
            output["loss"] = output["loss"] + self.bg_reg_weight * background_reg_loss
        return output
