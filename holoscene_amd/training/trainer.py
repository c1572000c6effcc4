"""Stage-1 training step -- the loop body of HoloSceneTrainRunner.run() (training/holoscene_train.py:332-428)
without its logging: zero_grad, forward, loss, backward, (gradient exchange), Adam, LR step.

No ``.item()`` inside the step (the reference pays ~25 device syncs per iteration for its grad-norm
print, holoscene_train.py:366-372); scalars are returned as device tensors.
"""
import torch

from ..model.loss import HoloSceneLoss
from ..model.network import HoloSceneNetwork
from ..utils.conf import Conf
from . import distributed as dist_util
from .optim import build_optimizer, build_scheduler


def stock_conf(num_rays=1024, S=128, d_out=32, num_levels=16, base_size=16, end_size=2048, logmap=19, beta=0.1, use_bg_reg=True,
               mlp_precision="fp32"):
    """confs/replica/room_0/replica_room_0.conf with 'R rays x S samples' mapped as SURVEY D5:
    N_samples_eval=S, N_samples=S/2, N_samples_extra=S/4."""
    return Conf(
        train=Conf(learning_rate=5.0e-4, lr_factor_for_grid=20.0, num_pixels=num_rays, add_objectvio_iter=25000, max_total_iters=200000,
                   stop_iter=100000, sched_decay_rate=0.1),
        loss=Conf(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, smooth_weight=0.005, depth_weight=0.5, normal_l1_weight=0.05,
                  normal_cos_weight=0.05, semantic_loss="torch.nn.MSELoss", use_obj_opacity=True, semantic_weight=5.0, reg_vio_weight=0.01,
                  bg_reg_weight=0.01, depth_type="marigold"),
        model=Conf(
            feature_vector_size=256, scene_bounding_sphere=1.0, use_bg_reg=use_bg_reg, render_bg_iter=10, mlp_precision=mlp_precision,
            implicit_network=Conf(d_in=3, d_out=d_out, dims=[256, 256], geometric_init=True, bias=0.9, skip_in=[4], weight_norm=True,
                                  multires=6, inside_outside=True, use_grid_feature=True, divide_factor=1.0, sigmoid=10,
                                  color_grid_feature=True, num_levels=num_levels, base_size=base_size, end_size=end_size, logmap=logmap),
            rendering_network=Conf(mode="idr", d_in=9, d_out=3, dims=[256, 256], weight_norm=True, multires_view=4, multires_point=4,
                                   multires_normal=4),
            density=Conf(params_init=Conf(beta=beta), beta_min=0.0001),
            ray_sampler=Conf(near=0.0, N_samples=S // 2, N_samples_eval=S, N_samples_extra=S // 4, eps=0.1, beta_iters=10, max_total_iters=5)))


def benchmark_model_state(model, beta, seed=42):
    """Measurement state of SURVEY 8(d): reference init, then the lin0 columns geometric init zeroes are
    refilled with N(0, 1e-2) (leaves the dead-gradient state, quirk Q5) and beta is set explicitly."""
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        v = model.implicit_network.lin0.weight_v
        v[:, 3:] = (torch.randn(v[:, 3:].shape, generator=g) * 1e-2).to(v.device)
        model.density.beta.fill_(beta)


class Stage1Trainer:
    def __init__(self, conf, device="cuda", num_images=8, seed=42, world_size=1, fused_adam=None):
        torch.manual_seed(seed)
        self.conf = conf
        self.device = torch.device(device)
        self.world_size = world_size
        self.model = HoloSceneNetwork(conf=conf.get_config("model"), graph_node_dict=None, num_images=num_images).to(self.device)
        self.loss = HoloSceneLoss(**conf.get_config("loss"))
        self.lr = conf.get_float("train.learning_rate")
        self.optimizer = build_optimizer(self.model, self.lr, conf.get_float("train.lr_factor_for_grid", default=1.0), fused=fused_adam)
        # nepochs * ds_len with ds_len = fix_length: decay_steps == max_total_iters (holoscene_train.py:110-116, 166-169)
        self.scheduler = build_scheduler(self.optimizer, conf.get_float("train.sched_decay_rate", default=0.1),
                                         conf.get_int("train.max_total_iters", default=200000))
        self.add_objectvio_iter = conf.get_int("train.add_objectvio_iter", default=100000)
        self.iter_step = 0

    def train_step(self, indices, model_input, ground_truth, rng=None):
        self.model.train()
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(model_input, indices, iter_step=self.iter_step, rng=rng)
        out["iter_step"] = self.iter_step
        loss_out = self.loss(out, ground_truth, call_reg=self.iter_step >= self.add_objectvio_iter)
        loss_out["loss"].backward()
        if self.world_size > 1:
            dist_util.average_gradients(self.model.parameters(), self.world_size)
        self.optimizer.step()
        self.scheduler.step()
        self.iter_step += 1
        return out, loss_out
