"""Optimiser / LR schedule wiring of the Stage-1 trainer (reference: training/holoscene_train.py:152-169)."""
import torch


def build_optimizer(model, lr, lr_factor_for_grid=1.0, fused=None):
    """Adam with the reference's three groups: hash grids at lr*factor, MLPs and beta at lr;
    betas (0.9, 0.99), eps 1e-15."""
    groups = [
        {"name": "encoding", "params": list(model.implicit_network.grid_parameters()), "lr": lr * lr_factor_for_grid},
        {"name": "net", "params": list(model.implicit_network.mlp_parameters()) + list(model.rendering_network.parameters()), "lr": lr},
        {"name": "density", "params": list(model.density.parameters()), "lr": lr},
    ]
    kw = {}
    if fused is not None:
        kw["fused"] = fused
    return torch.optim.Adam(groups, betas=(0.9, 0.99), eps=1e-15, **kw)


def build_scheduler(optimizer, decay_rate, decay_steps):
    """ExponentialLR with gamma = decay_rate^(1/decay_steps), stepped every iteration (holoscene_train.py:166-169, 428)."""
    return torch.optim.lr_scheduler.ExponentialLR(optimizer, decay_rate ** (1.0 / decay_steps))
