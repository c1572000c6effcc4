"""Laplace-CDF density of VolSDF (reference: model/density.py:5-30)."""
import contextlib

import torch
import torch.nn as nn


def laplace_density(sdf, beta):
    """sigma = (1/beta) * (0.5 + 0.5 * sign(s) * expm1(-|s|/beta))   (density.py:21-26)"""
    return (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


class _abs_shift(torch.autograd.Function):
    """|beta| + beta_min and its backward as one launch each (csrc/small_ops.hip) -- autograd's abs, add / sgn, mul are four ~5 us
    launches inside the replayed training graph for one float."""

    @staticmethod
    def forward(ctx, beta, beta_min):
        from ..hashencoder.backend import _backend
        ctx.save_for_backward(beta)
        return _backend.abs_shift(beta.detach().contiguous(), beta_min)

    @staticmethod
    def backward(ctx, g):
        from ..hashencoder.backend import _backend
        (beta,) = ctx.saved_tensors
        return _backend.abs_shift(beta.detach().contiguous(), gy=g.contiguous().float()), None


def _beta(beta, beta_min):
    if beta.is_cuda and beta.dtype == torch.float32:
        return _abs_shift.apply(beta, beta_min)
    return beta.abs() + beta_min


class Density(nn.Module):
    def __init__(self, params_init={}):
        super().__init__()
        for name, value in params_init.items():
            setattr(self, name, nn.Parameter(torch.tensor(value)))

    def forward(self, sdf, beta=None):
        return self.density_func(sdf, beta=beta)


class LaplaceDensity(Density):
    def __init__(self, params_init={}, beta_min=0.0001):
        super().__init__(params_init=params_init)
        self.register_buffer("beta_min", torch.tensor(beta_min), persistent=False)

    def density_func(self, sdf, beta=None):
        if beta is None:
            beta = self.get_beta()
        return laplace_density(sdf, beta)

    _shared = None

    def get_beta(self):
        if self._shared is not None:
            return self._shared
        return _beta(self.beta, self.beta_min)

    @contextlib.contextmanager
    def shared_beta(self):
        """Within the block every get_beta() returns ONE tensor evaluated on entry (with its autograd history): an iteration asks
        for beta in the sampler, the background sampler and the renderer, and the parameter cannot change in between."""
        self._shared = _beta(self.beta, self.beta_min)
        try:
            yield self._shared
        finally:
            self._shared = None
