"""Gradients of one proposal level's density field (oracle; test infrastructure only, see oracle/__init__.py).

PARITY UNPINNED like the rest of the oracle: jax is not available here, so what the reference's
`jax.value_and_grad(loss_fn)` (internal/train_utils.py:3128-3131) produces for the sub-graph
HashEncoding.__call__ (grid_utils.py:808-905) -> DensityMLP.run_network (geometry.py:155-168) ->
convert_raw_density (geometry.py:318-341) is restated as reverse-mode autodiff (torch.autograd) of the oracle's own
forward functions -- the same transposition jax performs: scatter-add into the tables for the trilinear gathers,
ReLU masks, safe_exp's custom derivative rule.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import hashgrid_ref, mathx
from .cache_ref import P, dense


class _SafeExp(torch.autograd.Function):
    """math.safe_exp with its custom_jvp (internal/math.py:153-171, 186-192): y = exp(clip(x, min, 70)),
    y_dot = y * x_dot -- the clip does not gate the gradient."""

    @staticmethod
    def forward(ctx, x):
        y = mathx.safe_exp(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return g * y


def density_backward(weights: Dict[str, torch.Tensor], cfg, level: int, means: torch.Tensor, d_density: torch.Tensor,
                     d_feature: torch.Tensor = None):
    """-> (grads {tensor name: gradient}, density [n], feature [n, 64]) for
    L = sum(d_density * density) + sum(d_feature * feature) over the parameters of Cache/Sampler/MLP_<level>."""
    gcfg = cfg.proposal_grids[level]
    base = f"Cache/Sampler/MLP_{level}"
    names = [k for k in weights if k.startswith(f"{P}{base}/density_grid/") or
             any(k.startswith(f"{P}{base}/{layer}/") for layer in ("density_layers_0", "density_layers_1", "output_density_layer"))]
    w = dict(weights)
    for k in names:
        w[k] = weights[k].detach().clone().requires_grad_(True)
    warped = mathx.contract_radius(means, cfg.contract_radius)
    x = hashgrid_ref.hash_encoding(w, f"{P}{base}/density_grid", gcfg, warped)
    h = torch.relu(dense(w, f"{base}/density_layers_0", x))
    h = torch.relu(dense(w, f"{base}/density_layers_1", h))
    raw = dense(w, f"{base}/output_density_layer", h)[..., 0]
    density = _SafeExp.apply(raw + cfg.density_bias)
    valid = ((warped > -gcfg.bbox) & (warped < gcfg.bbox)).all(dim=-1)
    density = torch.where(valid, density, torch.zeros_like(density))
    loss = (d_density * density).sum()
    if d_feature is not None:
        loss = loss + (d_feature * h).sum()
    grads = torch.autograd.grad(loss, [w[k] for k in names], allow_unused=True)
    out = {k: (torch.zeros_like(w[k]) if g is None else g) for k, g in zip(names, grads)}
    return out, density.detach(), h.detach()


def relu_margin(weights, cfg, level: int, means: torch.Tensor) -> torch.Tensor:
    """min |pre-activation| over the two hidden layers per point: how far a point is from a ReLU kink, where the
    gradient is discontinuous and float32 / float64 evaluations may legitimately land on different sides."""
    gcfg = cfg.proposal_grids[level]
    base = f"Cache/Sampler/MLP_{level}"
    warped = mathx.contract_radius(means, cfg.contract_radius)
    x = hashgrid_ref.hash_encoding(weights, f"{P}{base}/density_grid", gcfg, warped)
    z0 = dense(weights, f"{base}/density_layers_0", x)
    z1 = dense(weights, f"{base}/density_layers_1", torch.relu(z0))
    return torch.minimum(z0.abs().min(dim=-1).values, z1.abs().min(dim=-1).values)
