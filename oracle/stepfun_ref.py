"""Step-function sampling (oracle; see oracle/__init__.py).

Follows internal/stepfun.py:125-250, 306-314 and internal/math.py:412-457.
"""
from __future__ import annotations

import numpy as np
import torch

from . import mathx


def integrate_weights(w):
    """stepfun.py:125-144: cw0 = [0, min(1, cumsum(w[:-1])), 1]."""
    cw = torch.clamp(torch.cumsum(w[..., :-1], dim=-1), max=1.0)
    shape = cw.shape[:-1] + (1,)
    return torch.cat([torch.zeros(shape, dtype=w.dtype), cw, torch.ones(shape, dtype=w.dtype)], dim=-1)


def sorted_interp(x, xp, fp):
    """math.py:412-457 (non-TPU branch): searchsorted(side='right') + clamp + lerp."""
    eps = mathx.EPS ** 2
    idx = torch.searchsorted(xp.contiguous(), x.contiguous(), right=True)
    idx1 = torch.clamp(idx, max=xp.shape[-1] - 1)
    idx0 = torch.clamp(idx - 1, min=0)
    xp0 = torch.gather(xp, -1, idx0)
    xp1 = torch.gather(xp, -1, idx1)
    fp0 = torch.gather(fp, -1, idx0)
    fp1 = torch.gather(fp, -1, idx1)
    offset = torch.clamp((x - xp0) / torch.clamp(xp1 - xp0, min=eps), 0, 1)
    return fp0 + offset * (fp1 - fp0)


def sample_u(num_samples, jitter, batch_shape, dtype):
    """The `u` of stepfun.py:186-202 (single_jitter=True, deterministic_center=True).

    jitter: None -> the rng-is-None linspace (stepfun.py:189-191); otherwise a
    [..., 1] tensor of U[0, 1) numbers which is scaled by max_jitter
    (jax.random.uniform(maxval=max_jitter) == u01 * max_jitter).
    """
    eps = mathx.EPS
    if jitter is None:
        pad = 1 / (2 * num_samples)
        u = mathx.linspace(pad, 1.0 - pad - eps, num_samples, dtype)
        return u.expand(batch_shape + (num_samples,))
    u_max = eps + (1 - eps) / num_samples
    max_jitter = (1 - u_max) / (num_samples - 1) - eps
    return mathx.linspace(0.0, 1 - u_max, num_samples, dtype) + jitter.to(dtype) * max_jitter


def sample_intervals(jitter, t, w_logits, num_samples, domain=(0.0, 1.0)):
    """stepfun.py:207-250 (+ sample :158-204, invert_cdf :147-155)."""
    u = sample_u(num_samples, jitter, t.shape[:-1], t.dtype)
    w = torch.softmax(w_logits, dim=-1)
    cw = integrate_weights(w)
    centers = sorted_interp(u, cw, t)
    mid = (centers[..., 1:] + centers[..., :-1]) / 2
    first = 2 * centers[..., :1] - mid[..., :1]
    last = 2 * centers[..., -1:] - mid[..., -1:]
    samples = torch.cat([first, mid, last], dim=-1)
    samples = torch.sort(torch.clamp(samples, domain[0], domain[1]), dim=-1).values
    return samples


def interp(x, xp, fp):
    """jnp.interp (jax 0.4.x) for x:[n], xp,fp:[..., m] (vectorised over leading dims)."""
    m = xp.shape[-1]
    xb = x.expand(xp.shape[:-1] + (x.shape[-1],)).contiguous()
    i = torch.clamp(torch.searchsorted(xp.contiguous(), xb, right=True), 1, m - 1)
    fp_i, fp_im = torch.gather(fp, -1, i), torch.gather(fp, -1, i - 1)
    xp_i, xp_im = torch.gather(xp, -1, i), torch.gather(xp, -1, i - 1)
    df = fp_i - fp_im
    dx = xp_i - xp_im
    delta = xb - xp_im
    epsilon = float(np.spacing(np.finfo(np.float32).eps))
    dx0 = torch.abs(dx) <= epsilon
    f = torch.where(dx0, fp_im, fp_im + (delta / torch.where(dx0, torch.ones_like(dx), dx)) * df)
    f = torch.where(xb < xp[..., :1], fp[..., :1], f)
    f = torch.where(xb > xp[..., -1:], fp[..., -1:], f)
    return f


def weighted_percentile(t, w, ps):
    """stepfun.py:306-314."""
    cw = integrate_weights(w)
    return interp(torch.as_tensor(ps, dtype=t.dtype) / 100, cw, t)
