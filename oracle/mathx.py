"""Scalar/vector numerics of the hot path (oracle; see oracle/__init__.py).

Follows internal/math.py, internal/coord.py, internal/ref_utils.py of the
reference.  All functions are dtype-generic torch (float64 spec / float32
baseline).  The float32 constants tiny/min/max/eps are the *float32* ones in
both precisions because the reference hard-codes them (math.py:24-26).
"""
from __future__ import annotations

import math as pymath

import numpy as np
import torch

F32 = np.finfo(np.float32)
TINY = float(F32.tiny)   # math.py:24
MINV = float(F32.min)    # math.py:25
MAXV = float(F32.max)    # math.py:26
EPS = float(F32.eps)


def safe_log(x):
    """math.py:177-183: log(clip(x, tiny, max))."""
    return torch.log(torch.clamp(x, TINY, MAXV))


def safe_exp(x):
    """math.py:186-192: exp(clip(x, min, 70))."""
    return torch.exp(torch.clamp(x, MINV, 70.0))


def softplus(x):
    """jax.nn.softplus == logaddexp(x, 0)."""
    return torch.logaddexp(x, torch.zeros((), dtype=x.dtype))


def sigmoid(x):
    return torch.sigmoid(x)


def linspace(start, stop, num, dtype):
    """jnp.linspace(endpoint=True): start*(1-step)+stop*step, last := stop."""
    div = num - 1
    step = torch.arange(div, dtype=dtype) / div
    start_t = torch.as_tensor(start, dtype=dtype)
    stop_t = torch.as_tensor(stop, dtype=dtype)
    out = start_t * (1 - step) + stop_t * step
    return torch.cat([out, stop_t.reshape(1)])


def dot(a, b):
    """math.py:486-488 (keepdims=True)."""
    return (a * b).sum(-1, keepdim=True)


def l2_normalize(x):
    """ref_utils.py:45-72 forward value."""
    denom_sq = (x * x).sum(-1, keepdim=True)
    val = x / torch.sqrt(torch.clamp(denom_sq, min=TINY))
    return torch.where(denom_sq < TINY, torch.zeros_like(val), val)


def nan_to_num(x):
    """jnp.nan_to_num defaults: nan->0, +-inf->+-finfo.max of the dtype."""
    fi = torch.finfo(x.dtype)
    return torch.nan_to_num(x, nan=0.0, posinf=fi.max, neginf=fi.min)


def reflect(viewdirs, normals):
    """ref_utils.py:25-42: 2 (n.v) n - v."""
    return 2.0 * (normals * viewdirs).sum(-1, keepdim=True) * normals - viewdirs


def contract(x):
    """coord.py:63-69."""
    mag_sq = torch.clamp((x * x).sum(-1, keepdim=True), min=1.0)
    scale = (2 * torch.sqrt(mag_sq) - 1) / mag_sq
    return scale * x


def contract_radius(x, c):
    """coord.py:33-38 (contract_radius_2 / contract_radius_5): contract(x / c)."""
    return contract(x / c)


def power_ladder(x, p, premult):
    """math.py:295-316 for finite p not in {0, 1}."""
    x = x * premult
    xp = torch.abs(x)
    xs = xp / max(TINY, abs(p - 1))
    y = abs(p - 1) / p * ((xs + 1) ** p - 1)
    y = torch.clamp(y, MINV, MAXV)
    return torch.where(x < 0, -y, y)


def inv_power_ladder(y, p, premult):
    """math.py:319-341 for finite p not in {0, 1}."""
    yp = torch.abs(y)
    # power_ladder_max_output(p) = (p-1)/p for p < 0 (math.py:284-292); minus_eps.
    if p < 0:
        y_max = float(np.nextafter(np.float32((p - 1) / p), np.float32(-np.inf)))
        yp = torch.clamp(yp, -y_max, y_max)
    pm1 = abs(p - 1)
    x = pm1 * ((p / pm1 * yp + 1) ** (1 / p) - 1)
    x = torch.where(y < 0, -x, x)
    return x / premult


def pos_enc(x, min_deg, max_deg, append_identity=True):
    """coord.py:298-312."""
    scales = 2.0 ** torch.arange(min_deg, max_deg, dtype=x.dtype)
    scaled = (x[..., None, :] * scales[:, None]).reshape(x.shape[:-1] + (-1,))
    four = torch.sin(torch.cat([scaled, scaled + 0.5 * pymath.pi], dim=-1))
    return torch.cat([x, four], dim=-1) if append_identity else four


# ----------------------------------------------------------------------------
# Integrated directional encoding (ref_utils.py:92-192)
# ----------------------------------------------------------------------------
def _gen_binom(a, k):
    return np.prod(a - np.arange(k)) / pymath.factorial(k)


def _assoc_legendre_coeff(l, m, k):
    return ((-1) ** m * 2 ** l * pymath.factorial(l) / pymath.factorial(k)
            / pymath.factorial(l - k - m) * _gen_binom(0.5 * (l + k + m - 1.0), l))


def _sph_harm_coeff(l, m, k):
    return np.sqrt((2.0 * l + 1.0) * pymath.factorial(l - m)
                   / (4.0 * np.pi * pymath.factorial(l + m))) * _assoc_legendre_coeff(l, m, k)


def ide_tables(deg_view):
    """(ml_array [2, n], mat [l_max+1, n]) of ref_utils.py:118-153."""
    ml = []
    for i in range(deg_view):
        l = 2 ** i
        for m in range(l + 1):
            ml.append((m, l))
    ml = np.array(ml).T
    l_max = 2 ** (deg_view - 1)
    mat = np.zeros((l_max + 1, ml.shape[1]))
    for i, (m, l) in enumerate(ml.T):
        for k in range(l - m + 1):
            mat[k, i] = _sph_harm_coeff(l, m, k)
    return ml, mat


def ide(xyz, kappa_inv, deg_view):
    """ref_utils.py:155-190.  Returns [..., 2 * n_terms] (real parts, then imaginary)."""
    ml, mat = ide_tables(deg_view)
    dt = xyz.dtype
    x, y, z = xyz[..., 0:1], xyz[..., 1:2], xyz[..., 2:3]
    vmz = torch.cat([z ** i for i in range(mat.shape[0])], dim=-1)
    cdt = torch.complex128 if dt == torch.float64 else torch.complex64
    cxy = torch.complex(x, y).to(cdt)
    vmxy = torch.cat([cxy ** int(m) for m in ml[0, :]], dim=-1)
    poly = vmz @ torch.as_tensor(mat, dtype=dt)
    sph = vmxy * poly.to(cdt)
    sigma = torch.as_tensor(0.5 * ml[1, :] * (ml[1, :] + 1), dtype=dt)
    att = torch.exp(-sigma * kappa_inv)
    out = sph * att.to(cdt)
    return torch.cat([out.real, out.imag], dim=-1)
