"""Instant-NGP style multiresolution dense+hash encoding (oracle).

Follows internal/grid_utils.py:41-121 (hash trilerp), :352-445 (dense trilerp
with zero padding), :679-726 (trilerp dispatch / axis flip) and :808-905
(HashEncoding.__call__).  See oracle/__init__.py for the usage rules.
"""
from __future__ import annotations

import numpy as np
import torch

PI_2 = 19349663    # grid_utils.py:102
PI_3 = 83492791    # grid_utils.py:103


def hash_index(px, py, pz, table_size):
    """grid_utils.py:99-111.  px,py,pz: int64 tensors holding int32 corner coords.

    int32 -> uint32 wraparound, uint32 multiply (mod 2^32), xor, mod T.
    """
    m = 0xFFFFFFFF
    ux, uy, uz = px & m, py & m, pz & m
    h = ux ^ ((uy * PI_2) & m) ^ ((uz * PI_3) & m)
    return h % table_size


def _corner_weights(loc):
    """Corner offsets/weights in the reference order (grid_utils.py:68-89, 399-420):
    corner c = (b0, b1, b2) over loc[...,0..2] with b2 fastest."""
    floored = torch.floor(loc)
    ceil_w = loc - floored
    floor_w = 1.0 - ceil_w
    out = []
    for b0 in (0, 1):
        for b1 in (0, 1):
            for b2 in (0, 1):
                w = ((ceil_w if b0 else floor_w)[..., 0]
                     * (ceil_w if b1 else floor_w)[..., 1]
                     * (ceil_w if b2 else floor_w)[..., 2])
                out.append(((b0, b1, b2), w))
    return floored, out


def hash_resample_3d(table, coords):
    """grid_utils.py:41-121 (TRILINEAR, half_pixel_center=True).

    table: [T, F]; coords: [N, 3] = x01 * grid_size, (x, y, z) order.
    """
    loc = coords - 0.5
    floored, cw = _corner_weights(loc)
    fl = floored.to(torch.int64)
    out = None
    for (b0, b1, b2), w in cw:
        idx = hash_index(fl[..., 0] + b0, fl[..., 1] + b1, fl[..., 2] + b2, table.shape[0])
        g = table[idx] * w[..., None]
        out = g if out is None else out + g
    return out


def dense_resample_3d(grid, coords):
    """trilerp 'grid' branch: grid_utils.py:711-715 + jax_resample_3d :352-445.

    grid: [N, N, N, F] indexed [x, y, z]; coords: [P, 3] = x01 * N in (x, y, z) order.
    The reference flips to (z, y, x), shifts by -0.5, zero-pads by one voxel (+1),
    clamps corner indices to the padded volume and gathers data[loc2, loc1, loc0].
    """
    n = grid.shape[0]
    loc = torch.flip(coords - 0.5, dims=[-1]) + 1.0      # (z, y, x) + pad shift
    floored, cw = _corner_weights(loc)
    fl = floored.to(torch.int64)
    padded = torch.nn.functional.pad(grid, (0, 0, 1, 1, 1, 1, 1, 1))
    out = torch.zeros(coords.shape[:-1] + (grid.shape[-1],), dtype=grid.dtype)
    for (b0, b1, b2), w in cw:
        i0 = torch.clamp(fl[..., 0] + b0, 0, n + 1)    # z
        i1 = torch.clamp(fl[..., 1] + b1, 0, n + 1)    # y
        i2 = torch.clamp(fl[..., 2] + b2, 0, n + 1)    # x
        out = out + padded[i2, i1, i0] * w[..., None]
    return out


def hash_encoding(weights, prefix, gcfg, x):
    """HashEncoding.__call__ (grid_utils.py:808-905) with x_scale=None,
    per_level_fn=mean over one control point, feature_aggregator='concatenate'.

    weights: dict name -> tensor; prefix: e.g. 'params/Cache/Sampler/MLP_0/density_grid';
    x: [..., 3] already warped (contracted) coordinates.
    """
    lo, hi = -gcfg.bbox, gcfg.bbox
    shp = x.shape[:-1]
    x01 = ((x - lo) / (hi - lo)).reshape(-1, 3)
    feats = []
    for n in gcfg.grid_sizes:
        v = weights[f"{prefix}/{gcfg.level_name(n)}"].to(x.dtype)
        if gcfg.is_dense(n):
            f = dense_resample_3d(v, x01 * n)
        else:
            f = hash_resample_3d(v, x01 * n)
        feats.append(f)
    out = torch.cat(feats, dim=-1) * gcfg.precondition_scaling
    return out.reshape(shp + (out.shape[-1],))
