"""Shared helpers of the test-suite: seeded inputs for both the oracle and the HIP path."""
import functools

import numpy as np
import torch

import nrc_amd
from oracle import cache_ref


@functools.lru_cache(maxsize=4)
def weights_np(density_shift=0.0, seed=1):
    return nrc_amd.synthetic_weights(nrc_amd.hotdog_config(), seed=seed, density_shift=density_shift)


def weights_torch(density_shift=0.0, seed=1, dtype=None):
    w = weights_np(density_shift, seed)
    return {k: (torch.from_numpy(v) if dtype is None else torch.from_numpy(v).to(dtype)) for k, v in w.items()}


def rays_torch(rays, dtype=torch.float32):
    return {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in rays.hot_fields().items()}


def jitters(n, levels=3, seed=7):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [rng.uniform(size=(n, 1)).astype(np.float32) for _ in range(levels)]


def oracle_cache(n_rays, dtype=torch.float32, jitter_seed=None, density_shift=0.0, seed=20200823, **kw):
    cfg = nrc_amd.hotdog_config()
    rays = nrc_amd.synthetic_rays(n_rays, seed=seed)
    jit = None if jitter_seed is None else [torch.from_numpy(j) for j in jitters(n_rays, seed=jitter_seed)]
    return cache_ref.material_model_cache_only(weights_torch(density_shift), cfg, rays_torch(rays, dtype), jit, **kw)
