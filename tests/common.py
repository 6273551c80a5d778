"""Shared helpers of the test-suite: seeded inputs for both the oracle and the HIP path."""
import functools

import numpy as np
import torch

import nrc_amd
from oracle import cache_ref


@functools.lru_cache(maxsize=4)
def weights_np(density_shift=0.0, seed=1):
    return nrc_amd.synthetic_weights(nrc_amd.hotdog_config(), seed=seed, density_shift=density_shift)


@functools.lru_cache(maxsize=3)
def weights_material_np(smooth=False, seed=1):
    """Cache + material + light weights.  smooth=True: table amplitude 0.2 * 0.5**level (equal spatial
    gradient per level) -- removes the chaotic amplification of one-ulp position differences that random
    fine-level tables cause, so the two implementations can be compared tightly."""
    kw = dict(level_decay=0.5, table_range=0.2) if smooth else {}
    return nrc_amd.synthetic_weights(nrc_amd.hotdog_config(), passes=("cache", "material"), seed=seed, **kw)


def to_torch(w, dtype=None):
    return {k: (torch.from_numpy(v) if dtype is None else torch.from_numpy(v).to(dtype)) for k, v in w.items()}


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


def rays_dict_torch(rays: dict, dtype=torch.float32):
    return {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in rays.items()}


def secondary_case(n, seed=5):
    """Secondary rays as material.get_secondary_rays builds them (render_utils.py:927-1056): origins on
    surfaces inside the scene offset along the normal, unit directions, near 0.05, far 2 (hotdog config),
    plus explicit random inputs (per-level jitter, Gumbel noise)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p = rng.normal(size=(n, 3))
    p = p / np.linalg.norm(p, axis=-1, keepdims=True) * (0.9 * rng.uniform(size=(n, 1)) ** (1 / 3))
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    rays = dict(origins=f32(p + 1e-2 * nrm), directions=f32(d), viewdirs=f32(d), near=f32(np.full((n, 1), 0.05)),
                far=f32(np.full((n, 1), 2.0)), lights=f32(p), normals=f32(nrm), lossmult=f32(np.ones((n, 1))))
    rnd = {"jitter": [rng.uniform(size=(n,)).astype(np.float32) for _ in range(3)],
           "gumbel": rng.gumbel(size=(n, 32)).astype(np.float32)}
    return rays, rnd


@functools.lru_cache(maxsize=3)
def weights_transient_np(smooth=False, density_shift=0.0):
    kw = dict(level_decay=0.5, table_range=0.2) if smooth else {}
    return nrc_amd.synthetic_weights(nrc_amd.cornell_transient_config(), density_shift=density_shift, **kw)


def shadow_jitters(n_shadow, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [rng.uniform(size=(n_shadow, 1)).astype(np.float32) for _ in range(3)]


def oracle_transient(n_rays, jitter_seed=None, seed=20200823, smooth=False, dtype=torch.float32, occlusions=False,
                     shadow_jitter_seed=None, density_shift=0.0):
    """Time-resolved cornell cache on synthetic transient rays (oracle/transient_ref.py)."""
    from oracle import transient_ref
    cfg = nrc_amd.cornell_transient_config(use_occlusions=occlusions)
    rays = nrc_amd.synthetic_transient_rays(n_rays, seed=seed)
    jit = None if jitter_seed is None else [torch.from_numpy(j).to(dtype) for j in jitters(n_rays, seed=jitter_seed)]
    sj = None if shadow_jitter_seed is None else [torch.from_numpy(j).to(dtype) for j in shadow_jitters(n_rays * 32, shadow_jitter_seed)]
    return transient_ref.transient_forward(to_torch(weights_transient_np(smooth, density_shift), dtype), cfg, rays_torch(rays, dtype), jit, sj)


def make_rc(density_shift=0.0):
    """GPU handle with the synthetic hotdog weights loaded (gpu tests only)."""
    from nrc_amd import rc_ext
    rc = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
    rc.load_weights(weights_np(density_shift))
    return rc
