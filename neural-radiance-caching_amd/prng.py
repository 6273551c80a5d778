"""Counter-based random numbers with the stream layout of the reference's `jax.random`.

The reference pins jax==0.4.16 (requirements.txt:2), whose default PRNG is threefry2x32 with raw uint32[2] keys
and the *non-partitionable* counter layout.  jax is not part of /root/reference, so this module restates the
published algorithm (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11: Threefry-2x32, 20 rounds)
and jax 0.4.16's conventions on top of it:

  PRNGKey(seed)          key = [seed >> 32, seed & 0xffffffff]
  threefry over counts   counts.ravel() is cut in two halves; element i of the first half and element i of the
                         second half are the two words of block i (an odd count is padded with one 0 and the last
                         output dropped); the result is concat(out_word0, out_word1)
  split(key, n)          counts = iota(2 n), result reshaped [n, 2]
  fold_in(key, d)        one block with counter PRNGKey(d)
  random_bits(key, shp)  counts = iota(prod(shp))
  uniform                mantissa trick: (bits >> 9 | 0x3f800000) as float - 1, then *(max-min)+min, max(min, .)
  normal                 sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1)), erfinv = XLA's single-precision
                         polynomial (Giles, "Approximating the erfinv function")
  gumbel / categorical   -log(-log(uniform(tiny, 1))); argmax(gumbel + logits)

Pinned by the published known-answer values of Random123 and of the jax documentation (tests/golden/prng_kat.json,
tests/test_prng.py).  What CANNOT be checked in this container is the reference's *call order* end to end (no jax
here): `cache_pass_randoms` follows the `utils.random_split` sites cited in its docstring, read from the source.

Bits, keys and uniforms are integer / exactly-rounded float arithmetic and therefore bit-exact with jax; `normal` and
`gumbel` go through log/log1p/sqrt, which differ between XLA back ends by an ulp as well.

The device twin is rc_prng_fill (csrc/rc_prng.hip, include/rc_abi.h): same counter layout, filled in HBM.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_PARITY = np.uint32(0x1BD11BDA)

MODE_BITS, MODE_UNIFORM, MODE_NORMAL, MODE_GUMBEL = 0, 1, 2, 3


def _rotl(x, r):
    return (x << np.uint32(r)) | (x >> np.uint32(32 - r))


def threefry2x32(key, x0, x1) -> Tuple[np.ndarray, np.ndarray]:
    """Threefry-2x32, 20 rounds, on arrays of counter words (x0, x1) under one key (k0, k1)."""
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    ks = (k0, k1, np.uint32(k0 ^ k1 ^ _PARITY))
    x0 = np.array(x0, dtype=np.uint32, copy=True)
    x1 = np.array(x1, dtype=np.uint32, copy=True)
    with np.errstate(over="ignore"):
        x0 += ks[0]
        x1 += ks[1]
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 += x1
                x1 = _rotl(x1, r) ^ x0
            x0 += ks[(i + 1) % 3]
            x1 += ks[(i + 2) % 3] + np.uint32(i + 1)
    return x0, x1


def _threefry_counts(key, counts) -> np.ndarray:
    c = np.asarray(counts, dtype=np.uint32).ravel()
    odd = c.size & 1
    if odd:
        c = np.concatenate([c, np.zeros(1, np.uint32)])
    half = c.size // 2
    a, b = threefry2x32(key, c[:half], c[half:])
    out = np.concatenate([a, b])
    return out[:-1] if odd else out


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def as_key(key) -> np.ndarray:
    k = np.asarray(key)
    if k.shape != (2,) or k.dtype != np.uint32:
        raise ValueError("a PRNG key is a uint32 array of shape (2,)")
    return k


def is_key(x) -> bool:
    return isinstance(x, np.ndarray) and x.shape == (2,) and x.dtype == np.uint32


def split(key, num: int = 2) -> np.ndarray:
    return _threefry_counts(as_key(key), np.arange(2 * num, dtype=np.uint32)).reshape(num, 2)


def fold_in(key, data: int) -> np.ndarray:
    return _threefry_counts(as_key(key), PRNGKey(data))


def random_split(rng):
    """internal/utils.py:118-123."""
    if rng is None:
        return None, None
    k = split(rng)
    return k[0], k[1]


def random_bits(key, shape) -> np.ndarray:
    shape = tuple(int(s) for s in np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape
    n = int(np.prod(shape)) if len(shape) else 1
    if n >= 2 ** 32 - 1:
        raise ValueError("more than 2^32 - 2 values per call are not supported")
    return _threefry_counts(as_key(key), np.arange(n, dtype=np.uint32)).reshape(shape)


def uniform(key, shape=(), minval=0.0, maxval=1.0) -> np.ndarray:
    bits = random_bits(key, tuple(shape))
    f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, (f * (hi - lo)).astype(np.float32) + lo).astype(np.float32)


_ERFINV_CENTRAL = (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087,
                   -0.00125372503, -0.00417768164, 0.246640727, 1.50140941)
_ERFINV_TAIL = (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773,
                -0.0076224613, 0.00943887047, 1.00167406, 2.83297682)


def erfinv32(x) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    w = (-np.log1p(-x * x)).astype(np.float32)
    central = w < np.float32(5.0)
    with np.errstate(invalid="ignore"):
        ww = np.where(central, w - np.float32(2.5), np.sqrt(w) - np.float32(3.0)).astype(np.float32)
    p = np.where(central, np.float32(_ERFINV_CENTRAL[0]), np.float32(_ERFINV_TAIL[0])).astype(np.float32)
    for a, b in zip(_ERFINV_CENTRAL[1:], _ERFINV_TAIL[1:]):
        p = (np.where(central, np.float32(a), np.float32(b)) + p * ww).astype(np.float32)
    return (p * x).astype(np.float32)


def normal(key, shape=()) -> np.ndarray:
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, shape, lo, 1.0)
    return (np.float32(np.sqrt(2.0)) * erfinv32(u)).astype(np.float32)


def gumbel(key, shape=()) -> np.ndarray:
    u = uniform(key, shape, np.finfo(np.float32).tiny, 1.0)
    return (-np.log(-np.log(u))).astype(np.float32)


def categorical(key, logits, axis: int = -1, shape: Optional[Sequence[int]] = None) -> np.ndarray:
    """jax.random.categorical of jax 0.4.16: argmax over `axis` of gumbel noise + logits, the noise drawn with
    `logits.shape[axis]` re-inserted at `axis` of the requested batch shape."""
    logits = np.asarray(logits, dtype=np.float32)
    axis = axis % logits.ndim
    batch = tuple(np.delete(logits.shape, axis))
    shape = batch if shape is None else tuple(shape)
    prefix = shape[: len(shape) - len(batch)]
    lshape = list(shape[len(shape) - len(batch):])
    lshape.insert(axis, logits.shape[axis])
    g = gumbel(key, prefix + tuple(lshape))
    return np.argmax(g + logits.reshape((1,) * len(prefix) + logits.shape), axis=axis + len(prefix))


# ---------------------------------------------------------------------------------------------------------
# the reference's call order on the cache path
# ---------------------------------------------------------------------------------------------------------

def max_jitter(num_samples: int, eps: float = float(np.finfo(np.float32).eps)) -> float:
    """stepfun.py:197-198."""
    u_max = eps + (1 - eps) / num_samples
    return (1 - u_max) / (num_samples - 1) - eps


def sampler_randoms(sampler_rng, n_rays: int, num_samples: Sequence[int]):
    """Per-level jitter of ProposalVolumeSampler.__call__ given the rng it is called with (sampling.py:341-346:
    one random_split for sample_intervals, :408-409 one more for the density MLP, per level).  single_jitter=True:
    one uniform [n_rays, 1] per level (stepfun.py:196-202).  Returned as the UNIT uniform the C ABI takes
    (rc_randoms.jitter): jax's uniform(maxval=max_jitter) is float32(unit * max_jitter), the product the sampling
    kernel forms itself."""
    rng = as_key(sampler_rng)
    out = []
    for _ in num_samples:
        key, rng = random_split(rng)
        out.append(uniform(key, (n_rays, 1)))
        _, rng = random_split(rng)
    return out


def cache_keys(cache_rng) -> Dict[str, np.ndarray]:
    """Keys of the radiance cache's __call__ given its rng (models.py:710-712 sampler, :727 resample, :748 shader;
    the use_slf / env_map_only branches are not taken on this path)."""
    k_sampler, rng = random_split(as_key(cache_rng))
    k_resample, rng = random_split(rng)
    k_shader, rng = random_split(rng)
    return {"sampler": k_sampler, "resample": k_resample, "shader": k_shader}


def model_cache_rng(model_rng) -> np.ndarray:
    """rng the cache is called with, from the rng of BaseMaterialModel.__call__ (models.py:1156 bypass split,
    :1176 cache-pass split, :1375 split inside _handle_cache_pass)."""
    _, rng = random_split(as_key(model_rng))
    k_pass, rng = random_split(rng)
    k_cache, _ = random_split(k_pass)
    return k_cache


def cache_pass_randoms(model_rng, n_rays: int, num_samples: Sequence[int], resample: bool = False):
    """Explicit random tensors of one primary cache pass (the dict the C ABI's rc_randoms takes) derived from the
    model's rng like the reference would.  Gumbel noise [n_rays, S] of the num_resample = 1 categorical draw
    (models.py:240-247: one more random_split inside maybe_resample) only when `resample`."""
    ck = cache_keys(model_cache_rng(model_rng))
    rnd = {"jitter": sampler_randoms(ck["sampler"], n_rays, num_samples)}
    if resample:
        key, _ = random_split(ck["resample"])
        rnd["gumbel"] = gumbel(key, (n_rays, int(num_samples[-1]), 1))[..., 0]
    return rnd


def pixel_jitter(rng, shape, jitter: int = 1, jitter_scale: float = 1.0):
    """The sub-pixel offsets camera_utils.pixels_to_rays draws when jitter > 0 (internal/camera_utils.py:943-957):
    key, rng = split(rng); k1, k2 = split(key); jitter == 1: U(0, 1) - 0.5 each, otherwise N(0, 1) * 0.5; with
    jitter_scale > 1 a second uniform pair from split(k1) is added.  Returns (dx, dy) float32 arrays of `shape` --
    what rc_camera.pix_dx / pix_dy (cast_ray_batch(..., pix_jitter=...)) take.  As with the other key paths of this
    module the call ORDER is restated from the source and cannot be checked against a JAX run here."""
    key, _ = split(as_key(rng))
    k1, k2 = split(key)
    if jitter == 1:
        dx = uniform(k1, shape) - np.float32(0.5)
        dy = uniform(k2, shape) - np.float32(0.5)
    else:
        dx = normal(k1, shape) * np.float32(0.5)
        dy = normal(k2, shape) * np.float32(0.5)
    if jitter_scale > 1.0:
        k1, k2 = split(k1)
        dx = dx + (uniform(k1, shape) - np.float32(0.5))
        dy = dy + (uniform(k2, shape) - np.float32(0.5))
    return dx.astype(np.float32), dy.astype(np.float32)


def light_vmf_noise(shape, seed: int = 1) -> np.ndarray:
    """Constant mean noise of LightMLP.get_vmfs (light_sampler.py:135-144): normal(random_split(PRNGKey(seed))[0],
    vmf_params.shape[:-1] + (3,)); the caller scales it by vmf_scale / 2."""
    key, _ = random_split(PRNGKey(seed))
    return normal(key, tuple(shape))


def material_pass_randoms(model_rng, n_rays: int, cfg) -> Dict[str, object]:
    """Explicit random tensors of ONE material-stage forward (the dict rc_material_randoms / Model.apply(passes=
    ("cache", "light", "material")) take) derived from the model's rng at the reference's split sites:

      BaseMaterialModel.__call__            models.py:1156 (bypass), 1177 (cache pass), 1195 (_get_material_samples),
                                            1208 (light sampler), 1220 (_handle_material_pass)
      _get_material_samples                 models.py:1417 (first maybe_resample, no draw), 1431 (MaterialModel's resample)
        maybe_resample                      models.py:241 -> jax.random.categorical == argmax(logits + gumbel[R, S, 1])
      _handle_material_pass -> shader       models.py:1548; shading.py:289; material.py:1971, 1988, 2030, 2535
      get_outgoing_radiance                 material.py:1383 (indirect specular), 1416 (indirect diffuse); 1642, 1721, 1780
      get_secondary_rays / importance_...   render_utils.py:962, 767 + 324 (uh, uw = uniform(key, [N K, 2])), 784
      LightSampler.sample_directions        render_utils.py:1450, 1407, 1358 (lobe: categorical over 128 logits),
                                            1409 (normal [N, K, 2]), 1413 (uniform [N, K])
      secondary cache calls                 material.py:2190; models.py:712 / 727 (resample gumbel [R, K, S, 1]);
                                            their sampler runs on the constant PRNGKey(0) at render time
                                            (sampling.py:170-179) -- the specular and the diffuse trace therefore draw the
                                            SAME per-level jitter
    The call order is restated from the source and cannot be checked against a jax run here (no jax in the image)."""
    K = int(cfg.num_secondary_samples)
    Kd = int(round(K * cfg.diffuse_sample_fraction))
    Ks = int(round(K * (1.0 - cfg.diffuse_sample_fraction)))
    Kc = int(round(0.5 * Kd))
    Kl = Kd - Kc
    levels = [lvl[2] for lvl in cfg.sampling_strategy]
    S = int(levels[-1])
    rng = as_key(model_rng)
    out = dict(cache_pass_randoms(rng, n_rays, levels, resample=False))           # primary rays' jitter
    _, rng = random_split(rng)                # :1156
    _, rng = random_split(rng)                # :1177
    k_samples, rng = random_split(rng)        # :1195
    _, rng = random_split(rng)                # :1208 (LightMLP draws nothing from it: its noise is the PRNGKey(1) constant)
    k_mat, rng = random_split(rng)            # :1220
    # -- categorical pick of the shading sample
    _, r = random_split(k_samples)            # :1417
    k2, r = random_split(r)                   # :1431
    kg, _ = random_split(k2)                  # :241
    out["gumbel"] = gumbel(kg, (n_rays, S, 1))[..., 0]
    out["vmf_noise"] = light_vmf_noise((n_rays, 1, int(cfg.num_vmf), 3))[:, 0]
    # -- material shader
    k_sh, _ = random_split(k_mat)             # models.py:1548
    k_pa, _ = random_split(k_sh)              # shading.py:289
    _, r = random_split(k_pa)                 # material.py:1971
    _, r = random_split(r)                    # :1988
    k_int, _ = random_split(r)                # :2030
    k_out, _ = random_split(k_int)            # :2535
    k_spec, r = random_split(k_out)           # :1383 indirect specular
    k_diff, r = random_split(r)               # :1416 indirect diffuse (the env-map passes reuse these rays)

    def one_pass(k_pass, counts):
        """counts: samples per sampler, in sampler order.  Returns ([(uh, uw, sampler_rng)], key of the cache call)."""
        kh, _ = random_split(k_pass)          # :1642
        kh1, r2 = random_split(kh)            # :1721 get_secondary_rays
        kh2, _ = random_split(r2)             # :1780 radiance_cache_fn
        ki, _ = random_split(kh1)             # render_utils.py:962
        draws, r3 = [], ki
        for cnt in counts:
            ka, r3 = random_split(r3)         # :767
            ku, _ = random_split(ka)          # :324
            u = uniform(ku, (n_rays * cnt, 2))
            kb, r3 = random_split(r3)         # :784
            draws.append((u[:, 0].reshape(n_rays, cnt), u[:, 1].reshape(n_rays, cnt), kb))
        kc, _ = random_split(kh2)             # material.py:2190
        return draws, kc

    def trace_randoms(kc, Kp):
        ck = cache_keys(kc)                   # models.py:712 (sampler: replaced by PRNGKey(0)), 727, 748
        kg2, _ = random_split(ck["resample"])  # :241
        jit = sampler_randoms(PRNGKey(0), n_rays * Kp, levels)
        return jit, gumbel(kg2, (n_rays, Kp, S, 1))[..., 0].reshape(n_rays * Kp, S)

    (su1, su2, _), = one_pass(k_spec, [Ks])[0]
    _, kc_spec = one_pass(k_spec, [Ks])
    out["spec_u1"], out["spec_u2"] = su1, su2
    out["spec_jitter"], out["spec_gumbel"] = trace_randoms(kc_spec, Ks)
    draws, kc_diff = one_pass(k_diff, [Kc, Kl])
    out["cos_u1"], out["cos_u2"] = draws[0][0], draws[0][1]
    k_light = draws[1][2]
    ks, _ = random_split(k_light)             # LightSampler.sample_directions (render_utils.py:1450)
    kv1, r5 = random_split(ks)                # sample_vmf :1407
    kl, _ = random_split(kv1)                 # sample_vmf_vars :1358
    out["vmf_lobe_gumbel"] = gumbel(kl, (n_rays, int(cfg.num_vmf)))
    kn, r5 = random_split(r5)                 # :1409
    out["vmf_v"] = normal(kn, (n_rays, Kl, 2))
    kt, _ = random_split(r5)                  # :1413
    out["vmf_tmp"] = uniform(kt, (n_rays, Kl))
    out["diff_jitter"], out["diff_gumbel"] = trace_randoms(kc_diff, Kd)
    return out
