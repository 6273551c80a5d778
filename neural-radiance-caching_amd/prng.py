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


def light_vmf_noise(shape, seed: int = 1) -> np.ndarray:
    """Constant mean noise of LightMLP.get_vmfs (light_sampler.py:135-144): normal(random_split(PRNGKey(seed))[0],
    vmf_params.shape[:-1] + (3,)); the caller scales it by vmf_scale / 2."""
    key, _ = random_split(PRNGKey(seed))
    return normal(key, tuple(shape))
