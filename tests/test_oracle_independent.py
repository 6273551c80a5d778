"""The oracle against INDEPENDENT implementations (numpy / scipy), where one exists.

The reference cannot run here and ships no fixtures (parity unpinned, oracle/__init__.py); these tests at least pin
the oracle's building blocks to library code that shares no line with it: scipy's spherical harmonics for the
integrated directional encoding, scipy.ndimage.map_coordinates for the zero-padded dense trilinear lookup, numpy.interp
/ numpy.searchsorted for the step-function resampling and the percentiles, a literal triple loop for the hashed lookup.
"""
import numpy as np
import pytest
import torch

import nrc_amd
from oracle import hashgrid_ref, mathx, stepfun_ref

F64 = torch.float64


def test_ide_all_36_terms_equal_scipy_spherical_harmonics():
    """ref_utils.generate_ide_fn at kappa_inv = 0 is Y_l^m(direction) for l in {1, 2, 4, 8, 16}, m = 0..l
    (Condon-Shortley phase, complex); scipy.special computes the same functions from recurrences."""
    from scipy import special
    rng = np.random.default_rng(0)
    v = rng.normal(size=(64, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    enc = mathx.ide(torch.from_numpy(v), torch.zeros(64, 1, dtype=F64), 5).numpy()
    re, im = enc[:, :36], enc[:, 36:]
    polar = np.arccos(np.clip(v[:, 2], -1, 1))
    azim = np.arctan2(v[:, 1], v[:, 0])
    i = 0
    for l in (1, 2, 4, 8, 16):
        for m in range(l + 1):
            y = special.sph_harm_y(l, m, polar, azim) if hasattr(special, "sph_harm_y") else special.sph_harm(m, l, azim, polar)
            assert np.abs(re[:, i] - y.real).max() <= 2e-10 and np.abs(im[:, i] - y.imag).max() <= 2e-10, (l, m)
            i += 1
    assert i == 36
    # attenuation of every term: exp(-l (l + 1) / 2 * kappa_inv)
    enc_r = mathx.ide(torch.from_numpy(v), torch.full((64, 1), 0.3, dtype=F64), 5).numpy()
    sig = np.array([0.5 * l * (l + 1) for l in (1, 2, 4, 8, 16) for _ in range(l + 1)])
    assert np.abs(enc_r[:, :36] - re * np.exp(-0.3 * sig)).max() <= 1e-12


def test_dense_level_equals_scipy_map_coordinates_on_the_padded_volume():
    """jax_resample_3d (grid_utils.py:352-445): trilinear lookup in the volume zero-padded by one voxel, sample
    points at pixel centres (coords - 0.5), corners clamped to the padded volume."""
    from scipy import ndimage
    rng = np.random.default_rng(1)
    n = 6
    grid = rng.normal(size=(n, n, n, 2))
    coords = rng.uniform(-1.0, n + 1.0, size=(500, 3))             # also outside: everything beyond the pad is zero
    got = hashgrid_ref.dense_resample_3d(torch.from_numpy(grid), torch.from_numpy(coords)).numpy()
    padded = np.pad(grid, ((1, 1), (1, 1), (1, 1), (0, 0)))
    loc = coords - 0.5 + 1.0                                        # (x, y, z) in padded index space, grid is [x, y, z]
    for f in range(2):
        ref = ndimage.map_coordinates(padded[..., f], loc.T, order=1, mode="nearest", prefilter=False)
        assert np.abs(got[:, f] - ref).max() <= 1e-12


def test_hashed_level_equals_literal_loop():
    """jax_hash_resample_3d (grid_utils.py:41-121): 8 corners, int32 -> uint32 hash x ^ y * 19349663 ^ z * 83492791
    modulo the table size, trilinear weights."""
    rng = np.random.default_rng(2)
    T = 1 << 12
    table = rng.normal(size=(T, 3))
    coords = rng.uniform(-3.0, 40.0, size=(200, 3))
    got = hashgrid_ref.hash_resample_3d(torch.from_numpy(table), torch.from_numpy(coords)).numpy()
    for p in range(200):
        loc = coords[p] - 0.5
        fl = np.floor(loc).astype(np.int64)
        w1 = loc - fl
        acc = np.zeros(3)
        for bx in (0, 1):
            for by in (0, 1):
                for bz in (0, 1):
                    ix, iy, iz = int(fl[0]) + bx, int(fl[1]) + by, int(fl[2]) + bz
                    h = (ix & 0xFFFFFFFF) ^ ((iy * 19349663) & 0xFFFFFFFF) ^ ((iz * 83492791) & 0xFFFFFFFF)
                    w = (w1[0] if bx else 1 - w1[0]) * (w1[1] if by else 1 - w1[1]) * (w1[2] if bz else 1 - w1[2])
                    acc += w * table[h % T]
        assert np.abs(got[p] - acc).max() <= 1e-12


def test_interp_and_percentiles_equal_numpy():
    rng = np.random.default_rng(3)
    xp = np.sort(rng.uniform(size=(5, 33)), axis=-1)
    fp = rng.normal(size=(5, 33))
    x = rng.uniform(-0.1, 1.1, size=(17,))
    got = stepfun_ref.interp(torch.from_numpy(x), torch.from_numpy(xp), torch.from_numpy(fp)).numpy()
    for r in range(5):
        assert np.abs(got[r] - np.interp(x, xp[r], fp[r])).max() <= 1e-12
    # weighted_percentile = interp(ps / 100, integrate_weights(w), t)
    t = np.sort(rng.uniform(2, 6, size=(4, 33)), axis=-1)
    w = rng.uniform(size=(4, 32)); w /= w.sum(-1, keepdims=True)
    pct = stepfun_ref.weighted_percentile(torch.from_numpy(t), torch.from_numpy(w), (5.0, 50.0, 95.0)).numpy()
    for r in range(4):
        cw = np.concatenate([[0.0], np.minimum(1.0, np.cumsum(w[r][:-1])), [1.0]])
        assert np.abs(pct[r] - np.interp(np.array([0.05, 0.5, 0.95]), cw, t[r])).max() <= 1e-12


def test_sample_intervals_equals_numpy_inverse_cdf():
    """stepfun.sample (stepfun.py:147-250), deterministic branch: u = linspace(pad, 1 - pad - eps, n) pushed through the
    piecewise-linear inverse CDF (numpy.interp on (cw, t)), then midpoints with reflected ends, clipped and sorted."""
    rng = np.random.default_rng(4)
    P, S = 24, 16
    t = np.sort(rng.uniform(size=(3, P + 1)), axis=-1); t[:, 0], t[:, -1] = 0.0, 1.0
    lg = rng.normal(size=(3, P)) * 2
    got = stepfun_ref.sample_intervals(None, torch.from_numpy(t), torch.from_numpy(lg), S).numpy()
    eps = np.finfo(np.float32).eps
    pad = 1 / (2 * S)
    u = np.linspace(pad, 1 - pad - eps, S)
    for r in range(3):
        w = np.exp(lg[r] - lg[r].max()); w /= w.sum()
        cw = np.concatenate([[0.0], np.minimum(1.0, np.cumsum(w[:-1])), [1.0]])
        c = np.interp(u, cw, t[r])
        mid = (c[1:] + c[:-1]) / 2
        want = np.sort(np.clip(np.concatenate([[2 * c[0] - mid[0]], mid, [2 * c[-1] - mid[-1]]]), 0.0, 1.0))
        # the oracle keeps jnp.linspace's float32 end points (pad, 1 - pad - eps are rounded to float32 as in the
        # reference) and its start (1 - s) + stop s form: within one float32 ulp of numpy's double linspace
        assert np.abs(got[r] - want).max() <= 2e-7


# ---------------------------------------------------------------------------------------------
# material stage: samplers and densities
# ---------------------------------------------------------------------------------------------
def test_vmf_density_equals_scipy_and_sampler_inverts_its_cdf():
    from scipy import stats
    from oracle import material_ref as mr
    rng = np.random.default_rng(5)
    mu = rng.normal(size=3); mu /= np.linalg.norm(mu)
    x = rng.normal(size=(200, 3)); x /= np.linalg.norm(x, axis=-1, keepdims=True)
    for kappa in (0.5, 5.0, 50.0):
        got = mr.eval_vmf(torch.from_numpy(x), torch.from_numpy(mu)[None], torch.tensor(kappa, dtype=F64)).numpy()
        ref = stats.vonmises_fisher(mu, kappa).pdf(x)
        assert np.abs(got / ref - 1).max() <= 1e-10, kappa
    # sample_vmf: w = cos(angle to the mean) = 1 + log(u + (1 - u) exp(-2 kappa)) / kappa is the inverse of the
    # vMF marginal CDF F(w) = (exp(kappa w) - exp(-kappa)) / (exp(kappa) - exp(-kappa))
    n = 64
    vm = {"vmf_means": torch.from_numpy(np.tile(mu, (n, 4, 1))), "vmf_kappas": torch.full((n, 4, 1), 7.0, dtype=F64),
          "vmf_logits": torch.zeros(n, 4, 1, dtype=F64)}
    u = rng.uniform(size=(n, 3))
    dirs, pdf = mr.light_sample(vm, torch.zeros(n, dtype=torch.int64), torch.from_numpy(rng.normal(size=(n, 3, 2))), torch.from_numpy(u))
    w = (dirs.numpy() * mu).sum(-1)
    cdf = (np.exp(7.0 * w) - np.exp(-7.0)) / (np.exp(7.0) - np.exp(-7.0))
    assert np.abs(cdf - u).max() <= 1e-9 and np.abs(np.linalg.norm(dirs.numpy(), axis=-1) - 1).max() <= 1e-12
    assert np.abs(pdf.numpy() / stats.vonmises_fisher(mu, 7.0).pdf(dirs.numpy().reshape(-1, 3)).reshape(n, 3) - 1).max() <= 1e-9


def test_ggx_normal_distribution_is_normalised_and_sampled_by_inverse_cdf():
    from scipy import integrate
    from oracle import material_ref as mr
    for a in (0.05, 0.3, 0.9):
        # D(h) cos(theta_h) integrates to 1 over the hemisphere
        f = lambda th: float(mr.ggx_d(torch.tensor(np.cos(th), dtype=F64), torch.tensor(a, dtype=F64))) * np.cos(th) * np.sin(th) * 2 * np.pi
        val, _ = integrate.quad(f, 0.0, np.pi / 2, limit=200)
        assert abs(val - 1.0) <= 1e-6, a
    # the sampler draws the microfacet normal with tan^2(theta) = a^2 u / (1 - u), i.e. u = CDF(theta)
    rng = np.random.default_rng(6)
    u1 = torch.from_numpy(rng.uniform(0.01, 0.99, size=(50, 1))); u2 = torch.from_numpy(rng.uniform(size=(50, 1)))
    wo = torch.tensor([[0.0, 0.0, 1.0]], dtype=F64).expand(50, 1, 3)          # view along the normal: reflect(wo, n) = 2 n_z n - wo
    alpha = torch.full((50, 1), 0.4, dtype=F64)
    d, pdf = mr.microfacet_sample(u1, u2, wo, alpha)
    n = (d + wo); n = n / torch.linalg.norm(n, dim=-1, keepdim=True)            # half vector = sampled microfacet normal
    tan2 = (1 - n[..., 2] ** 2) / n[..., 2] ** 2
    assert np.abs((tan2 / (0.4 ** 2 + tan2)).numpy() - u1.numpy()).max() <= 1e-9
    assert np.abs(pdf.numpy() / mr.microfacet_pdf(wo, d, alpha).numpy() - 1).max() <= 1e-9


def test_cosine_sampler_density_by_quadrature():
    from oracle import material_ref as mr
    rng = np.random.default_rng(7)
    u1, u2 = torch.from_numpy(rng.uniform(size=(4000,))), torch.from_numpy(rng.uniform(size=(4000,)))
    d, pdf = mr.cosine_sample(u1, u2)
    d = d.numpy()
    assert np.abs(np.linalg.norm(d, axis=-1) - 1).max() <= 1e-9 and np.abs(pdf.numpy() - d[:, 2] / np.pi).max() <= 1e-12
    assert abs(d[:, 2].mean() - 2.0 / 3.0) <= 0.02                              # E[cos] under p = cos / pi
