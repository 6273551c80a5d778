"""Known-answer tests of the time-resolved oracle pieces (oracle/transient_ref.py).

The reference keeps no fixtures for this path either (parity unpinned); what CAN be pinned
independently is pinned here: the time shift against scipy.ndimage.map_coordinates (the function
jax.scipy.ndimage.map_coordinates re-implements, internal/render.py:480-496), the temporal filter against
scipy.signal.convolve(mode="same") (internal/render.py:415-417), the flattened direct scatter against a literal
loop (internal/render.py:436-477), and the travel-time masks against hand-computed bins."""
import os

import numpy as np
import pytest
import torch

import common
import nrc_amd
from oracle import transient_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_shift_map_coordinates_matches_scipy():
    from scipy import ndimage
    rng = np.random.default_rng(0)
    n, nb = 5, 64
    hist = rng.uniform(size=(n, nb, 3)).astype(np.float32)
    move = np.array([0.0, 0.0312, 0.251, 0.6399, 0.8], np.float32)       # distances; exposure 0.01 -> 0, 3.12, 25.1, 63.99, 80 bins
    out = transient_ref.shift_map_coordinates(torch.from_numpy(hist), torch.from_numpy(move), 0.01, nb).numpy()
    X, Y, Z = np.meshgrid(np.arange(n), np.arange(nb), np.arange(3), indexing="ij")
    Yf = Y.astype(np.float32) - (move / np.float32(0.01))[:, None, None]
    # jax.scipy.ndimage.map_coordinates(mode="constant") validates each interpolation corner separately, i.e. it
    # interpolates towards cval across the edge: scipy calls that mode "grid-constant"
    ref = ndimage.map_coordinates(hist, np.stack([X.astype(np.float32), Yf, Z.astype(np.float32)]), order=1,
                                  mode="grid-constant", cval=0.0, prefilter=False)
    assert np.abs(out - ref).max() <= 2e-6
    assert np.array_equal(out[0], hist[0])                    # zero shift is the identity
    assert np.all(out[4] == 0)                                # shifted out of the histogram


def test_temporal_filter_matches_scipy_convolve_same():
    from scipy import signal
    f = transient_ref.gauss_filter(3.0).numpy()
    assert f.shape == (25,) and abs(f.sum() - 1) < 1e-6 and np.allclose(f, f[::-1])
    k = np.arange(-12, 13)
    want = np.exp(-(k ** 2) / 18.0) - np.exp(-8.0)
    assert np.abs(f - want / want.sum()).max() <= 1e-7
    # the filter as applied inside transient_integrate
    rng = np.random.default_rng(1)
    x = rng.uniform(size=(3, 700, 3)).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 2, 1).reshape(9, 1, 700)
    y = torch.nn.functional.conv1d(xt, torch.flip(torch.from_numpy(f), [0]).reshape(1, 1, -1), padding=12)
    y = y.reshape(3, 3, 700).permute(0, 2, 1).numpy()
    ref = signal.convolve(x, f[None, :, None], mode="same")
    assert np.abs(y - ref).max() <= 2e-6


def test_shift_direct_is_the_flattened_scatter():
    """Bins >= n_bins of ray r land in ray r + 1 (dropped for the last ray); floor / ceil weights."""
    nb = 10
    dists = torch.tensor([[2.25, 9.5, 11.0], [0.0, 3.0, 12.75]])
    rgb = torch.tensor([[[1.0, 2, 3], [4, 5, 6], [7, 8, 9]], [[1.0, 1, 1], [2, 2, 2], [3, 3, 3]]])
    w = torch.tensor([[0.5, 1.0, 2.0], [1.0, 0.25, 4.0]])
    out = transient_ref.shift_direct(dists, rgb, w, nb).numpy()
    ref = np.zeros((2 * nb, 3), np.float64)
    for r in range(2):
        for s in range(3):
            d = float(dists[r, s]); lo = max(np.floor(d), 0); hi = np.ceil(d)
            v = float(w[r, s]) * rgb[r, s].numpy().astype(np.float64)
            for idx, wt in ((r * nb + int(lo), 1 - (d - lo)), (r * nb + int(hi), d - lo)):
                if 0 <= idx < 2 * nb:
                    ref[idx] += v * wt
    assert np.abs(out.reshape(-1, 3) - ref).max() <= 1e-6
    assert out[0, 2, 0] == pytest.approx(0.5 * 0.75) and out[0, 3, 0] == pytest.approx(0.5 * 0.25)
    assert out[1, 1, 0] == pytest.approx(2.0 * 7.0)           # ray 0, bin 11 -> ray 1, bin 1
    assert out[1, 0, 0] == pytest.approx(1.0 + 0.5 * 4.0)     # ray 0's 9.5 -> (9, 10): bin 10 spills; + ray 1's own 0.0


def test_zero_invalid_bins_masks():
    cfg = nrc_amd.cornell_transient_config()
    t = cfg.transient
    rays = dict(origins=torch.tensor([[0.0, 0, 0]]), cam_origins=torch.tensor([[0.0, 0, 0]]), lights=torch.tensor([[0.0, 0, 0.5]]))
    means = torch.tensor([[[0.0, 0.0, 2.0], [0.0, 0.0, 0.6]]])           # light distances 1.5 and 0.1
    ones = torch.ones(1, 2, t.n_bins, 3)
    d, s = transient_ref.zero_invalid_bins(cfg, rays, means, ones, 2 * ones)
    d = d.numpy()
    # sample 0: kept bins satisfy (b + 100) * 0.01 >= 1.5  and  b * 0.01 + 2.0 <= 6.99
    kept = np.nonzero(d[0, 0, :, 0])[0]
    assert kept.min() == 50 and kept.max() == 499
    assert np.all(d[0, 1] == 0)                                # light closer than light_near = 0.7: everything zeroed
    assert np.array_equal(s.numpy() != 0, d != 0)


def test_transient_render_is_self_consistent():
    out = common.oracle_transient(24, jitter_seed=3)
    r = {k: v.numpy() for k, v in out["render"].items()}
    assert r["rgb"].shape == (24, 700, 3) and np.all(np.isfinite(r["rgb"])) and r["rgb"].min() >= 0
    assert np.abs(r["rgb"] - (r["transient_direct_viz"] + r["transient_indirect_viz"])).max() <= 1e-7
    assert np.abs(r["integrated_rgb"] - r["rgb"].sum(1)).max() <= 1e-5
    assert np.abs(r["direct_rgb"] + r["indirect_rgb"] - r["integrated_rgb"]).max() <= 2e-5
    # the temporal filter is normalised and the shifts only move energy (or drop it at the ends)
    nf = r["transient_direct_no_filter"]
    assert np.all(r["transient_direct_viz"].sum(1) <= nf.sum(1) * (1 + 1e-5) + 1e-6)
    sh = out["shader"]
    w = sh["weights"].numpy()
    unshifted = (w[..., None, None] * sh["transient_indirect"].numpy()).sum(1)
    assert np.all(r["transient_indirect_viz"].sum(1) <= unshifted.sum(1) * (1 + 1e-5) + 1e-6)
    # per-sample sums over bins feed the scalar composites
    assert np.abs(sh["indirect_diffuse_rgb"].numpy() - sh["transient_indirect_diffuse"].numpy().sum(-2)).max() <= 1e-5


def test_shadow_occlusion_thresholds_and_darkens():
    """use_occlusions: occ is 0 or > occ_threshold, never in between; it can only remove direct light."""
    a = common.oracle_transient(8, jitter_seed=5, density_shift=6.0)
    b = common.oracle_transient(8, jitter_seed=5, occlusions=True, shadow_jitter_seed=13, density_shift=6.0)
    occ = b["shader"]["occ"].numpy()
    lit = b["shader"]["n_dot_l_rgb"].numpy() > 0
    vals = occ[lit]
    assert np.all((vals == 0) | (vals > 0.9)) and np.all(occ[~lit] == 1)
    assert 0.02 < (vals > 0).mean() < 0.98                 # dense field (+6 on the density bias): some shadow rays saturate
    assert np.all(b["render"]["direct_rgb"].numpy() <= a["render"]["direct_rgb"].numpy() * (1 + 1e-6) + 1e-7)
    # the indirect part does not see the shadow rays
    assert np.abs(a["render"]["transient_indirect_viz"].numpy() - b["render"]["transient_indirect_viz"].numpy()).max() <= 1e-7


@pytest.mark.parametrize("name", ["transient_16_det.npz", "transient_16_jit.npz", "transient_8_occ.npz"])
def test_transient_oracle_vs_golden(name):
    g = dict(np.load(os.path.join(GOLD, name)))
    n, js, occ, sjs = (int(v) for v in g["meta"])
    r = common.oracle_transient(n, jitter_seed=None if js < 0 else js, occlusions=bool(occ),
                                shadow_jitter_seed=None if sjs < 0 else sjs)["render"]
    for k in ("rgb", "integrated_rgb", "acc", "diffuse_rgb", "specular_rgb", "distance_median"):
        want = g["render_" + k]
        got = r[k].numpy()
        tol = 2e-4 * max(1.0, float(np.abs(want).max()))       # fp32 oracle against the fp64 golden
        assert np.abs(got - want).max() <= tol, k
