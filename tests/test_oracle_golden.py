"""The float32 oracle (the CPU baseline / the checker of the HIP path) against the committed
float64 golden vectors (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import common
import nrc_amd
from oracle import hashgrid_ref, mathx, stepfun_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.mark.parametrize("name", ["hotdog_cache_256_det.npz", "hotdog_cache_256_jit.npz", "hotdog_cache_64_shell.npz"])
def test_cache_fp32_oracle_vs_fp64_golden(name):
    g = _load(name)
    n, js, shift = int(g["meta"][0]), int(g["meta"][1]), float(g["meta"][2])
    out = common.oracle_cache(n, dtype=torch.float32, jitter_seed=None if js < 0 else js, density_shift=shift)
    # fp32 vs fp64 of the same arithmetic: the tolerance is the fp32 noise of the path itself
    # (hash-grid coordinates at N=2048 amplify 1-ulp position differences on random tables).
    r = out["render"]
    assert np.abs(r["rgb"].numpy() - g["render_rgb"]).max() <= 1e-4
    assert np.abs(r["acc"].numpy() - g["render_acc"]).max() <= 1e-4
    for k in ("diffuse_rgb", "specular_rgb", "direct_rgb", "indirect_rgb", "albedo_rgb", "normals_pred"):
        assert np.abs(r[k].numpy() - g["render_" + k]).max() <= 5e-4, k
    for k in ("distance_mean", "distance_median", "distance_percentile_5", "distance_percentile_95"):
        assert np.abs(r[k].numpy() - g["render_" + k]).max() <= 2e-3, k
    for l in range(3):
        assert np.abs(out["sampler"][l]["tdist"].numpy() - g[f"l{l}_tdist"]).max() <= 1e-4
        assert np.abs(out["sampler"][l]["weights"].numpy() - g[f"l{l}_weights"]).max() <= 5e-4
    assert set("render_" + k for k in r) == set(k for k in g if k.startswith("render_"))


def test_hashgrid_fp32_vs_golden():
    g = _load("operators.npz")
    cfg = nrc_amd.hotdog_config()
    wt = common.weights_torch()
    pts = torch.from_numpy(g["points"])
    grids = [("params/Cache/Sampler/MLP_0/density_grid", cfg.proposal_grids[0]),
             ("params/Cache/Sampler/MLP_1/density_grid", cfg.proposal_grids[1]),
             ("params/Cache/Sampler/MLP_2/density_grid", cfg.proposal_grids[2]),
             ("params/Cache/Shader/appearance_grid", cfg.appearance_grid)]
    for gid, (prefix, gc) in enumerate(grids):
        out = hashgrid_ref.hash_encoding(wt, prefix, gc, mathx.contract_radius(pts, cfg.contract_radius)).numpy()
        assert out.shape == g[f"grid{gid}"].shape
        assert np.abs(out - g[f"grid{gid}"]).max() <= 2e-3     # contracted far points: 1-ulp coordinate noise x N=2048


def test_sample_intervals_fp32_vs_golden():
    g = _load("operators.npz")
    t, lg, jit = torch.from_numpy(g["si_t"]), torch.from_numpy(g["si_logits"]), torch.from_numpy(g["si_jitter"])
    assert np.abs(stepfun_ref.sample_intervals(None, t, lg, 32).numpy() - g["si_out_det"]).max() <= 1e-4
    assert np.abs(stepfun_ref.sample_intervals(jit, t, lg, 32).numpy() - g["si_out_jit"]).max() <= 1e-4
    t1 = torch.tensor([[0.0, 1.0]]).repeat(4, 1)
    assert np.abs(stepfun_ref.sample_intervals(None, t1, torch.zeros(4, 1), 64).numpy() - g["si_out_1bin"]).max() <= 1e-6
