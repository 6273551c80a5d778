"""The two witnesses against each other: the torch oracle (oracle/*_ref.py, run in float64) and the numpy float64 spec
(oracle/spec_np.py), which was written from the reference lines in a separate pass and shares no code with it
(asserted below).  Agreement to float64 round-off on every fixture says that neither restatement carries a
transcription error the other lacks -- sampler level loop (sampling.py:284-639), passive shader (nerf.py:940-1090),
integrator (render.py:172-247), material strategy table (material.py:2705-2808), transient composite
(render.py:250-507), analytic normals (autograd there, a hand-written backward pass here).  It says nothing about the
reference itself: no reference-held vectors exist and it cannot be imported here -- parity stays unpinned."""
import ast
import os

import numpy as np
import pytest
import torch

import common
import nrc_amd
from oracle import cache_ref, material_ref, spec_np

F64 = torch.float64
TOL = 1e-9          # float64 round-off through ~200 chained operations; fp32-level disagreement would be 1e-6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _worst(spec: dict, ref: dict, keys=None):
    worst = {}
    for k in (keys or spec.keys()):
        if k in ref and ref[k] is not None and spec[k] is not None:
            a = np.asarray(spec[k])
            worst[k] = float(np.abs(a - ref[k].numpy().reshape(a.shape)).max())
    return worst


def test_spec_shares_no_code_with_the_torch_oracle():
    tree = ast.parse(open(os.path.join(ROOT, "oracle", "spec_np.py")).read())
    mods = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            mods.update(a.name for a in node.names)
        elif isinstance(node, ast.ImportFrom):
            mods.add(("." * node.level) + (node.module or ""))
    assert mods <= {"__future__", "itertools", "math", "numpy"}, mods


@pytest.mark.parametrize("jitter_seed,shift", [(None, 0.0), (7, 0.0), (11, 4.0)])
def test_cache_forward_primary_rays(jitter_seed, shift):
    cfg = nrc_amd.hotdog_config()
    n = 40
    rays = nrc_amd.synthetic_rays(n)
    jit = None if jitter_seed is None else common.jitters(n, seed=jitter_seed)
    w = common.weights_np(shift)
    s = spec_np.cache_forward(w, cfg, rays.hot_fields(), jit)
    o = cache_ref.material_model_cache_only(common.weights_torch(shift, dtype=F64), cfg, common.rays_torch(rays, F64),
                                            None if jit is None else [torch.from_numpy(j).to(F64) for j in jit])
    assert set(s["render"]) == set(o["render"])
    worst = _worst(s["render"], o["render"])
    assert max(worst.values()) <= TOL, max(worst.items(), key=lambda kv: kv[1])
    for l in range(3):
        for k in ("sdist", "tdist", "density", "weights", "means"):
            assert np.abs(s["levels"][l][k] - o["sampler"][l][k].numpy()).max() <= TOL, (l, k)
    # analytic normals: explicit backward pass (spec) vs autograd (torch oracle)
    assert np.abs(s["levels"][2]["normals"] - o["sampler"][2]["normals"].numpy()).max() <= 1e-8
    assert np.abs(s["per_sample"]["rgb"] - o["shader"]["rgb"].numpy()).max() <= TOL


def test_cache_forward_secondary_and_resampled_rays():
    cfg = nrc_amd.hotdog_config()
    n = 40
    w, wt = common.weights_np(), common.weights_torch(dtype=F64)
    sr, rnd = common.secondary_case(n, seed=5)
    for use_env in (True, False):
        s = spec_np.cache_forward(w, cfg, sr, rnd["jitter"], secondary=True, gumbel=rnd["gumbel"], use_env_map=use_env)
        o = cache_ref.cache_forward(wt, cfg, common.rays_dict_torch(sr, F64), [torch.from_numpy(j)[:, None].to(F64) for j in rnd["jitter"]],
                                    is_secondary=True, gumbel=torch.from_numpy(rnd["gumbel"]).to(F64), use_env_map=use_env,
                                    want_grad_normals=False)
        assert np.array_equal(s["inds"], o["filtered_sampler_inds"][:, 0].numpy())
        worst = _worst(s["integrator"], o["render"])
        assert len(worst) >= 30 and max(worst.values()) <= TOL, max(worst.items(), key=lambda kv: kv[1])
    rays = nrc_amd.synthetic_rays(n)
    g = np.random.default_rng(3).gumbel(size=(n, 32))
    s = spec_np.cache_forward(w, cfg, rays.hot_fields(), None, resample=True, gumbel=g, want_normals=False)
    o = cache_ref.cache_forward(wt, cfg, common.rays_torch(rays, F64), None, resample=True, gumbel=torch.from_numpy(g), want_grad_normals=False)
    assert np.array_equal(s["inds"], o["filtered_sampler_inds"][:, 0].numpy())
    worst = _worst(s["integrator"], o["render"])
    assert max(worst.values()) <= TOL, max(worst.items(), key=lambda kv: kv[1])


@pytest.mark.parametrize("smooth", [True, False])
def test_material_stage(smooth):
    cfg = nrc_amd.hotdog_config()
    wn = common.weights_material_np(smooth)
    n = 10
    rays = nrc_amd.synthetic_rays(n, seed=78)
    rnd = material_ref.draw_randoms(cfg, n, seed=5)
    o = material_ref.material_forward(common.to_torch(wn, F64), cfg, common.rays_torch(rays, F64), rnd)
    s = spec_np.material_forward(wn, cfg, rays.hot_fields(), rnd)
    assert np.array_equal(s["inds"], o["inds"][:, 0].numpy())
    for name in ("specular", "diffuse"):
        assert np.array_equal(s["debug"][name]["inds"], o["debug"][name]["inds"].numpy())
        assert np.abs(s["debug"][name]["wi"] - o["debug"][name]["local_lightdirs"].numpy()).max() <= TOL
        assert np.abs(s["debug"][name]["weight"] - o["debug"][name]["weight"].numpy()).max() <= TOL
        p = o["debug"][name]["pdf"].numpy()
        assert (np.abs(s["debug"][name]["pdf"] - p) <= TOL * (1.0 + p)).all()
    worst = _worst(s["render"], o["render"])
    assert len(worst) >= 45 and max(worst.values()) <= TOL, max(worst.items(), key=lambda kv: kv[1])
    worst = _worst(s["shader"], o["shader"])
    assert max(worst.values()) <= TOL


def test_transient_composite():
    """render.volumetric_transient_rendering: the spec's literal loops against the torch oracle's vectorised integrator on
    the oracle's own per-sample shader outputs (incl. bins spilling into the next ray and the 25-tap filter)."""
    from oracle import transient_ref
    cfg = nrc_amd.cornell_transient_config()
    out = common.oracle_transient(6, jitter_seed=5, dtype=F64)
    sh = out["shader"]
    s = spec_np.transient_composite(cfg.transient, sh["direct_rgb"].numpy(), sh["transient_indirect"].numpy(),
                                    sh["weights"].numpy(), sh["ray_dists"][..., 0].numpy(), sh["light_dists"][..., 0].numpy())
    r = out["render"]
    assert np.abs(s["transient_direct_no_filter"] - r["transient_direct_no_filter"].numpy()).max() <= TOL
    assert np.abs(s["transient_indirect"] - r["transient_indirect_viz"].numpy()).max() <= TOL
    # the 25 filter taps are float32 constants in the reference (jax default precision): numpy's and torch's float32
    # exp differ by one ulp on some of them (4e-9 after normalisation), which is all that separates the filtered outputs
    for k, kr in (("transient_direct", "transient_direct_viz"), ("rgb", "rgb"), ("integrated_rgb", "integrated_rgb")):
        assert np.abs(s[k] - r[kr].numpy()).max() <= 2e-8, k
    assert float(np.abs(s["rgb"]).max()) > 1e-4          # not a comparison of zeros


@pytest.mark.parametrize("name", ["hotdog_cache_256_det.npz", "hotdog_cache_256_jit.npz", "hotdog_cache_64_shell.npz"])
def test_goldens_are_what_the_spec_computes(name):
    """The committed float64 goldens (written by the torch oracle) equal what the independent spec computes, to the
    float32 rounding they are stored with: `make_golden.py --spec` regenerates them from this file."""
    g = dict(np.load(os.path.join(GOLD, name)))
    n, js, shift = int(g["meta"][0]), int(g["meta"][1]), float(g["meta"][2])
    n_check = min(n, 64)                                  # the first rays of the fixture (rays are independent)
    rays = nrc_amd.synthetic_rays(n)
    fields = {k: np.asarray(v)[:n_check] for k, v in rays.hot_fields().items()}
    jit = None if js < 0 else [j[:n_check] for j in common.jitters(n, seed=js)]
    s = spec_np.cache_forward(common.weights_np(shift), nrc_amd.hotdog_config(), fields, jit)
    for k, v in s["render"].items():
        want = g["render_" + k][:n_check]
        assert np.abs(v.reshape(want.shape) - want).max() <= 2e-6 * max(1.0, float(np.abs(want).max())), k
    for l in range(3):
        assert np.abs(s["levels"][l]["tdist"] - g[f"l{l}_tdist"][:n_check]).max() <= 1e-6


def test_material_golden_is_what_the_spec_computes():
    g = dict(np.load(os.path.join(GOLD, "hotdog_material_64_smooth.npz")))
    n, rays_seed, rnd_seed = (int(v) for v in g["meta"])
    cfg = nrc_amd.hotdog_config()
    m = 8                                                  # the first rays: their secondary rays are rows [i*K, (i+1)*K)
    rays = nrc_amd.synthetic_rays(n, seed=rays_seed)
    rnd = material_ref.draw_randoms(cfg, n, seed=rnd_seed)
    Ks = Kd = cfg.num_secondary_samples // 2
    cut = lambda a, k=1: np.asarray(a)[: m * k]
    sub = {k: (cut(v) if not isinstance(v, list) else v) for k, v in rnd.items()}
    sub["jitter"] = [cut(j) for j in rnd["jitter"]]
    for nm, K in (("spec", Ks), ("diff", Kd)):
        sub[nm + "_jitter"] = [cut(j, K) for j in rnd[nm + "_jitter"]]
        sub[nm + "_gumbel"] = cut(rnd[nm + "_gumbel"], K)
    fields = {k: np.asarray(v)[:m] for k, v in rays.hot_fields().items()}
    s = spec_np.material_forward(common.weights_material_np(True), cfg, fields, sub)
    assert np.array_equal(s["inds"], g["inds"][:m])
    assert np.array_equal(s["debug"]["specular"]["inds"], g["spec_inds"][: m * Ks])
    for k in ("rgb", "direct_rgb", "indirect_rgb", "diffuse_rgb", "specular_rgb", "lighting_irradiance", "material_albedo", "acc"):
        want = g["render_" + k][:m]
        assert np.abs(s["render"][k].reshape(want.shape) - want).max() <= 2e-6, k
