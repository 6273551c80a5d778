#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle in float64 ("spec" precision).

The reference (JAX/Flax/gin) cannot be imported or run in the build image and ships no
fixtures for this path, so these vectors are produced by the oracle itself:
they pin the oracle against regressions and give the HIP path a float64 target, they do NOT
pin the oracle to the reference (parity unpinned, see oracle/__init__.py).

Inputs are regenerated from seeds (nrc_amd.synthetic_rays / synthetic_weights), only outputs
are stored (float32-rounded float64 results).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import nrc_amd  # noqa: E402
import common  # noqa: E402
from oracle import cache_ref, hashgrid_ref, mathx, stepfun_ref  # noqa: E402

F64 = torch.float64


def cache_case(n_rays, jitter_seed, density_shift, name):
    out = common.oracle_cache(n_rays, dtype=F64, jitter_seed=jitter_seed, density_shift=density_shift)
    d = {}
    for l, lvl in enumerate(out["sampler"]):
        for k in ("sdist", "tdist", "density", "weights"):
            d[f"l{l}_{k}"] = lvl[k].numpy().astype(np.float32)
    d["shade_rgb"] = out["shader"]["rgb"].numpy().astype(np.float32)
    for k, v in out["render"].items():
        d["render_" + k] = v.numpy().astype(np.float32)
    d["meta"] = np.array([n_rays, -1 if jitter_seed is None else jitter_seed, density_shift], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, {k: v.shape for k, v in list(d.items())[:4]}, "...")


def cache_case_spec(n_rays, jitter_seed, density_shift, name, out_dir=None):
    """The same fixture computed by the independent numpy float64 spec (oracle/spec_np.py) instead of the torch oracle:
    `python tests/golden/make_golden.py --spec [dir]`.  The two agree to float64 round-off (tests/test_oracle_spec.py),
    so after the float32 rounding of the stored arrays the files are equal up to the last bit of a few entries."""
    from oracle import spec_np
    rays = nrc_amd.synthetic_rays(n_rays)
    jit = None if jitter_seed is None else common.jitters(n_rays, seed=jitter_seed)
    out = spec_np.cache_forward(common.weights_np(density_shift), nrc_amd.hotdog_config(), rays.hot_fields(), jit)
    d = {}
    for l, lvl in enumerate(out["levels"]):
        for k in ("sdist", "tdist", "density", "weights"):
            d[f"l{l}_{k}"] = lvl[k].astype(np.float32)
    d["shade_rgb"] = out["per_sample"]["rgb"].astype(np.float32)
    for k, v in out["render"].items():
        d["render_" + k] = v.astype(np.float32)
    d["meta"] = np.array([n_rays, -1 if jitter_seed is None else jitter_seed, density_shift], dtype=np.float64)
    np.savez_compressed(os.path.join(out_dir or HERE, name), **d)
    print("[spec]", name)


def operator_cases():
    cfg = nrc_amd.hotdog_config()
    wt = common.weights_torch(dtype=F64)
    rng = np.random.default_rng(123)
    pts = rng.uniform(-3, 3, size=(512, 3)).astype(np.float32)
    pts[:8] = [[0, 0, 0], [1, 1, 1], [-1, -1, -1], [2, 0, 0], [0, -2, 0], [0.999, 0.999, 0.999], [5, 5, 5], [-5, 3, 0.1]]
    d = {"points": pts}
    grids = [("params/Cache/Sampler/MLP_0/density_grid", cfg.proposal_grids[0]),
             ("params/Cache/Sampler/MLP_1/density_grid", cfg.proposal_grids[1]),
             ("params/Cache/Sampler/MLP_2/density_grid", cfg.proposal_grids[2]),
             ("params/Cache/Shader/appearance_grid", cfg.appearance_grid)]
    for gid, (prefix, g) in enumerate(grids):
        x = mathx.contract_radius(torch.from_numpy(pts).to(F64), cfg.contract_radius)
        d[f"grid{gid}"] = hashgrid_ref.hash_encoding(wt, prefix, g, x).numpy().astype(np.float32)
    # sample_intervals: random histogram, spiky histogram, single bin
    P, S, n = 64, 32, 64
    t = np.sort(rng.uniform(size=(n, P + 1)), axis=-1).astype(np.float32)
    t[:, 0], t[:, -1] = 0, 1
    lg = (rng.normal(size=(n, P)) * 3).astype(np.float32)
    lg[:8] = -30.0
    lg[np.arange(8), rng.integers(0, P, 8)] = 10.0        # spiky
    jit = rng.uniform(size=(n, 1)).astype(np.float32)
    d.update(si_t=t, si_logits=lg, si_jitter=jit)
    d["si_out_det"] = stepfun_ref.sample_intervals(None, torch.from_numpy(t).to(F64), torch.from_numpy(lg).to(F64), S).numpy().astype(np.float32)
    d["si_out_jit"] = stepfun_ref.sample_intervals(torch.from_numpy(jit).to(F64), torch.from_numpy(t).to(F64), torch.from_numpy(lg).to(F64), S).numpy().astype(np.float32)
    t1 = np.tile(np.array([[0.0, 1.0]], dtype=np.float32), (4, 1))
    d["si_out_1bin"] = stepfun_ref.sample_intervals(None, torch.from_numpy(t1).to(F64), torch.zeros(4, 1, dtype=F64), 64).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "operators.npz"), **d)
    print("operators.npz", list(d.keys()))


def transient_case(n_rays, jitter_seed, name, occlusions=False, shadow_jitter_seed=None):
    """Time-resolved cornell cache (oracle/transient_ref.py) in float64; only the render dict is stored."""
    out = common.oracle_transient(n_rays, jitter_seed=jitter_seed, dtype=F64, occlusions=occlusions,
                                  shadow_jitter_seed=shadow_jitter_seed)
    d = {"render_" + k: v.numpy().astype(np.float32) for k, v in out["render"].items()
         if k not in ("transient_direct", "transient_indirect", "transient_direct_no_filter", "transient_indirect_no_filter",
                      "weights", "dists")}
    d["meta"] = np.array([n_rays, -1 if jitter_seed is None else jitter_seed, 1 if occlusions else 0,
                          -1 if shadow_jitter_seed is None else shadow_jitter_seed], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, sorted(d.keys())[:6], "...")


def material_case(n_rays, rays_seed, rnd_seed, name):
    """Material stage (configs[2]) on the smooth weight set in float64: the render dict plus the categorical picks of
    the primary rays and of the secondary trace (handed to the HIP path through rc_material_randoms.*resample_inds)."""
    from oracle import material_ref
    cfg = nrc_amd.hotdog_config()
    wt = common.to_torch(common.weights_material_np(True), F64)
    rays = nrc_amd.synthetic_rays(n_rays, seed=rays_seed)
    rnd = material_ref.draw_randoms(cfg, n_rays, seed=rnd_seed)
    out = material_ref.material_forward(wt, cfg, common.rays_torch(rays, F64), rnd)
    d = {"render_" + k: v.numpy().astype(np.float32) for k, v in out["render"].items()}
    d["inds"] = out["inds"][:, 0].numpy().astype(np.int8)
    d["spec_inds"] = out["debug"]["specular"]["inds"].numpy().astype(np.int8)
    d["diff_inds"] = out["debug"]["diffuse"]["inds"].numpy().astype(np.int8)
    d["meta"] = np.array([n_rays, rays_seed, rnd_seed], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, sorted(d.keys())[:6], "...")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--spec":          # the cache fixtures from the independent numpy spec
        out_dir = sys.argv[2] if len(sys.argv) > 2 else None
        cache_case_spec(256, None, 0.0, "hotdog_cache_256_det.npz", out_dir)
        cache_case_spec(256, 7, 0.0, "hotdog_cache_256_jit.npz", out_dir)
        cache_case_spec(64, 11, 4.0, "hotdog_cache_64_shell.npz", out_dir)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "material":        # only the material fixture
        material_case(64, 78, 5, "hotdog_material_64_smooth.npz")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "transient":       # only the transient fixtures
        transient_case(16, None, "transient_16_det.npz")
        transient_case(16, 5, "transient_16_jit.npz")
        transient_case(8, 5, "transient_8_occ.npz", occlusions=True, shadow_jitter_seed=13)
        sys.exit(0)
    transient_case(16, None, "transient_16_det.npz")
    transient_case(16, 5, "transient_16_jit.npz")
    transient_case(8, 5, "transient_8_occ.npz", occlusions=True, shadow_jitter_seed=13)
    material_case(64, 78, 5, "hotdog_material_64_smooth.npz")
    cache_case(256, None, 0.0, "hotdog_cache_256_det.npz")
    cache_case(256, 7, 0.0, "hotdog_cache_256_jit.npz")
    cache_case(64, 11, 4.0, "hotdog_cache_64_shell.npz")
    operator_cases()
