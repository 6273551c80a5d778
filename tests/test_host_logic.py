"""Host-side logic that needs no GPU: weight inventory, shard/unshard, render_image's chunking /
padding / Welford statistics, ray sharding bounds."""
import numpy as np
import pytest

import nrc_amd
from nrc_amd import model as M
from nrc_amd.config import RenderConfig
from nrc_amd.weights import ide_dim, param_shapes


def test_param_inventory_matches_survey_shapes():
    s = param_shapes(nrc_amd.hotdog_config(), passes=("cache", "material"))
    assert s["params/Cache/Sampler/MLP_0/density_grid/grid_016"] == (16, 16, 16, 1)
    assert s["params/Cache/Sampler/MLP_0/density_grid/hash_512"] == (524288, 1)
    assert s["params/Cache/Sampler/MLP_1/density_layers_0/kernel"] == (7, 64)
    assert s["params/Cache/Sampler/MLP_2/density_grid/hash_2048"] == (524288, 4)
    assert s["params/Cache/Sampler/MLP_2/pred_normals_layer/kernel"] == (64, 3)
    assert "params/Cache/Sampler/MLP_0/pred_normals_layer/kernel" not in s
    assert s["params/Cache/Shader/bottleneck_layer/kernel"] == (96, 128)
    assert s["params/Cache/Shader/integrated_brdf_layers_0/kernel"] == (129, 64)
    assert s["params/Cache/Shader/SurfaceLightField/layer_0/kernel"] == (200, 128)
    assert s["params/Cache/Shader/SurfaceLightField/layer_bottleneck/kernel"] == (328, 128)
    assert s["params/Cache/Shader/EnvMap/layer_0/kernel"] == (38, 128)
    assert s["params/Cache/EnvMap/layer_0/kernel"] == (27, 256)
    assert s["params/Cache/EnvMap/layer_bottleneck/kernel"] == (283, 128)
    assert s["params/MaterialShader/pred_brdf_layer/kernel"] == (128, 10)
    assert s["params/LightSampler/output_layer/kernel"] == (64, 640)
    assert ide_dim(5) == 72 and ide_dim(4) == 38
    total = sum(int(np.prod(v)) for k, v in param_shapes(nrc_amd.hotdog_config()).items()) * 4 / 2 ** 20
    assert 105 < total < 109          # ~106 MiB fp32 for the cache-only stage (SURVEY §8a')


def test_synthetic_weights_are_seeded():
    cfg = nrc_amd.hotdog_config()
    a = nrc_amd.synthetic_weights(cfg, seed=3)
    b = nrc_amd.synthetic_weights(cfg, seed=3)
    k = "params/Cache/Shader/tint_layer/kernel"
    assert np.array_equal(a[k], b[k]) and a[k].dtype == np.float32
    c = nrc_amd.synthetic_weights(cfg, seed=3, density_shift=4.0)
    kb = "params/Cache/Sampler/MLP_2/output_density_layer/bias"
    assert np.allclose(c[kb] - a[kb], 4.0)


def test_shard_unshard_roundtrip():
    x = np.arange(24, dtype=np.float32).reshape(8, 3)
    s = M.shard(x)
    assert s.shape == (1, 8, 3)
    assert np.array_equal(M.unshard(s), x)
    assert np.array_equal(M.unshard(s, padding=3), x[:5])


def test_shard_bounds_cover_everything():
    for n in (1, 7, 1024, 640000):
        for world in (1, 2, 3, 8):
            spans = [M.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= -(-n // world) for lo, hi in spans)


class _FakeCfg:
    render_chunk_size = 8


def _fake_render_fn(noise_seq):
    """rgb = origins (+ per-repeat noise), acc = near; mimics the [1, 1, m, ...] pmap axes."""
    calls = {"i": 0}

    def fn(rng, rays, passes, resample=None):
        o = rays.origins.reshape(-1, 3)
        noise = noise_seq[calls["i"] % len(noise_seq)]
        calls["i"] += 1
        out = {"rgb": (o + noise)[None, None], "acc": rays.near.reshape(-1)[None, None],
               "transient_foo": o[None, None], "distance_median": rays.far.reshape(-1)[None, None]}
        return out, rng

    return fn, calls


def test_render_image_chunks_pads_and_scatters():
    rays = nrc_amd.synthetic_camera_rays(5, 4)           # 20 rays, chunk 8 -> 3 chunks, last padded by 4
    fn, calls = _fake_render_fn([0.0])
    out, _ = M.render_image(fn, None, rays, _FakeCfg(), ("cache",), verbose=False)
    assert calls["i"] == 3
    assert out["rgb"].shape == (5, 4, 3) and out["acc"].shape == (5, 4)
    assert np.allclose(out["rgb"], rays.origins) and np.allclose(out["acc"], rays.near[..., 0])
    assert "transient_foo" not in out                     # models.py:2459 filter


def test_render_image_welford_mean_and_variance():
    rays = nrc_amd.synthetic_camera_rays(2, 4)
    noise = [0.0, 0.3, -0.6, 0.9]
    fn, _ = _fake_render_fn(noise)
    out, _ = M.render_image(fn, None, rays, _FakeCfg(), ("cache",), verbose=False, num_repeats=4, compute_variance=True)
    assert np.allclose(out["rgb"], rays.origins + np.mean(noise), atol=1e-6)
    # reference formula: M2 / (n - 1) * n  (models.py:2511)
    exp_var = np.var(noise, ddof=1) * 4
    assert np.allclose(out["rgb_variance"], exp_var, atol=1e-5)
    # non-stat keys keep the first repeat's value
    assert np.allclose(out["distance_median"], rays.far[..., 0])


def test_model_needs_the_hip_library(monkeypatch, tmp_path):
    from nrc_amd import rc_ext
    monkeypatch.setattr(rc_ext, "_LIB", None)
    monkeypatch.setattr(rc_ext, "library_path", lambda: str(tmp_path / "missing.so"))
    with pytest.raises(RuntimeError):
        M.Model(RenderConfig(), 0)


def test_flatten_variables():
    tree = {"params": {"Cache": {"Sampler": {"MLP_0": {"density_layers_0": {"kernel": 1, "bias": 2}}}}}}
    flat = M.flatten_variables(tree)
    assert flat == {"params/Cache/Sampler/MLP_0/density_layers_0/kernel": 1,
                    "params/Cache/Sampler/MLP_0/density_layers_0/bias": 2}
