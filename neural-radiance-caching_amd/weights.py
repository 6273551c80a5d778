"""Parameter inventory of the hotdog radiance cache, keyed by Flax tree path.

`rc_load_weights` accepts exactly these names (SURVEY.md §8a').  Dense kernels
are Flax layout `[in, out]`, `y = x @ kernel + bias` (flax.linen.Dense).  Grid
names follow internal/grid_utils.py:796-798, 851-852; module names follow
internal/models.py:784-811, 2196-2226, internal/sampling.py:126,
internal/geometry.py:123-153, internal/nerf.py:232-346,
internal/surface_light_field.py:343-403, internal/material.py,
internal/light_sampler.py.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

from .config import GridConfig, RenderConfig

P = "params/"


def _grid(shapes, prefix: str, g: GridConfig):
    for n in g.grid_sizes:
        if g.is_dense(n):
            shapes[f"{prefix}/{g.level_name(n)}"] = (n, n, n, g.num_features)
        else:
            shapes[f"{prefix}/{g.level_name(n)}"] = (g.hash_map_size, g.num_features)


def _dense(shapes, path: str, fan_in: int, fan_out: int):
    shapes[f"{path}/kernel"] = (fan_in, fan_out)
    shapes[f"{path}/bias"] = (fan_out,)


def ide_dim(deg_view: int) -> int:
    return 2 * sum(2 ** i + 1 for i in range(deg_view))


def _transient_shader_shapes(s, cfg: RenderConfig, sh: str, feat: int, B: int):
    """TransientNeRFMLP (internal/nerf.py:232-414 with Config.use_transient) and its
    TransientSurfaceLightFieldMLP (surface_light_field.py:343-403, use_lights, use_indirect) as configured by
    transient_simulation_ngp_yobo(_cornell).gin; only what the active branch (nerf.py:691-938) with
    use_ambient=False reads."""
    t = cfg.transient
    nl = 3 + 6 * t.deg_lights                      # pos_enc(lights, 0, deg_lights) + identity
    _dense(s, f"{sh}/bottleneck_layer", feat, B)
    _dense(s, f"{sh}/roughness_layer", feat, 1)
    _dense(s, f"{sh}/tint_layer", feat, 3)
    _dense(s, f"{sh}/direct_tint_layer", feat, 3)
    _dense(s, f"{sh}/albedo_layer", feat, 3)
    _dense(s, f"{sh}/integrated_brdf_layers_0", B + 1, cfg.ibrdf_width)
    _dense(s, f"{sh}/integrated_brdf_layers_1", cfg.ibrdf_width, cfg.ibrdf_width)
    _dense(s, f"{sh}/output_integrated_brdf_layer", cfg.ibrdf_width, 1)
    _dense(s, f"{sh}/brdf_layers_0", B + 3 + 6 * t.deg_brdf, t.brdf_width)
    _dense(s, f"{sh}/brdf_layers_1", t.brdf_width, t.brdf_width)
    _dense(s, f"{sh}/output_brdf_layer", t.brdf_width, 1)
    _dense(s, f"{sh}/irradiance_layers_0", feat + nl, t.irradiance_width)
    _dense(s, f"{sh}/irradiance_layers_1", t.irradiance_width, t.irradiance_width)
    _dense(s, f"{sh}/transient_indirect_layer", t.irradiance_width, 3 * t.n_bins)
    s[f"{sh}/light_power"] = (1,)
    sl = f"{sh}/SurfaceLightField"
    in_dim = B + ide_dim(cfg.slf_deg_view) + nl
    _dense(s, f"{sl}/layer_0", in_dim, cfg.slf_width)
    _dense(s, f"{sl}/layer_1", cfg.slf_width, cfg.slf_width)
    _dense(s, f"{sl}/layer_2", cfg.slf_width, cfg.slf_width)
    _dense(s, f"{sl}/layer_bottleneck", cfg.slf_width + in_dim, cfg.slf_width)
    _dense(s, f"{sl}/output_rgba_layer", cfg.slf_width, 3 * t.n_bins + 1)


def param_shapes(cfg: RenderConfig, passes: Tuple[str, ...] = ("cache",)) -> "OrderedDict[str, tuple]":
    """name -> shape for every tensor the given passes read."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    W = cfg.density_width
    for lvl, g in enumerate(cfg.proposal_grids):
        base = f"{P}Cache/Sampler/MLP_{lvl}"
        _grid(s, f"{base}/density_grid", g)
        _dense(s, f"{base}/density_layers_0", g.out_dim, W)
        _dense(s, f"{base}/density_layers_1", W, W)
        _dense(s, f"{base}/output_density_layer", W, 1)
        if lvl == cfg.num_levels - 1:
            _dense(s, f"{base}/pred_normals_layer", W, 3)
    sh = f"{P}Cache/Shader"
    _grid(s, f"{sh}/appearance_grid", cfg.appearance_grid)
    feat = W + cfg.appearance_grid.out_dim
    B = cfg.bottleneck_width
    if cfg.transient is not None:
        _transient_shader_shapes(s, cfg, sh, feat, B)
        return s
    _dense(s, f"{sh}/bottleneck_layer", feat, B)
    _dense(s, f"{sh}/roughness_layer", feat, 1)
    _dense(s, f"{sh}/tint_layer", feat, 3)
    _dense(s, f"{sh}/ambient_irradiance_layer", feat, 3)
    _dense(s, f"{sh}/irradiance_layer", feat, 3)
    _dense(s, f"{sh}/integrated_brdf_layers_0", B + 1, cfg.ibrdf_width)
    _dense(s, f"{sh}/integrated_brdf_layers_1", cfg.ibrdf_width, cfg.ibrdf_width)
    _dense(s, f"{sh}/output_integrated_brdf_layer", cfg.ibrdf_width, 1)

    def slf(path, in_dim, width, bott):
        _dense(s, f"{path}/layer_0", in_dim, width)
        _dense(s, f"{path}/layer_1", width, width)
        _dense(s, f"{path}/layer_2", width, width)
        _dense(s, f"{path}/layer_bottleneck", width + in_dim, bott)
        _dense(s, f"{path}/output_rgba_layer", bott, 4)
        _dense(s, f"{path}/output_ambient_rgb_layer", bott, 3)

    slf(f"{sh}/SurfaceLightField", B + ide_dim(cfg.slf_deg_view), cfg.slf_width, cfg.slf_width)
    slf(f"{sh}/EnvMap", ide_dim(cfg.cache_env_deg_view), cfg.slf_width, cfg.slf_width)
    slf(f"{P}Cache/EnvMap", 3 + 6 * cfg.env_deg_view, cfg.env_width, cfg.env_bottleneck_width)
    if "material" in passes:
        m = f"{P}MaterialShader"
        _grid(s, f"{m}/material_grid", cfg.material_grid)
        _dense(s, f"{m}/bottleneck_layer", cfg.material_grid.out_dim, B)
        _dense(s, f"{m}/pred_brdf_layer", B, 10)
    if "light" in passes or "material" in passes:
        l = f"{P}LightSampler"
        _grid(s, f"{l}/light_grid", cfg.light_grid)
        _dense(s, f"{l}/layers_0", cfg.light_grid.out_dim, 64)
        _dense(s, f"{l}/layers_1", 64, 64)
        _dense(s, f"{l}/output_layer", 64, 5 * cfg.num_vmf)
    return s


def synthetic_weights(cfg: RenderConfig, passes=("cache",), seed: int = 1, density_shift: float = 0.0,
                      table_range: float = 0.05, level_decay: float = 1.0) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights: tables U(+-table_range), dense kernels
    He-uniform U(+-sqrt(6/fan_in)), biases U(+-0.1).  `density_shift` is added to the
    output_density_layer biases (the "shell" set uses +4 so rays saturate).
    `level_decay` < 1 scales the table of grid level l by level_decay**l: with 0.5 every level
    contributes the same spatial gradient (like a trained multiresolution field) instead of the
    white-noise field of level_decay = 1, whose fine levels amplify one-ulp position differences
    by N = 2048 ("smooth" set of the strict parity tests).

    There is no trained checkpoint in the build environment; these stand in for one.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(cfg, passes).items():
        leaf = name.rsplit("/", 1)[-1]
        if leaf.startswith(("grid_", "hash_")):
            level = int(round(np.log2(int(leaf.split("_")[1]) / 16.0)))
            a = rng.uniform(-table_range, table_range, size=shape) * (level_decay ** level)
        elif leaf == "kernel":
            lim = np.sqrt(6.0 / shape[0])
            a = rng.uniform(-lim, lim, size=shape)
        elif leaf == "light_power":
            a = np.full(shape, cfg.transient.light_power_bias)      # nerf.py:400-409 (light_init)
        else:
            a = rng.uniform(-0.1, 0.1, size=shape)
            if name.endswith("output_density_layer/bias"):
                a = a + density_shift
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out
