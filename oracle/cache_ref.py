"""Radiance-cache forward pass (oracle; see oracle/__init__.py).

Restates, for the resolved hotdog configuration:
  ProposalVolumeSampler.__call__   internal/sampling.py:142-649
  render.cast_rays / alpha weights internal/render.py:26-169
  DensityMLP                       internal/geometry.py:155-168, 199-341, 381-586
  NeRFMLP cache shader             internal/shading.py:133-220, internal/nerf.py:561-689, 940-1090
  SurfaceLightFieldMLP             internal/surface_light_field.py:480-499, 782-1069
  VolumeIntegrator                 internal/integration.py:112-289, internal/render.py:172-247
  Model.maybe_resample             internal/models.py:193-292
  BaseNeRFModel.__call__           internal/models.py:657-774 (+ _handle_secondary :309-460)
"""
from __future__ import annotations

import math as pymath
from typing import Dict, Optional

import torch

from . import hashgrid_ref, mathx, stepfun_ref

P = "params/"


def dense(weights, path, x):
    """flax.linen.Dense: y = x @ kernel + bias, kernel [in, out]."""
    k = weights[f"{P}{path}/kernel"].to(x.dtype)
    b = weights[f"{P}{path}/bias"].to(x.dtype)
    return x @ k + b


# ----------------------------------------------------------------------------
# Ray casting and compositing weights
# ----------------------------------------------------------------------------
def cast_ray_means(tdist, origins, directions):
    """render.py:49-59, 26-28, 106-131 (means only; covs unused with the 'mean' basis)."""
    t0, t1 = tdist[..., :-1], tdist[..., 1:]
    s = t0 + t1
    d = t1 - t0
    eps = mathx.EPS ** 2
    ratio = d ** 2 / torch.clamp(3 * s ** 2 + d ** 2, min=eps)
    t_mean = s * (1 / 2 + ratio)
    return directions[..., None, :] * t_mean[..., None] + origins[..., None, :]


def compute_alpha_weights(density, tdist, dirs):
    """render.py:134-169 with opaque_background=False."""
    t_delta = tdist[..., 1:] - tdist[..., :-1]
    delta = t_delta * torch.linalg.norm(dirs[..., None, :], dim=-1)
    dd = density * torch.abs(delta)
    alpha = 1 - torch.exp(-dd)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd[..., :-1], dim=-1)], dim=-1))
    return alpha * trans, alpha, trans


# ----------------------------------------------------------------------------
# Density MLP (proposal levels 0-1, NeRF level 2)
# ----------------------------------------------------------------------------
def density_mlp(weights, cfg, level, rays, means, want_grad_normals=True):
    """BaseDensityMLP.__call__ -> predict_density_normals (geometry.py:381-586)."""
    gcfg = cfg.proposal_grids[level]
    base = f"Cache/Sampler/MLP_{level}"
    last = level == cfg.num_levels - 1
    need_grad = last and want_grad_normals

    m = means.detach().clone().requires_grad_(need_grad)
    warped = mathx.contract_radius(m, cfg.contract_radius)
    x = hashgrid_ref.hash_encoding(weights, f"{P}{base}/density_grid", gcfg, warped)
    # run_network (geometry.py:155-168): depth 2, width 64, no skip (skip_layer=4).
    h = torch.relu(dense(weights, f"{base}/density_layers_0", x))
    h = torch.relu(dense(weights, f"{base}/density_layers_1", h))
    raw = dense(weights, f"{base}/output_density_layer", h)[..., 0]

    out = {}
    if need_grad:
        (grad,) = torch.autograd.grad(raw.sum(), m)
        out["normals"] = mathx.nan_to_num(-mathx.l2_normalize(grad))          # geometry.py:460
    raw, h, warped = raw.detach(), h.detach(), warped.detach()

    # convert_raw_density (geometry.py:318-341)
    density = mathx.safe_exp(raw + cfg.density_bias)
    valid = ((warped > -gcfg.bbox) & (warped < gcfg.bbox)).all(dim=-1)
    density = torch.where(valid, density, torch.zeros_like(density))
    out["density"] = density
    out["feature"] = h
    if last:
        grad_pred = dense(weights, f"{base}/pred_normals_layer", h)
        out["normals_pred"] = mathx.nan_to_num(-mathx.l2_normalize(grad_pred))  # geometry.py:467-471
        out["normals_to_use"] = out["normals_pred"]                              # geometry.py:479
    out["ray_dists"] = torch.linalg.norm(rays["origins"][..., None, :] - means, dim=-1, keepdim=True)
    out["light_dists"] = torch.linalg.norm(rays["lights"][..., None, :] - means, dim=-1, keepdim=True)
    return out


# ----------------------------------------------------------------------------
# Proposal sampler
# ----------------------------------------------------------------------------
def secondary_near(cfg, rays):
    """sampling.py:182-205."""
    near, far = rays["near"], rays["far"]
    if rays.get("normals") is not None:
        dotprod = mathx.dot(rays["viewdirs"], rays["normals"])
        off = torch.minimum(torch.maximum(cfg.shadow_normal_eps_dot_min / torch.clamp(dotprod, min=1e-5), near), far)
        off = torch.where(dotprod > 0, off, near)
        near = torch.maximum(near, off.reshape(near.shape))
        near = torch.minimum(torch.maximum(near, torch.full_like(near, 1e-5)), far - 1e-5)
    return near


def make_s_to_t(cfg, near, far, use_raydist_fn):
    """coord.construct_ray_warps (coord.py:223-260) as wired at sampling.py:236-253."""
    if not use_raydist_fn:
        return lambda s: s * far + (1 - s) * near
    p, pm = cfg.raydist_p, cfg.raydist_premult
    s_near = mathx.power_ladder(near, p, pm)
    s_far = mathx.power_ladder(far, p, pm)
    return lambda s: mathx.inv_power_ladder(s * s_far + (1 - s) * s_near, p, pm)


def proposal_sampler(weights, cfg, rays, jitters, is_secondary=False, use_raydist_fn=False,
                     want_grad_normals=True):
    """ProposalVolumeSampler.__call__ (sampling.py:142-649).

    jitters: None (rng=None branch) or a list of per-level [R, 1] U[0,1) tensors.
    Returns the list of per-level result dicts (ray_history).
    """
    rays = dict(rays)
    if is_secondary:
        rays["near"] = secondary_near(cfg, rays)
    near, far = rays["near"], rays["far"]
    s_to_t = make_s_to_t(cfg, near, far, use_raydist_fn)
    dt = rays["origins"].dtype
    sdist = torch.cat([torch.zeros_like(near), torch.ones_like(far)], dim=-1)
    resample_w = torch.ones_like(near)
    history = []
    for lvl, (_, _, n) in enumerate(cfg.sampling_strategy):
        logits = cfg.anneal * mathx.safe_log(resample_w + cfg.resample_padding)   # sampling.py:339
        jit = None if jitters is None else jitters[lvl]
        sdist = stepfun_ref.sample_intervals(jit, sdist, logits, n, domain=(0.0, 1.0))
        tdist = s_to_t(sdist)
        means = cast_ray_means(tdist, rays["origins"], rays["directions"])
        res = density_mlp(weights, cfg, lvl, rays, means, want_grad_normals)
        # rectified normals (sampling.py:519-526)
        for k in [k for k in res if k.startswith("normals")]:
            pdot = (res[k] * rays["viewdirs"][..., None, :]).sum(-1, keepdim=True)
            res[k + "_rectified"] = res[k] * torch.where(pdot > 0, -1.0, 1.0).to(dt)
        w, a, tr = compute_alpha_weights(res["density"], tdist, rays["directions"])
        resample_w = w
        res.update(points=means, means=means, tdist=tdist, sdist=sdist, weights=w, alphas=a, trans=tr,
                   lossmult=rays["lossmult"])
        history.append(res)
    return history


# ----------------------------------------------------------------------------
# Categorical resampling (Model.maybe_resample)
# ----------------------------------------------------------------------------
PER_SAMPLE_KEYS = ("density", "feature", "normals", "normals_pred", "normals_to_use", "normals_rectified",
                   "normals_pred_rectified", "normals_to_use_rectified", "ray_dists", "light_dists",
                   "points", "means", "weights", "alphas", "trans")


def maybe_resample(cfg, sampler_results, resample, gumbel=None, inds=None, logits_mult=1.0):
    """models.py:193-292.  gumbel: [R, S] standard Gumbel noise standing in for the
    jax.random.categorical draw (argmax(logits + g)); ignored when `inds` is given."""
    res = dict(sampler_results)
    if not resample:
        res["weights_no_filter"] = res["weights"]
        return res, None
    w = sampler_results["weights"]
    logits = mathx.safe_log(w + 0.0) * logits_mult
    probs = torch.softmax(logits, dim=-1)
    n = cfg.num_resample
    if inds is None:
        assert n == 1, "oracle implements num_resample == 1 (models.py:118)"
        inds = torch.argmax(logits + gumbel.to(logits.dtype), dim=-1, keepdim=True)
    out = {}
    for k, v in sampler_results.items():
        if k in ("tdist", "sdist", "lossmult"):
            out[k] = v
        elif v.dim() == w.dim():
            out[k] = torch.gather(v, -1, inds)
        else:
            out[k] = torch.gather(v, -2, inds[..., None].expand(inds.shape + (v.shape[-1],)))
    out["weights_no_filter"] = sampler_results["weights"]
    fprobs = torch.gather(probs, -1, inds)
    out["weights"] = out["weights"] / (n * fprobs + 1e-8)
    return out, inds


# ----------------------------------------------------------------------------
# Surface-light-field style MLP (cache SurfaceLightField, cache EnvMap, model EnvMap)
# ----------------------------------------------------------------------------
def slf_trunk(weights, path, x):
    """run_surface_lightfield_network (surface_light_field.py:480-499):
    depth 4 (layer_0..2, layer_bottleneck), input skip after i == 2."""
    inp = x
    x = torch.relu(dense(weights, f"{path}/layer_0", x))
    x = torch.relu(dense(weights, f"{path}/layer_1", x))
    x = torch.relu(dense(weights, f"{path}/layer_2", x))
    x = torch.cat([x, inp], dim=-1)
    return torch.relu(dense(weights, f"{path}/layer_bottleneck", x))


def cache_slf_ambient(weights, cfg, bottleneck, refdirs, roughness):
    """Cache/Shader/SurfaceLightField as configured (surface_light_field.py:845-1069):
    x = shader_bottleneck ++ IDE_5(refdirs, roughness); ambient head."""
    x = torch.cat([bottleneck, mathx.ide(refdirs, roughness, cfg.slf_deg_view)], dim=-1)
    x = slf_trunk(weights, "Cache/Shader/SurfaceLightField", x)
    amb = mathx.softplus(dense(weights, "Cache/Shader/SurfaceLightField/output_ambient_rgb_layer", x)
                         + cfg.slf_ambient_bias)
    return torch.clamp(amb, min=0.0)


def cache_env_ambient(weights, cfg, refdirs, roughness):
    """Cache/Shader/EnvMap (IDE_4 only) -- multiplied by an exact 0 downstream (nerf.py:1032-1037)."""
    x = mathx.ide(refdirs, roughness, cfg.cache_env_deg_view)
    x = slf_trunk(weights, "Cache/Shader/EnvMap", x)
    amb = mathx.softplus(dense(weights, "Cache/Shader/EnvMap/output_ambient_rgb_layer", x) + cfg.slf_ambient_bias)
    return torch.clamp(amb, min=0.0)


def model_env_map_rgb(weights, cfg, viewdirs):
    """Cache/EnvMap queried by Model._handle_env_map (models.py:360-421): x = pos_enc(dir, 0, 4),
    rgb = softplus(raw_rgba[:3] + rgb_bias), clip(0, inf) (surface_light_field.py:1037-1058)."""
    x = mathx.pos_enc(viewdirs, 0, cfg.env_deg_view, append_identity=True)
    x = slf_trunk(weights, "Cache/EnvMap", x)
    raw = dense(weights, "Cache/EnvMap/output_rgba_layer", x)
    return torch.clamp(mathx.softplus(raw[..., :-1] + cfg.env_rgb_bias), min=0.0)


# ----------------------------------------------------------------------------
# Cache shader (NeRFMLP, passive branch)
# ----------------------------------------------------------------------------
def cache_shader(weights, cfg, rays, sres, exec_dead_envmap=False):
    """BaseShader.__call__ -> predict_appearance -> _predict_appearance_passive
    (shading.py:276-339, nerf.py:561-689, 940-1090)."""
    means = sres["means"]
    viewdirs = rays["viewdirs"]
    app = hashgrid_ref.hash_encoding(weights, f"{P}Cache/Shader/appearance_grid", cfg.appearance_grid,
                                     mathx.contract_radius(means, cfg.contract_radius))
    feature = torch.cat([sres["feature"], app], dim=-1)                    # shading.py:156-220
    bottleneck = dense(weights, "Cache/Shader/bottleneck_layer", feature)  # nerf.py:394-396
    roughness = mathx.softplus(dense(weights, "Cache/Shader/roughness_layer", feature) + cfg.roughness_bias)
    normals = sres["normals_to_use"]

    ambient_diffuse = torch.clamp(
        mathx.softplus(dense(weights, "Cache/Shader/ambient_irradiance_layer", feature)
                       + cfg.ambient_irradiance_bias), 0.0, cfg.rgb_max)
    tint = mathx.sigmoid(dense(weights, "Cache/Shader/tint_layer", feature))
    # get_integrated_brdf (nerf.py:461-482)
    dotprod = mathx.dot(normals, -viewdirs[..., None, :])
    x = torch.cat([bottleneck, dotprod], dim=-1)
    x = torch.relu(dense(weights, "Cache/Shader/integrated_brdf_layers_0", x))
    x = torch.relu(dense(weights, "Cache/Shader/integrated_brdf_layers_1", x))
    ibrdf = mathx.sigmoid(dense(weights, "Cache/Shader/output_integrated_brdf_layer", x) + pymath.log(3.0))

    refdirs = mathx.reflect(-viewdirs[..., None, :], normals)              # nerf.py:1344-1358
    if exec_dead_envmap:
        env_rgb = cache_env_ambient(weights, cfg, refdirs, roughness)
    else:
        env_rgb = torch.zeros_like(ambient_diffuse)
    indirect_diffuse = torch.clamp(
        mathx.softplus(dense(weights, "Cache/Shader/irradiance_layer", feature) + cfg.irradiance_bias),
        0.0, cfg.rgb_max)
    ref_rgb = cache_slf_ambient(weights, cfg, bottleneck, refdirs, roughness)
    ref_acc = torch.ones_like(ref_rgb[..., :1])                            # incoming_weights == 1
    ambient_specular = torch.clamp(tint * ibrdf * (env_rgb * (1.0 - ref_acc)), 0.0, cfg.rgb_max)
    indirect_specular = torch.clamp(tint * ibrdf * (ref_rgb * ref_acc), 0.0, cfg.rgb_max)

    ambient = ambient_diffuse + ambient_specular
    indirect = indirect_diffuse + indirect_specular
    rgb = ambient + indirect
    z = torch.zeros_like(rgb)
    out = dict(
        rgb=rgb, diffuse_rgb=ambient_diffuse + indirect_diffuse, specular_rgb=ambient_specular + indirect_specular,
        ambient_rgb=ambient, indirect_rgb=indirect, albedo_rgb=tint, occ=z, indirect_occ=ref_acc * torch.ones_like(rgb),
        direct_rgb=ambient, indirect_diffuse_rgb=indirect_diffuse, direct_diffuse_rgb=ambient_diffuse,
        direct_specular_rgb=ambient_specular, indirect_specular_rgb=indirect_specular,
        ambient_diffuse_rgb=ambient_diffuse, ambient_specular_rgb=ambient_specular,
        n_dot_l_rgb=z, light_radiance_rgb=z, irradiance_rgb=z,
        ray_dists=torch.linalg.norm(rays["origins"][..., None, :] - means, dim=-1, keepdim=True),
        roughness=roughness,
    )
    # shading.py:336-339: shader results + every sampler key not already present.
    for k, v in sres.items():
        out.setdefault(k, v)
    return out


# ----------------------------------------------------------------------------
# Volume integrator
# ----------------------------------------------------------------------------
EXTRAS_ALWAYS = (  # integration.py:199-231
    "diffuse_rgb", "specular_rgb", "occ", "indirect_occ", "direct_rgb", "indirect_rgb", "ambient_rgb",
    "irradiance_rgb", "light_radiance_rgb", "n_dot_l_rgb", "albedo_rgb", "direct_diffuse_rgb",
    "direct_specular_rgb", "indirect_diffuse_rgb", "indirect_specular_rgb", "ambient_diffuse_rgb",
    "ambient_specular_rgb", "means", "normals", "normals_pred", "normals_to_use", "light_dists", "ray_dists",
)


def volume_integrate(cfg, shader_results, bg):
    """VolumeIntegrator.__call__ (integration.py:112-289) + render.volumetric_rendering
    (render.py:172-247), compute_extras=False, equal bg range."""
    eps = mathx.EPS
    w = shader_results["weights"]
    wnf = shader_results["weights_no_filter"]
    tdist = shader_results["tdist"]
    acc = wnf.sum(-1)
    bg_w = torch.clamp(1 - acc[..., None], min=0.0)
    r = {}
    r["rgb"] = (w[..., None] * shader_results["rgb"]).sum(-2) + bg_w * bg
    r["acc"] = acc
    wnf_norm = wnf / torch.clamp(acc[..., None], min=eps)
    for k in EXTRAS_ALWAYS:
        v = shader_results.get(k)
        if v is not None:
            r[k] = (w[..., None] * v).sum(-2)
    t_mids = 0.5 * (tdist[..., :-1] + tdist[..., 1:])
    expect = (wnf * torch.log(t_mids)).sum(-1) / torch.clamp(acc, min=eps)
    dm = torch.exp(expect)
    fi = torch.finfo(dm.dtype)
    # render.py:233-237 / :308 write jnp.nan_to_num(x, jnp.inf): the 2nd positional parameter of jax 0.4.16's
    # nan_to_num(x, copy=True, nan=0.0, posinf=None, neginf=None) is `copy` -> nan stays at its default 0.0
    dm = torch.nan_to_num(dm, nan=0.0, posinf=fi.max, neginf=fi.min)
    r["distance_mean"] = torch.minimum(torch.maximum(dm, tdist[..., 0]), tdist[..., -1])
    pct = stepfun_ref.weighted_percentile(tdist, wnf_norm, cfg.percentiles)
    for i, p in enumerate(cfg.percentiles):
        name = "median" if p == 50 else "percentile_" + str(int(p))
        r["distance_" + name] = pct[..., i]
    return r


# ----------------------------------------------------------------------------
# BaseNeRFModel.__call__
# ----------------------------------------------------------------------------
def cache_forward(weights, cfg, rays: Dict[str, torch.Tensor], jitters=None, is_secondary=False,
                  resample=False, gumbel=None, inds=None, use_env_map=True, want_grad_normals=True,
                  exec_dead_envmap=False):
    """BaseNeRFModel.__call__ (models.py:657-774).

    Primary rays: linear t, bg = 1, no resampling (NeRFModel.resample_render=False).
    Secondary rays: far=min(far, env_map_distance), power-ladder distances, bg = 0, resample to
    one sample, model-level EnvMap composited with (1 - acc).  At render time the reference
    replaces the sampler rng of secondary rays by the constant PRNGKey(0) (sampling.py:170-179),
    i.e. it still takes the jittered branch of stepfun.sample; threefry is not reproduced here,
    so `jitters` stands in for that stream (jitters=None selects the rng=None linspace branch).
    """
    rays = dict(rays)
    do_resample = bool(resample) or is_secondary                # models.py:156-167 (resample_secondary=True)
    if is_secondary:
        rays["far"] = torch.clamp(rays["far"], max=cfg.env_map_distance)     # models.py:670-673
    bg = 0.0 if is_secondary else cfg.bg_intensity                            # models.py:183-191
    use_raydist_fn = is_secondary                                             # use_raydist_for_secondary_only
    history = proposal_sampler(weights, cfg, rays, jitters, is_secondary, use_raydist_fn, want_grad_normals)
    filtered, inds = maybe_resample(cfg, history[-1], do_resample, gumbel, inds)
    shader_results = cache_shader(weights, cfg, rays, filtered, exec_dead_envmap)
    integ = volume_integrate(cfg, shader_results, bg)
    if is_secondary:                                                          # models.py:309-460
        for k in list(integ.keys()):
            if "rgb" in k or "acc" in k:
                integ[k + "_no_stopgrad"] = integ[k].clone()
        if use_env_map:
            env = model_env_map_rgb(weights, cfg, rays["viewdirs"])
            integ["rgb"] = integ["rgb"] + env * (1.0 - integ["acc"][..., None])
            integ["rgb_no_stopgrad"] = integ["rgb_no_stopgrad"] + env * (1.0 - integ["acc"][..., None])
            integ["env_map_rgb"] = env
    return {"sampler": history, "filtered_sampler_inds": inds, "shader": shader_results,
            "geometry": history[-1], "integrator": integ, "render": integ}


FINAL_INTEGRATOR_KEYS = (  # models.py:2087-2111
    "rgb", "normals", "normals_pred", "incoming_rgb", "env_map_rgb", "incoming_s_dist", "diffuse_rgb",
    "specular_rgb", "occ", "indirect_occ", "direct_rgb", "indirect_rgb", "ambient_rgb", "irradiance_rgb",
    "light_radiance_rgb", "n_dot_l_rgb", "albedo_rgb", "direct_diffuse_rgb", "direct_specular_rgb",
    "indirect_diffuse_rgb", "indirect_specular_rgb", "ambient_diffuse_rgb", "ambient_specular_rgb",
)


def material_model_cache_only(weights, cfg, rays, jitters=None, **kw):
    """BaseMaterialModel.__call__ with use_material=False (models.py:1144-1254, 2065-2171):
    the `render` dict of the cache-only stage."""
    out = cache_forward(weights, cfg, rays, jitters, **kw)
    render = dict(out["integrator"])
    for k in FINAL_INTEGRATOR_KEYS:
        if k in out["integrator"]:
            render["cache_" + k] = out["integrator"][k]
    ones = torch.ones_like(render["rgb"][..., :1])
    render["vignette"] = ones
    render["lossmult"] = rays["lossmult"] * torch.ones_like(render["rgb"])    # models.py:2055-2063
    out["render"] = render
    return out
