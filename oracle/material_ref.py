"""Material pass: secondary-ray BRDF integration against the radiance cache (oracle).

Restates, for the resolved hotdog `material_light_from_scratch(_resample)` stage at render time
(SURVEY.md §8a rows a19-a23, §3.3):
  MaterialMLP._predict_material_and_feature / _get_microfacet_material  internal/material.py:2073-2123, 1290-1322, 957-1023
  LightMLP.predict_lighting / get_vmfs                                  internal/light_sampler.py:135-214
  render_utils.get_rotation_matrix, Cosine/Microfacet/Light samplers,
    importance_sample_rays, get_secondary_rays, sample_vmf, eval_vmf     internal/inverse_render/render_utils.py:145-168,
                                                                          417-546, 722-1056, 1335-1490
  render_utils.get_lobe, integrate_reflect_rays                         internal/inverse_render/render_utils.py:566-695, 1102-1193
  MaterialMLP.get_outgoing_radiance(_helper), _make_radiance_cache_fn,
    _make_env_map_fn, integration strategy                              internal/material.py:1352-1565, 1684-1864, 2174-2314, 2705-2808
  BaseMaterialModel._get_material_samples / _handle_material_pass /
    _handle_brdf_pass / _finalize_outputs                               internal/models.py:1398-1694, 1845-1912, 2074-2171

Every random quantity of the reference (jax.random) is an explicit input here, see `draw_randoms`.
See oracle/__init__.py for the usage rules (test infrastructure only, parity unpinned).
"""
from __future__ import annotations

import math as pymath

import numpy as np
import torch

from . import cache_ref, hashgrid_ref, mathx

P = "params/"
DENOM_EPS = 1e-5           # render_utils.DENOMINATOR_EPS
EPS = mathx.EPS


def ir_normalize(v):
    """inverse_render/math.py:normalize: v / sqrt(1e-10 + |v|^2)."""
    return v / torch.sqrt(1e-10 + (v * v).sum(-1, keepdim=True))


def ir_reflect(w, v):
    """inverse_render/math.py:reflect: 2 (v.w) v - w."""
    return 2.0 * (v * w).sum(-1, keepdim=True) * v - w


# ----------------------------------------------------------------------------
# Material and light heads
# ----------------------------------------------------------------------------
def material_mlp(weights, cfg, means):
    """feature = Dense(32->128)(material_grid(contract(x))), brdf = Dense(128->10), microfacet material."""
    g = hashgrid_ref.hash_encoding(weights, f"{P}MaterialShader/material_grid", cfg.material_grid,
                                   mathx.contract_radius(means, cfg.contract_radius))
    feat = cache_ref.dense(weights, "MaterialShader/bottleneck_layer", g)
    b = cache_ref.dense(weights, "MaterialShader/pred_brdf_layer", feat)
    r0 = cfg.min_roughness ** 2
    return dict(
        albedo=mathx.sigmoid(b[..., 0:3] - 1.0),
        specular_albedo=mathx.sigmoid(b[..., 5:6] - 1.0),
        roughness=mathx.sigmoid(b[..., 6:7] - 1.0) * (1.0 - r0) + r0,
        F_0=torch.full_like(b[..., 9:10], cfg.default_F_0),
        metalness=mathx.sigmoid(b[..., 8:9] + 0.0),
        diffuseness=torch.zeros_like(b[..., 3:4]),
        mirrorness=torch.zeros_like(b[..., 4:5]),
    )


def light_vmfs(weights, cfg, means, vmf_noise):
    """LightMLP: light_grid -> Dense64-ReLU x2 -> Dense(640) -> 128 x (mean, kappa, logit).
    vmf_noise: [..., 128, 3] standard normal standing in for jax.random.normal(PRNGKey(1))."""
    g = hashgrid_ref.hash_encoding(weights, f"{P}LightSampler/light_grid", cfg.light_grid,
                                   mathx.contract_radius(means, cfg.contract_radius))
    x = torch.relu(cache_ref.dense(weights, "LightSampler/layers_0", g))
    x = torch.relu(cache_ref.dense(weights, "LightSampler/layers_1", x))
    p = cache_ref.dense(weights, "LightSampler/output_layer", x).reshape(means.shape[:-1] + (cfg.num_vmf, 5))
    means_random = vmf_noise.to(p.dtype) * cfg.vmf_scale / 2.0
    vm = p[..., 0:3] * cfg.vmf_scale + 0.0 + means_random - means[..., None, :]
    kap = torch.clamp(mathx.softplus(p[..., 3:4] + 1.0), max=50.0)
    lg = torch.clamp(p[..., 4:5] + 1.0, min=-50.0)
    return dict(vmf_means=vm, vmf_kappas=kap, vmf_logits=lg)


# ----------------------------------------------------------------------------
# Tangent frames and samplers (local frame: z = normal)
# ----------------------------------------------------------------------------
def rotation_matrix(normal):
    """render_utils.get_rotation_matrix (y_up=False): columns (new_x, new_y, new_z)."""
    z = torch.tensor([0.0, 0.0, 1.0], dtype=normal.dtype)
    y = torch.tensor([0.0, 1.0, 0.0], dtype=normal.dtype)
    up = torch.where(torch.abs(normal[..., 2:3]) < 0.9, z, y)
    nx = torch.cross(up, normal, dim=-1)
    nx = nx / (torch.linalg.norm(nx, dim=-1, keepdim=True) + 1e-10)
    ny = torch.cross(normal, nx, dim=-1)
    ny = ny / (torch.linalg.norm(ny, dim=-1, keepdim=True) + 1e-10)
    return torch.stack([nx, ny, normal], dim=-1)


def global_to_local(d, R):
    return d[..., 0:1] * R[..., 0, :] + d[..., 1:2] * R[..., 1, :] + d[..., 2:3] * R[..., 2, :]


def local_to_global(d, R):
    return d[..., 0:1] * R[..., 0] + d[..., 1:2] * R[..., 1] + d[..., 2:3] * R[..., 2]


def ggx_d(costheta, a):
    return a ** 2 / torch.clamp(pymath.pi * (costheta ** 2 * (a ** 2 - 1.0) + 1.0) ** 2, min=EPS)


def cosine_sample(u1, u2):
    r = torch.sqrt(u1)
    phi = u2 * 2.0 * pymath.pi - pymath.pi
    x, y = r * torch.cos(phi), r * torch.sin(phi)
    z = torch.sqrt(torch.clamp(1.0 - x ** 2 - y ** 2, min=DENOM_EPS))
    return torch.stack([x, y, z], -1), torch.clamp(z / pymath.pi, min=0.0)


def cosine_pdf(wi):
    pdf = wi[..., 2] / pymath.pi
    return torch.clamp(torch.where(wi[..., 2] < 0, torch.zeros_like(pdf), pdf), min=0.0)


def microfacet_sample(u1, u2, wo, alpha):
    """MicrofacetSampler.sample_directions (sample_visible=False); alpha [..., K]."""
    tan2 = alpha ** 2 * u1 / torch.clamp(1.0 - u1, min=EPS)
    cost = 1.0 / torch.sqrt(torch.clamp(1.0 + tan2, min=EPS))
    sint = torch.sqrt(torch.clamp(1.0 - cost ** 2, min=DENOM_EPS))
    phi = u2 * 2.0 * pymath.pi - pymath.pi
    n = torch.stack([sint * torch.cos(phi), sint * torch.sin(phi), cost], -1)
    npdf = torch.clamp(ggx_d(cost, alpha) * torch.abs(cost), min=0.0)
    d = ir_reflect(wo, n)
    wn = (wo * n).sum(-1)
    pdf = npdf * (1.0 / torch.clamp(4.0 * wn, min=EPS))
    pdf = torch.where(wn <= 0.0, torch.zeros_like(pdf), pdf)
    return ir_normalize(d), torch.clamp(pdf, min=0.0)


def microfacet_pdf(wo, wi, alpha):
    n = ir_normalize(wo + wi)
    wn = (wo * n).sum(-1)
    pdf = ggx_d(n[..., 2], alpha) * torch.abs(n[..., 2]) * (1.0 / torch.clamp(4.0 * wn, min=EPS))
    return torch.clamp(torch.where(wn <= 0.0, torch.zeros_like(pdf), pdf), min=0.0)


def eval_vmf(x, means, kappa):
    """render_utils.eval_vmf with inverse_render.math.safe_exp (= exp(min(x, 80)))."""
    val = kappa * torch.exp(torch.clamp(kappa * (x * means).sum(-1), max=80.0)) / (4 * pymath.pi * torch.sinh(kappa))
    return torch.where(kappa <= EPS, torch.ones_like(val) / (4.0 * pymath.pi), val)


def light_pdf(wi, vmfs):
    """LightSampler.pdf: softmax(logits)-weighted vMF mixture at GLOBAL directions wi [N, K, 3]."""
    means = mathx.l2_normalize(vmfs["vmf_means"])          # [N, 128, 3]
    kap = vmfs["vmf_kappas"][..., 0]
    w = torch.softmax(vmfs["vmf_logits"][..., 0], dim=-1)
    pdf = (w[..., None, :] * eval_vmf(wi[..., None, :], means[..., None, :, :], kap[..., None, :])).sum(-1)
    return torch.clamp(pdf, min=0.0)


def light_sample(vmfs, lobe, v, tmp):
    """LightSampler.sample_directions -> sample_vmf: one lobe per point (`lobe` [N] stands in for
    jax.random.categorical over the logits), v [N, K, 2] ~ N(0,1), tmp [N, K] ~ U[0,1)."""
    means = mathx.l2_normalize(vmfs["vmf_means"])
    kap_all = vmfs["vmf_kappas"][..., 0]
    idx = lobe.reshape(-1, 1, 1).expand(-1, 1, 3)
    mean = torch.gather(means, -2, idx)[..., 0, :]
    kappa = torch.gather(kap_all, -1, lobe.reshape(-1, 1))[..., 0]
    t = mathx.l2_normalize(torch.stack([-mean[..., 1], mean[..., 0], torch.zeros_like(mean[..., 0])], -1))
    b = mathx.l2_normalize(torch.cross(mean, t, dim=-1))
    rot = torch.stack([t, b, mean], dim=-1)
    v = mathx.l2_normalize(v.to(mean.dtype))
    tmp = tmp.to(mean.dtype)
    w = 1.0 + (1.0 / torch.clamp(kappa[..., None], min=EPS)) * mathx.safe_log(
        tmp + (1.0 - tmp) * torch.exp(-2.0 * kappa[..., None]))
    s = torch.sqrt(torch.clamp(1.0 - w ** 2, 0.0, mathx.MAXV))        # math.safe_sqrt
    d = torch.stack([s * v[..., 0], s * v[..., 1], w], -1)
    dirs = (rot[..., None, :, :] @ d[..., None])[..., 0]
    return dirs, light_pdf(dirs, vmfs)


# ----------------------------------------------------------------------------
# Importance sampling of secondary directions (importance_sample_rays)
# ----------------------------------------------------------------------------
def sample_specular(global_view, normal, material, u1, u2):
    """samplers = [(microfacet, 1)]: no MIS, weight 1."""
    R = rotation_matrix(normal)
    lv = global_to_local(global_view, R)
    K = u1.shape[-1]
    lvk = lv[..., None, :].expand(-1, K, -1)
    ld, pdf = microfacet_sample(u1, u2, lvk, material["roughness"])   # roughness [N,1] broadcasts over K
    return dict(local_lightdirs=ld, local_viewdirs=lvk, global_lightdirs=local_to_global(ld, R[..., None, :, :]),
                pdf=pdf[..., None], weight=torch.ones_like(pdf)[..., None])


def sample_diffuse(global_view, normal, material, u1, u2, vmfs, lobe, v, tmp):
    """samplers = [(cosine, 1), (light, 1)], use_mis: power heuristic over both (render_utils.py:817-853)."""
    R = rotation_matrix(normal)
    lv = global_to_local(global_view, R)
    K = u1.shape[-1]
    lvk = lv[..., None, :].expand(-1, K, -1)
    Rk = R[..., None, :, :]
    out_dirs, out_pdf, out_w = [], [], []
    cos_d, cos_p = cosine_sample(u1, u2)
    lgt_gd, lgt_p = light_sample(vmfs, lobe, v, tmp)
    lgt_d = global_to_local(lgt_gd, Rk)
    for ld, pdf in ((cos_d, cos_p), (lgt_d, lgt_p)):
        gl = local_to_global(ld, Rk)
        denom = (cosine_pdf(ld) * 1) ** 2 + (light_pdf(gl, vmfs) * 1) ** 2
        pdf = torch.clamp(pdf, min=0.0)
        wgt = (1 * pdf) ** 2 / torch.clamp(denom, min=DENOM_EPS) * (2.0 / 1.0)
        out_dirs.append(ld); out_pdf.append(pdf); out_w.append(wgt)
    ld = torch.cat(out_dirs, -2)
    K2 = ld.shape[-2]
    return dict(local_lightdirs=ld, local_viewdirs=lv[..., None, :].expand(-1, K2, -1),
                global_lightdirs=local_to_global(ld, Rk), pdf=torch.cat(out_pdf, -1)[..., None],
                weight=torch.cat(out_w, -1)[..., None])


# ----------------------------------------------------------------------------
# BRDF lobes and the Monte-Carlo estimator
# ----------------------------------------------------------------------------
def get_lobe(wi, wo, material, kind):
    """render_utils.get_lobe in the local frame (normal = +z), brdf_correction = 1,
    use_specular_albedo = use_mirrorness = use_diffuseness = False."""
    albedo = material["albedo"][..., None, :]
    metal = material["metalness"][..., None, :]
    a = material["roughness"][..., None, :]
    F0 = albedo * metal + material["F_0"][..., None, :] * (1.0 - metal)
    h = ir_normalize(wi + wo)
    n_v = torch.clamp(wo[..., 2:3], min=0.0)
    n_l = torch.clamp(wi[..., 2:3], min=0.0)
    n_h = torch.clamp(h[..., 2:3], min=0.0)
    l_h = torch.clamp((wi * h).sum(-1, keepdim=True), min=0.0)
    D = ggx_d(n_h, a)
    F = F0 + (1.0 - F0) * torch.clamp(1.0 - l_h, 0.0, 1.0) ** 5
    k = a / 2
    G = (n_v / torch.clamp(n_v * (1.0 - k) + k, min=EPS)) * (n_l / torch.clamp(n_l * (1.0 - k) + k, min=EPS))
    ggx = D * F * G / torch.clamp(4.0 * n_v, min=EPS)
    lambert = n_l * albedo / pymath.pi
    if kind == "microfacet_specular":
        return ggx * 1.0 * torch.ones_like(metal)
    return lambert * 1.0 * (1.0 - metal)


def integrate_reflect_rays(kind, material, s, rgb_max):
    lobe = get_lobe(s["local_lightdirs"], s["local_viewdirs"], material, kind)
    denom = torch.clamp(s["pdf"], min=DENOM_EPS)
    w = torch.clamp(s["weight"], min=0.0)
    w = torch.where(s["local_lightdirs"][..., 2:] > 0.0, w, torch.zeros_like(w))
    rad = (torch.clamp(s["radiance_in"] * lobe, 0.0, rgb_max) * w / denom).mean(1)
    dl = torch.clamp(s["local_lightdirs"][..., 2:], min=0.0) / pymath.pi
    irr = (torch.clamp(s["radiance_in"] * dl, 0.0, rgb_max) * w / denom).mean(1)
    return dict(radiance_out=rad, indirect_occ=s["indirect_occ"].mean(1), irradiance=irr)


# ----------------------------------------------------------------------------
# Random inputs
# ----------------------------------------------------------------------------
def draw_randoms(cfg, n_rays, seed=0):
    """All random tensors of one material-stage forward (numpy, float32)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Ks = int(round(cfg.num_secondary_samples * (1.0 - cfg.diffuse_sample_fraction)))
    Kd = int(round(cfg.num_secondary_samples * cfg.diffuse_sample_fraction))
    kc = int(round(0.5 * Kd))
    S = cfg.sampling_strategy[-1][2]
    f = lambda *s: rng.uniform(size=s).astype(np.float32)
    return dict(
        jitter=[f(n_rays) for _ in range(3)], gumbel=rng.gumbel(size=(n_rays, S)).astype(np.float32),
        vmf_noise=rng.normal(size=(n_rays, cfg.num_vmf, 3)).astype(np.float32),
        spec_u1=f(n_rays, Ks), spec_u2=f(n_rays, Ks), cos_u1=f(n_rays, kc), cos_u2=f(n_rays, kc),
        vmf_lobe=rng.integers(0, cfg.num_vmf, size=(n_rays,)).astype(np.int32),
        vmf_v=rng.normal(size=(n_rays, Kd - kc, 2)).astype(np.float32), vmf_tmp=f(n_rays, Kd - kc),
        spec_jitter=[f(n_rays * Ks) for _ in range(3)], spec_gumbel=rng.gumbel(size=(n_rays * Ks, S)).astype(np.float32),
        diff_jitter=[f(n_rays * Kd) for _ in range(3)], diff_gumbel=rng.gumbel(size=(n_rays * Kd, S)).astype(np.float32),
    )


# ----------------------------------------------------------------------------
# The material stage
# ----------------------------------------------------------------------------
def _secondary_trace(weights, cfg, origins, dirs, lights, jitter, gumbel, inds=None):
    """_make_radiance_cache_fn: cache(is_secondary=True, resample=True, use_env_map=False), normals=None
    (MaterialMLP.shadow_eps_indirect=False), near = MaterialMLP.near_min, far = Config.secondary_far."""
    n = origins.shape[0]
    dt = origins.dtype
    rays = dict(origins=origins, directions=dirs, viewdirs=dirs, lights=lights,
                near=torch.full((n, 1), cfg.secondary_near, dtype=dt), far=torch.full((n, 1), cfg.secondary_far, dtype=dt),
                lossmult=torch.ones((n, 1), dtype=dt))
    out = cache_ref.cache_forward(weights, cfg, rays, [torch.as_tensor(j).reshape(-1, 1) for j in jitter], is_secondary=True,
                                  gumbel=None if gumbel is None else torch.as_tensor(gumbel),
                                  inds=None if inds is None else torch.as_tensor(inds).long().reshape(-1, 1),
                                  use_env_map=False, want_grad_normals=False)
    r = out["render"]
    rgb = torch.clamp(mathx.nan_to_num(r["rgb"]), min=0.0)
    return rgb, r["acc"], out["filtered_sampler_inds"]


def material_forward(weights, cfg, rays, rnd, want_grad_normals=False):
    """BaseMaterialModel.__call__ with use_material=True, use_light_sampler=True, MaterialModel.resample_render=True,
    passes ("cache", "light", "material"), train=False.  Returns {"render": ..., "cache": ..., "debug": ...}."""
    dt = rays["origins"].dtype
    T = lambda a: torch.as_tensor(a).to(dt) if not torch.is_tensor(a) or a.is_floating_point() else torch.as_tensor(a)
    # --- cache pass on the primary rays (all 32 samples shaded)
    cache = cache_ref.cache_forward(weights, cfg, rays, [T(j).reshape(-1, 1) for j in rnd["jitter"]],
                                    want_grad_normals=want_grad_normals)
    geo = cache["sampler"][-1]
    # --- _get_material_samples: categorical resample to one sample per ray
    # rnd["resample_inds"] / rnd["{spec,diff}_resample_inds"] (optional) hand the categorical picks over instead of
    # drawing them from the Gumbel noise (what rc_material_randoms.resample_inds / .sec_resample_inds do for the HIP path)
    pin = rnd.get("resample_inds")
    filt, inds = cache_ref.maybe_resample(cfg, geo, True, gumbel=None if rnd.get("gumbel") is None else T(rnd["gumbel"]),
                                          inds=None if pin is None else torch.as_tensor(pin).long().reshape(-1, 1))
    R = rays["origins"].shape[0]
    pts = filt["means"][:, 0]                       # [R, 3]
    nrm = filt["normals_to_use"][:, 0]
    view = rays["viewdirs"]
    cache_shader = cache_ref.cache_shader(weights, cfg, rays, filt)
    # --- light sampler, material
    vmfs = light_vmfs(weights, cfg, pts, T(rnd["vmf_noise"]))
    mat = material_mlp(weights, cfg, pts)
    gview = -view
    origins = pts + nrm * cfg.secondary_normal_eps
    spec = sample_specular(gview, nrm, mat, T(rnd["spec_u1"]), T(rnd["spec_u2"]))
    if rnd.get("vmf_lobe") is not None:
        lobe = torch.as_tensor(rnd["vmf_lobe"]).long()
    else:       # sample_vmf_vars (render_utils.py:1357-1372): categorical over the lobe logits = argmax(logits + Gumbel noise)
        lobe = torch.argmax(vmfs["vmf_logits"][..., 0] + T(rnd["vmf_lobe_gumbel"]), dim=-1)
    diff = sample_diffuse(gview, nrm, mat, T(rnd["cos_u1"]), T(rnd["cos_u2"]), vmfs, lobe, T(rnd["vmf_v"]), T(rnd["vmf_tmp"]))
    integ = {}
    dbg = {}
    for name, s, kind, jit, gum, sinds in (
            ("specular", spec, "microfacet_specular", rnd["spec_jitter"], rnd.get("spec_gumbel"), rnd.get("spec_resample_inds")),
            ("diffuse", diff, "microfacet_diffuse", rnd["diff_jitter"], rnd.get("diff_gumbel"), rnd.get("diff_resample_inds"))):
        K = s["local_lightdirs"].shape[1]
        s["weight"] = torch.where(s["local_lightdirs"][..., 2:] > 0.0, s["weight"], torch.zeros_like(s["weight"]))
        o = origins[:, None, :].expand(-1, K, -1).reshape(-1, 3)
        d = s["global_lightdirs"].reshape(-1, 3)
        lg = rays["lights"][:, None, :].expand(-1, K, -1).reshape(-1, 3)
        rgb, acc, sec_inds = _secondary_trace(weights, cfg, o, d, lg, [T(j) for j in jit], None if gum is None else T(gum), sinds)
        # indirect: radiance from the cache
        s_ind = dict(s, radiance_in=mathx.nan_to_num(rgb).reshape(R, K, 3), indirect_occ=acc.reshape(R, K, 1))
        integ["indirect_" + name] = integrate_reflect_rays(kind, mat, s_ind, cfg.rgb_max)
        # direct: learned env map along the same rays, attenuated by (1 - acc) (_make_env_map_fn)
        env = torch.clamp(cache_ref.model_env_map_rgb(weights, cfg, d), min=0.0) * (1.0 - acc[:, None])
        s_dir = dict(s, radiance_in=mathx.nan_to_num(env).reshape(R, K, 3), indirect_occ=acc.reshape(R, K, 1))
        integ["direct_" + name] = integrate_reflect_rays(kind, mat, s_dir, cfg.rgb_max)
        dbg[name] = dict(origins=o, dirs=d, pdf=s["pdf"], weight=s["weight"], local_lightdirs=s["local_lightdirs"],
                         local_viewdirs=s["local_viewdirs"],
                         rgb=rgb, acc=acc, env=env, inds=sec_inds[:, 0])
    # --- integration strategy (material.py:2705-2808)
    ro = lambda a, b: integ[a + "_" + b]["radiance_out"]
    sh = {}
    sh["rgb"] = ro("direct", "diffuse") + ro("direct", "specular") + ro("indirect", "diffuse") + ro("indirect", "specular")
    sh["direct_rgb"] = ro("direct", "diffuse") + ro("direct", "specular")
    sh["indirect_rgb"] = ro("indirect", "diffuse") + ro("indirect", "specular")
    sh["diffuse_rgb"] = ro("direct", "diffuse") + ro("indirect", "diffuse")
    sh["specular_rgb"] = ro("direct", "specular") + ro("indirect", "specular")
    sh["direct_diffuse_rgb"] = ro("direct", "diffuse") + 0.0      # + emission (zeros)
    sh["direct_specular_rgb"] = ro("direct", "specular")
    sh["indirect_diffuse_rgb"] = ro("indirect", "diffuse")
    sh["indirect_specular_rgb"] = ro("indirect", "specular")
    sh["indirect_occ"] = integ["indirect_specular"]["indirect_occ"] * 0.5
    sh["lighting_irradiance"] = (integ["direct_diffuse"]["irradiance"] + integ["indirect_diffuse"]["irradiance"]) * 0.5
    RENDERED_MATERIAL = ("albedo", "roughness", "F_0", "metalness", "diffuseness", "mirrorness")   # integration.py:151-160
    for k in RENDERED_MATERIAL:
        sh["material_" + k] = mat[k]
    # --- MaterialIntegrator over the ONE filtered sample (compute_extras=True, compute_distance=False)
    w = filt["weights"]                          # [R, 1]
    acc = filt["weights_no_filter"].sum(-1)
    bgw = torch.clamp(1 - acc[:, None], min=0.0)
    render = {"rgb": w * sh["rgb"] + bgw * cfg.bg_intensity, "acc": acc}
    for k, v in sh.items():
        if k != "rgb":
            render[k] = w * v
    render["occ"] = w * cache_shader["occ"][:, 0]                    # filtered_results_material["occ"]
    for k in ("means", "normals", "normals_pred", "normals_to_use"):
        if k in filt:
            render[k] = w * filt[k][:, 0]
    render["ray_dists"] = w * torch.linalg.norm(rays["origins"] - pts, dim=-1, keepdim=True)
    render["light_dists"] = w * torch.linalg.norm(rays["lights"] - pts, dim=-1, keepdim=True)
    # --- _handle_brdf_pass: material-only shader on all samples, composited with the unfiltered weights
    mat_all = material_mlp(weights, cfg, geo["means"])
    for k in RENDERED_MATERIAL:
        render["material_" + k] = (geo["weights"][..., None] * mat_all[k]).sum(-2)
    # --- distances from the cache integrator, cache_* aliases, constants (_finalize_outputs)
    ci = cache["integrator"]
    for k in ci:
        if "distance" in k:
            render[k] = ci[k]
    for k in cache_ref.FINAL_INTEGRATOR_KEYS:
        if k in ci:
            render["cache_" + k] = ci[k]
    render["material_rgb"] = render["rgb"]
    render["normals"] = ci["normals"] if "normals" in ci else None
    render["normals_pred"] = ci["normals_pred"]
    if render["normals"] is None:
        render.pop("normals")
    render["vignette"] = torch.ones_like(render["rgb"][..., :1])
    render["lossmult"] = torch.ones_like(render["rgb"][..., :1])     # models.py:2046-2053 (all-true mask)
    return {"render": render, "cache": cache, "inds": inds, "debug": dbg, "material": mat, "vmfs": vmfs,
            "shader": sh, "filtered": filt, "lobe": lobe}
