"""Python host layer: the reference's call signatures on top of the C ABI.

Mirrors, for the radiance-cache hot path only:
  * model.apply(variables, rng, rays, **render_kwargs) -> {"render": {...}, ...}
      BaseMaterialModel.__call__              internal/models.py:1144-1254
  * create_render_fn(model, dataset, mapping_fn) -> render_eval_pfn(variables, rng, train_frac, cameras, lights,
                                                                    rays, passes, resample) -> (renderings, rng)
      train_utils.create_render_fn            internal/train_utils.py:3742-3831
      (called as at engine/trainer.py:822-832; `bind_render_fn` builds the 4-argument closure the Trainer wraps
       around it, engine/trainer.py:821-832)
  * render_image(render_fn, rng, rays, config, passes, ...) -> (rendering, rng)
      models.render_image                     internal/models.py:2361-2525
  * utils.shard / utils.unshard               internal/utils.py:333-343

One process drives one GPU (torch.distributed rank); `n_local_devices` is therefore 1 and the
leading device axis of the reference's pmap outputs is kept with size 1 so callers that index
`v[0]` keep working.  Ray batches are sharded across ranks at image granularity by
`render_image_distributed`, with one all-gather of the consumed keys per image (RCCL over xGMI
via torch.distributed's "nccl" backend; "gloo" on CPU for the tests).

Randomness: `rng` may be
  * None                         -> the reference's rng=None deterministic branch,
  * a uint32[2] key (prng.PRNGKey)-> the tensors jax.random would draw from that key at the reference's
                                    random_split sites (prng.py; threefry2x32 pinned by published known answers,
                                    the call order restated from the source and not checkable without jax),
  * a numpy Generator / int seed -> per-level ray jitter (and Gumbel noise) drawn here,
  * a dict {"jitter": [...], "gumbel": ..., "resample_inds": ...} of explicit tensors.
"""
from __future__ import annotations

import time
from collections.abc import Mapping
from typing import Any, Dict, Optional, Tuple

import numpy as np

from . import prng, rc_ext
from .config import RenderConfig
from .rays import Pixels, Rays

# keys of _finalize_outputs (internal/models.py:2087-2111) that get a `cache_` alias
_FINAL_INTEGRATOR_KEYS = (
    "rgb", "normals", "normals_pred", "incoming_rgb", "env_map_rgb", "incoming_s_dist", "diffuse_rgb",
    "specular_rgb", "occ", "indirect_occ", "direct_rgb", "indirect_rgb", "ambient_rgb", "irradiance_rgb",
    "light_radiance_rgb", "n_dot_l_rgb", "albedo_rgb", "direct_diffuse_rgb", "direct_specular_rgb",
    "indirect_diffuse_rgb", "indirect_specular_rgb", "ambient_diffuse_rgb", "ambient_specular_rgb",
)
# device outputs the primary cache pass computes
_SECONDARY_DEVICE_KEYS = ("env_map_rgb", "rgb_no_env")
_CACHE_DEVICE_KEYS = ("rgb", "acc", "distance_mean", "distance_percentile_5", "distance_median",
                      "distance_percentile_95", "diffuse_rgb", "specular_rgb", "direct_rgb", "indirect_rgb",
                      "albedo_rgb", "indirect_diffuse_rgb", "indirect_specular_rgb", "indirect_occ", "means",
                      "normals", "normals_pred", "ray_dists", "light_dists")


def flatten_variables(variables: Dict[str, Any], prefix: str = "") -> Dict[str, Any]:
    """Flax variable tree {"params": {"Cache": {...}}} -> {"params/Cache/...": array}."""
    flat = {}
    for k, v in variables.items():
        name = f"{prefix}/{k}" if prefix else k
        if isinstance(v, dict):
            flat.update(flatten_variables(v, name))
        else:
            flat[name] = v
    return flat


def _leaf_refs(variables, out):
    """(container, key, leaf, version) of every leaf: what `Model.apply` re-checks to notice an updated tree."""
    for k, v in variables.items():
        if isinstance(v, dict):
            _leaf_refs(v, out)
        else:
            out.append((variables, k, v, getattr(v, "_version", None)))
    return out


def _draw_randoms(rng, n: int, cfg: RenderConfig, need_gumbel: bool):
    if rng is None:
        return None, None
    if isinstance(rng, dict):
        return rng, None
    if prng.is_key(rng):
        return prng.cache_pass_randoms(rng, n, [lvl[2] for lvl in cfg.sampling_strategy], need_gumbel), None
    if isinstance(rng, (int, np.integer)):
        rng = np.random.Generator(np.random.PCG64(int(rng)))
    out = {"jitter": [rng.uniform(size=(n,)).astype(np.float32) for _ in range(cfg.num_levels)]}
    if need_gumbel:
        out["gumbel"] = rng.gumbel(size=(n, cfg.sampling_strategy[-1][2])).astype(np.float32)
    return out, rng


class RenderDict(Mapping):
    """The `render` dict of one ray batch (keys of _finalize_outputs, internal/models.py:2074-2171) over ONE flat
    device allocation: the kernel writes every output slot into `flat`; a key's tensor ([n, 3], [n, 1] or [n], plus
    `lead` leading axes of size 1) is a view made when the key is read.  Aliases (`cache_<k>`, `ambient_rgb`, ...)
    share a region, exact constants (`occ`, `vignette`, ...) share one cached tensor per batch size.  The host loop
    of render_image never touches the views: it keeps `flat` per chunk and unpacks the whole image once."""

    __slots__ = ("flat", "layout", "consts", "extras", "lead", "_views")

    def __init__(self, flat, layout, consts, extras=(), lead=0):
        self.flat, self.layout, self.consts, self.extras, self.lead = flat, layout, consts, extras, lead
        self._views = {}

    def with_lead(self, lead: int) -> "RenderDict":
        return RenderDict(self.flat, self.layout, self.consts, self.extras, lead)

    def __getitem__(self, key):
        v = self._views.get(key)
        if v is None:
            kind, a, shape = self.layout[key]
            if kind == "flat":
                cnt = 1
                for d in shape:
                    cnt *= d
                v = self.flat[a: a + cnt].view(shape)
            elif kind == "const":
                v = self.consts(a, shape)
            else:
                v = self.extras[a]
            for _ in range(self.lead):
                v = v[None]
            self._views[key] = v
        return v

    def __iter__(self):
        return iter(self.layout)

    def __len__(self):
        return len(self.layout)


def _cache_layout(offs, n: int, secondary: bool, lossmult_extra: bool):
    """Key -> ("flat", offset, shape) | ("const", value, shape) | ("extra", index, shape): the integrator keys plus the
    aliases / exact constants of the configured passive shader and of _finalize_outputs (internal/nerf.py:1044-1084,
    internal/models.py:2074-2171)."""
    lay = {}
    for k in _CACHE_DEVICE_KEYS + (("env_map_rgb",) if secondary else ()):
        off, shape = offs[k]
        lay[k] = ("flat", off, (n, 1) if k in ("ray_dists", "light_dists") else shape)
    for k in ("ambient_rgb", "direct_diffuse_rgb", "ambient_diffuse_rgb"):
        lay[k] = lay["direct_rgb"]
    for k in ("occ", "irradiance_rgb", "light_radiance_rgb", "n_dot_l_rgb", "direct_specular_rgb", "ambient_specular_rgb"):
        lay[k] = ("const", 0.0, (n, 3))
    lay["normals_to_use"] = lay["normals_pred"]
    for k in _FINAL_INTEGRATOR_KEYS:
        if k in lay:
            lay["cache_" + k] = lay[k]
    lay["vignette"] = ("const", 1.0, (n, 1))
    lay["lossmult"] = ("extra", 0, (n, 3)) if lossmult_extra else ("const", 1.0, (n, 3))
    if secondary:
        # Model._handle_secondary (internal/models.py:309-460): *_no_stopgrad copies (the env composite runs on the device)
        lay["rgb_no_stopgrad"] = lay["rgb"]
        lay["acc_no_stopgrad"] = lay["acc"]
    return lay


class Model:
    """Cache stage of MaterialModel (use_material=False): `apply` keeps the reference signature."""

    def __init__(self, config: Optional[RenderConfig] = None, device: int = 0):
        self.config = config or RenderConfig()
        self.device = device
        self.rc = rc_ext.RadianceCache(self.config, device)   # raises if librc_hip.so is missing
        self._variables_ref = None
        self._variables_checked = False       # render_image's device loop: the tree was checked for this image already
        self._out_arena = None                # render_image's device loop: (arena [n_chunks, floats], next row)
        self._leaves = []
        self._plans = {}
        self._const_cache = {}

    # -- weights ------------------------------------------------------------------------------------
    def load_variables(self, variables: Dict[str, Any]):
        """Upload a parameter tree (nested Flax tree or the flat {"params/...": array} dict).  A tree replicated for
        pmap (flax.jax_utils.replicate: every leaf with a leading [n_local_devices = 1] axis, engine/trainer.py:647-670)
        is accepted as it is."""
        nested = any(isinstance(v, dict) for v in variables.values())
        flat = flatten_variables(variables) if nested else dict(variables)
        bias = next((v for k, v in flat.items() if k.endswith("/bias")), None)
        if bias is not None and len(np.shape(bias)) == 2:          # replicated: strip the device axis
            flat = {k: v[0] for k, v in flat.items()}
        self.rc.load_weights(flat)
        self._variables_ref = variables
        self._leaves = _leaf_refs(variables, [])                   # holds the leaves: their ids cannot be recycled

    def _ensure_variables(self, variables):
        """Re-upload when `variables` is another tree than the loaded one, when a leaf of the same tree was replaced
        (functional update in place of the container) or when a torch leaf was written in place (its `_version` moved).
        In-place writes into numpy leaves cannot be seen: call load_variables after those."""
        if variables is None:
            return
        if self._variables_checked and variables is self._variables_ref:
            return                                                 # same tree, verified once for the image being rendered
        if variables is self._variables_ref:
            for cont, key, leaf, ver in self._leaves:
                cur = cont.get(key)
                if cur is not leaf or (ver is not None and cur._version != ver):
                    break
            else:
                return
        self.load_variables(variables)

    # -- per-batch-size plans and constants -----------------------------------------------------------
    def _consts(self, value: float, shape):
        import torch
        key = (value, shape)
        t = self._const_cache.get(key)
        if t is None:
            if len(self._const_cache) > 64:
                self._const_cache.clear()
            t = torch.full(shape, value, dtype=torch.float32, device=f"cuda:{self.device}")
            self._const_cache[key] = t
        return t

    def _plan(self, n: int, secondary: bool, lossmult_extra: bool):
        key = (n, secondary, lossmult_extra)
        p = self._plans.get(key)
        if p is None:
            if len(self._plans) > 64:
                self._plans.clear()
            names = _CACHE_DEVICE_KEYS + (_SECONDARY_DEVICE_KEYS if secondary else ())
            rc_plan = self.rc.output_plan(names, n)
            p = (rc_plan, _cache_layout(rc_plan[1], n, secondary, lossmult_extra))
            self._plans[key] = p
        return p

    # -- model.apply --------------------------------------------------------------------------------------
    def apply(self, variables, rng, rays: Rays, *, train_frac: float = 1.0, train: bool = False,
              passes: Tuple[str, ...] = ("cache",), compute_extras: bool = False, is_secondary: bool = False,
              resample: Any = None, sampling_strategy=None, **unused_render_kwargs):
        """model.apply(variables, rng, rays, ...) (internal/models.py:1144-1254).

        `variables` may be None once `load_variables` was called (they live on the device).
        Only the inference branch exists here: train must be False and train_frac 1.0.
        """
        if train or train_frac != 1.0:
            raise NotImplementedError("only the render-time path (train=False, train_frac=1.0) is accelerated")
        if sampling_strategy is not None and tuple(sampling_strategy) != tuple(self.config.sampling_strategy):
            raise NotImplementedError("sampling_strategy is fixed when the handle is created")
        if "material" in passes:
            return self._apply_material(variables, rng, rays)
        if self.config.transient is not None:
            return self._apply_transient(variables, rng, rays, is_secondary or "is_secondary" in passes, resample)
        self._ensure_variables(variables)
        fields = rays.hot_fields() if isinstance(rays, Rays) else rays
        near = fields["near"]
        n = near.numel() if hasattr(near, "numel") else int(np.prod(np.shape(near)))
        mask = rc_ext.RC_PASS_CACHE
        if is_secondary or "is_secondary" in passes:
            mask |= rc_ext.RC_PASS_SECONDARY
        if resample:
            mask |= rc_ext.RC_PASS_RESAMPLE
        need_gumbel = bool(mask & (rc_ext.RC_PASS_SECONDARY | rc_ext.RC_PASS_RESAMPLE))
        randoms, _ = _draw_randoms(rng, n, self.config, need_gumbel)
        secondary = bool(mask & rc_ext.RC_PASS_SECONDARY)
        if secondary and unused_render_kwargs.get("use_env_map") is False:
            mask |= rc_ext.RC_PASS_NO_ENVMAP
        lossmult = fields.get("lossmult")
        rc_plan, layout = self._plan(n, secondary, lossmult is not None)
        out_flat = None
        if self._out_arena is not None:                            # a row of the image's output arena (device loop)
            arena, row = self._out_arena
            if row < arena.shape[0] and arena.shape[1] == rc_plan[0]:
                out_flat = arena[row]
                self._out_arena = (arena, row + 1)
        flat, _ = self.rc.render_chunk(fields, randoms, mask, rc_plan, out_flat)
        extras = ()
        if lossmult is not None:       # models.py:2055-2063: rays.lossmult broadcast over the colour channels
            extras = (self.rc._dev(lossmult).reshape(-1, 1).expand(-1, 3),)
        render = RenderDict(flat, layout, self._consts, extras)
        return {"render": render, "main": {"integrator": render}, "cache_main": {"integrator": render}}

    __call__ = apply

    def _apply_transient(self, variables, rng, rays, is_secondary, resample):
        """TransientNeRFModel (internal/models.py:912-985) as the cache of TransientMaterialModel with
        use_material=False: sampler -> TransientNeRFMLP -> TransientVolumeIntegrator on primary rays.
        `rgb` is the [n, n_bins, 3] transient; models.render_image drops every other `transient*` key except
        the two `*_viz` ones (internal/models.py:2403, 2459-2472)."""
        import torch

        if is_secondary or resample:
            raise NotImplementedError("the time-resolved cache renders primary rays without resampling "
                                      "(TransientNeRFModel.resample_render = False)")
        self._ensure_variables(variables)
        fields = rays.hot_fields() if isinstance(rays, Rays) else dict(rays)
        if fields.get("lights") is None or fields.get("cam_origins") is None:
            raise ValueError("transient rays need `lights` and `cam_origins`")
        n = int(np.prod(np.shape(fields["near"])))
        randoms, _ = _draw_randoms(rng, n, self.config, False)
        r = dict(self.rc.render_transient(fields, randoms))
        zeros3 = torch.zeros_like(r["integrated_rgb"])
        r["transient_indirect"] = r["transient_indirect_viz"]           # render.py:503 (final value of the key)
        r["transient_direct"] = r["transient_direct_viz"]               # dark_level = 0
        for k in ("ambient_rgb", "ambient_diffuse_rgb", "ambient_specular_rgb"):   # use_ambient = False
            r[k] = zeros3
        r["normals_to_use"] = r["normals_pred"]
        r["ray_dists"] = r["ray_dists"][:, None]
        r["light_dists"] = r["light_dists"][:, None]
        for k in _FINAL_INTEGRATOR_KEYS:
            if k in r:
                r["cache_" + k] = r[k]
        r["vignette"] = torch.ones_like(r["integrated_rgb"][:, :1])
        lossmult = fields.get("lossmult")
        lm = torch.ones_like(r["vignette"]) if lossmult is None else self.rc._dev(lossmult).reshape(-1, 1)
        r["lossmult"] = lm * torch.ones_like(r["integrated_rgb"])
        return {"render": r, "main": {"integrator": r}, "cache_main": {"integrator": r}}

    def _apply_material(self, variables, rng, rays):
        """passes ("cache", "light", "material") with use_material / use_light_sampler /
        MaterialModel.resample_render (stage material_light_from_scratch_resample; internal/models.py:1144-1254,
        1398-1694).  `rng`: the dict of explicit random tensors (see rc_material_randoms in include/rc_abi.h;
        oracle-compatible generator: oracle.material_ref.draw_randoms), or a uint32[2] key, from which the tensors are
        derived at the reference's split sites (prng.material_pass_randoms).

        NOT parity-verified: the threefry primitives are pinned by published known answers, but the ~30 split sites
        material_pass_randoms restates by hand have never been compared with a jax run (none is possible in this
        pipeline): a mis-ordered split would yield a different, equally valid-looking stream.  Parity claims of the
        material stage are made on explicit random tensors only."""
        import torch

        n_rays = int(np.prod(np.shape(rays.origins if isinstance(rays, Rays) else rays["origins"])[:-1]))
        if prng.is_key(rng):
            rng = prng.material_pass_randoms(rng, n_rays, self.config)
        if not isinstance(rng, dict):
            raise ValueError("the material stage needs a uint32[2] key or explicit random tensors: pass rng as the dict "
                             "described by rc_material_randoms")
        if "vmf_noise" not in rng:
            # the reference's constant: normal(random_split(PRNGKey(1))[0], [R, 1, 128, 3]) (light_sampler.py:135-144)
            rng = dict(rng, vmf_noise=prng.light_vmf_noise((n_rays, 1, self.config.num_vmf, 3))[:, 0])
        self._ensure_variables(variables)
        fields = rays.hot_fields() if isinstance(rays, Rays) else dict(rays)
        cres, mres = self.rc.render_material(fields, rng)
        cache = self._finalize(cres, fields)
        r = dict(mres)
        zeros3 = torch.zeros_like(r["rgb"])
        for k in ("indirect_occ", "material_roughness", "material_metalness", "material_F_0", "ray_dists", "light_dists"):
            r[k] = r[k][:, None]
        r["material_diffuseness"] = torch.zeros_like(r["material_F_0"])     # constants of the configured
        r["material_mirrorness"] = torch.zeros_like(r["material_F_0"])      # microfacet material
        r["occ"] = zeros3
        for k in ("distance_mean", "distance_median", "distance_percentile_5", "distance_percentile_95"):
            r[k] = cache[k]
        for k, v in cache.items():
            if k.startswith("cache_"):
                r[k] = v
        r["material_rgb"] = r["rgb"]
        r["normals"] = cache["normals"]
        r["normals_pred"] = cache["normals_pred"]
        r["vignette"] = torch.ones_like(r["rgb"][:, :1])
        r["lossmult"] = torch.ones_like(r["rgb"][:, :1])
        return {"render": r, "main": {"integrator": r}, "cache_main": {"integrator": cache}}

    def _finalize(self, dev: Dict[str, Any], fields) -> Dict[str, Any]:
        """Integrator keys + the aliases/constants of _finalize_outputs (internal/models.py:2074-2171)."""
        import torch

        r = dict(dev)
        zeros3 = torch.zeros_like(r["rgb"])
        # exact duplicates / exact zeros of the configured passive shader (internal/nerf.py:1044-1084)
        r["ambient_rgb"] = r["direct_rgb"]
        r["direct_diffuse_rgb"] = r["direct_rgb"]
        r["ambient_diffuse_rgb"] = r["direct_rgb"]
        for k in ("occ", "irradiance_rgb", "light_radiance_rgb", "n_dot_l_rgb", "direct_specular_rgb",
                  "ambient_specular_rgb"):
            r[k] = zeros3
        r["normals_to_use"] = r["normals_pred"]
        r["ray_dists"] = r["ray_dists"][:, None]
        r["light_dists"] = r["light_dists"][:, None]
        for k in _FINAL_INTEGRATOR_KEYS:
            if k in r:
                r["cache_" + k] = r[k]
        r["vignette"] = torch.ones_like(r["rgb"][:, :1])
        lossmult = fields.get("lossmult")
        lm = torch.ones_like(r["rgb"][:, :1]) if lossmult is None else self.rc._dev(lossmult).reshape(-1, 1)
        r["lossmult"] = lm * torch.ones_like(r["rgb"])
        return r


# ------------------------------------------------------------------------------------------------
# utils.shard / unshard, create_render_fn, render_image
# ------------------------------------------------------------------------------------------------
def shard(xs, n_local_devices: int = 1):
    """utils.shard (internal/utils.py:333-335) with jax.local_device_count() == 1 per process."""
    fn = lambda x: x.reshape((n_local_devices, -1) + tuple(x.shape[1:]))
    return xs.tree_map(fn) if isinstance(xs, (Rays, Pixels)) else fn(xs)


def unshard(x, padding: int = 0):
    """utils.unshard (internal/utils.py:338-343)."""
    y = x.reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))
    return y[:-padding] if padding > 0 else y


def _strip_device_axis(a, ndim: int):
    """Arguments the Trainer replicates for pmap (flax.jax_utils.replicate) carry a leading [n_local_devices = 1]."""
    a = np.asarray(a)
    while a.ndim > ndim:
        a = a[0]
    return a


def _cast_pixels(model: Model, cameras, lights, pixels: Pixels, camtype) -> Rays:
    """camera_utils.cast_ray_batch(cameras, lights, pixels, camtype) (internal/camera_utils.py:1225-1329) on the device
    (rc_cast_rays): per-pixel camera lookup by `cam_idx`, rays with the batch shape of the pixels.  cameras =
    (pixtocams, camtoworlds, distortion_params, pixtocam_ndc[, z_range]) as the reference's tuple."""
    import torch

    from .camera import Camera
    if cameras is None or camtype is None:
        raise AssertionError("When passing Pixels into render_eval_fn, cameras and camtype needs to be not None. "
                             f"Got cameras={cameras} camtype={camtype}.")      # train_utils.py:3785-3789
    cams = tuple(cameras)
    z_range = None if len(cams) <= 4 or cams[4] is None else tuple(float(v) for v in np.asarray(_strip_device_axis(cams[4], 1)).reshape(-1)[:2])
    distortion = cams[2] if len(cams) > 2 else None
    ndc = None if len(cams) <= 3 or cams[3] is None else _strip_device_axis(cams[3], 2)
    ctype = getattr(camtype, "value", camtype)
    pixtocams = _strip_device_axis(cameras[0], 3)
    camtoworlds = _strip_device_axis(cameras[1], 3)
    lights = None if lights is None else _strip_device_axis(lights, 2)
    to_np = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    sh = tuple(np.shape(pixels.near)[:-1])                      # batch shape SH; metadata fields are SH + [1]
    cam_idx = to_np(pixels.cam_idx).reshape(-1).astype(np.int64)
    px = to_np(pixels.pix_x_int).reshape(-1).astype(np.int32)
    py = to_np(pixels.pix_y_int).reshape(-1).astype(np.int32)
    if px.size != cam_idx.size:
        raise ValueError("Pixels fields must share their batch shape")

    def cam(i):
        p2c = pixtocams if pixtocams.ndim == 2 else pixtocams[i]
        c2w = camtoworlds if camtoworlds.ndim == 2 else camtoworlds[i]
        light = None if lights is None else (lights if lights.ndim == 1 else lights[i])     # lights[cam_idx], :1288
        return Camera(pixtocam=p2c, camtoworld=c2w[:3, :4], light=light, near=0.0, far=0.0, camtype=ctype,
                      distortion_params=distortion, pixtocam_ndc=ndc, z_range=z_range)

    uniq = np.unique(cam_idx)
    if len(uniq) == 1:
        rays = model.rc.cast_rays(cam(int(uniq[0])), px, py)
    else:
        parts, order = [], []
        for i in uniq:
            sel = np.nonzero(cam_idx == i)[0]
            parts.append(model.rc.cast_rays(cam(int(i)), px[sel], py[sel]))
            order.append(sel)
        inv = torch.from_numpy(np.argsort(np.concatenate(order), kind="stable")).to(parts[0].origins.device)
        rays = Rays(**{k: (None if v is None else torch.cat([getattr(p, k) for p in parts])[inv])
                       for k, v in vars(parts[0]).items()})
    dev = rays.origins.device
    meta = lambda a, dt=torch.float32: torch.as_tensor(to_np(a)).to(device=dev, dtype=dt).reshape(sh + (1,))
    rays = rays.tree_map(lambda t: t.reshape(sh + (t.shape[-1],)))
    return rays.replace(lossmult=meta(pixels.lossmult), near=meta(pixels.near), far=meta(pixels.far),
                        cam_idx=meta(pixels.cam_idx, torch.int32), light_idx=meta(pixels.light_idx, torch.int32),
                        pix_x_int=pixels.pix_x_int, pix_y_int=pixels.pix_y_int)


def create_render_fn(model: Model, dataset: Any = None, mapping_fn: Any = None):
    """train_utils.create_render_fn(model, dataset, mapping_fn) (internal/train_utils.py:3742-3831).

    Returns `render_eval_pfn(variables, rng, train_frac, cameras, lights, rays, passes, resample=None)` ->
    (renderings, rng), called exactly as at engine/trainer.py:822-832.  One process drives one GPU, so the pmap axes
    have size 1: `rng` is the per-device key array [1, 2] (or a single key / None / seed / dict, see the module
    docstring), `rays` is sharded [1, m, .] (or a `Pixels` batch, cast on the device), and every value of the result
    carries the [n_dev = 1, n_dev = 1, m, ...] leading axes of pmap + all_gather so `unshard(v[0], padding)` applies
    unchanged.  `mapping_fn` (jax.pmap / jax.vmap in the reference) has nothing to map here and is ignored.
    `dataset` supplies `camtype` for the Pixels branch; its env_map / mesh inputs must be unset (the BASELINE
    configs render without them)."""
    camtype = getattr(dataset, "camtype", None) if dataset is not None else None
    if dataset is not None and getattr(dataset, "env_map", None) is not None:
        raise NotImplementedError("dataset.env_map is not part of the accelerated path")

    def render_eval_pfn(variables, rng, train_frac, cameras, lights, rays, passes, resample=None):
        if isinstance(rays, Pixels):
            rays = _cast_pixels(model, cameras, lights, rays, camtype)
        dev_keys = isinstance(rng, np.ndarray) and rng.dtype == np.uint32 and rng.shape == (1, 2)
        key = rng[0] if dev_keys else rng
        if prng.is_key(key):
            # render_eval_fn: one split for model.apply, one for the rng handed back (train_utils.py:3794, 3817-3818)
            apply_key, key = prng.random_split(key)
            next_key, _ = prng.random_split(key)
            next_rng = next_key[None] if dev_keys else next_key
        else:
            apply_key, next_rng = rng, rng
        out = model.apply(variables, apply_key, rays, train_frac=train_frac, train=False, passes=tuple(passes),
                          resample=resample, compute_extras=True)
        render = out["render"]
        if isinstance(render, RenderDict):
            render = render.with_lead(2)
        else:
            render = {k: v[None, None] for k, v in render.items()}
        return render, next_rng

    render_eval_pfn.device = model.device
    render_eval_pfn.model = model
    render_eval_pfn.plain = True           # made by create_render_fn: a bound closure may be short-cut (bind_render_fn)
    return render_eval_pfn


def bind_render_fn(render_eval_pfn, variables=None, train_frac: float = 1.0, cameras=None, lights=None):
    """The closure Trainer.render_primary_rays hands to render_image (engine/trainer.py:821-832):
    render_fn(rng, rays, passes, resample) -> render_eval_pfn(variables, rng, train_frac, cameras, lights, rays, ...)."""

    def render_fn(rng, rays, passes, resample=None):
        return render_eval_pfn(variables, rng, train_frac, cameras, lights, rays, passes, resample)

    render_fn.device = getattr(render_eval_pfn, "device", None)
    render_fn.model = getattr(render_eval_pfn, "model", None)
    # render_image's device loop may call the model directly for the plain cache pass: this closure adds nothing to
    # what Model.apply receives but `variables`
    render_fn.variables = variables
    render_fn.direct_ok = render_fn.model is not None and train_frac == 1.0 and getattr(render_eval_pfn, "plain", False)
    return render_fn


_STAT_KEYS = ("rgb", "integrated_rgb", "lighting_irradiance", "direct_rgb", "indirect_rgb", "material_rgb",
              "specular_rgb", "diffuse_rgb", "material_albedo", "acc")
_VAR_KEYS = ("rgb", "integrated_rgb")
_TRANSIENT_KEEP = ("transient_direct_viz", "transient_indirect_viz")


def _skip_key(k: str) -> bool:
    return ("transient" in k) and (k not in _TRANSIENT_KEEP)       # models.py:2459, 2472


def render_image(render_fn, rng, rays: Rays, config, passes: Tuple[str, ...], verbose: bool = True,
                 resample: Any = None, num_repeats: int = 1, compute_variance: bool = False):
    """models.render_image (internal/models.py:2361-2525): chunked host loop, edge padding,
    Welford mean over repeats, row-major scatter into [H, W, ...] float32 numpy arrays.

    When `render_fn` comes from this module (`render_fn.device` names a GPU) the loop stays on the device: the rays
    are uploaded once, every chunk is enqueued without a host synchronisation (chunks alternate between two HIP
    streams so one chunk's gathers overlap the other's matrix work), the per-chunk results stay in HBM, and the
    image is unpacked and copied to the host ONCE at the end.  Same chunks, same padding, same Welford update,
    same returned arrays as the host loop (which any other callable gets); aliased keys (`cache_rgb` is `rgb`)
    share one array."""
    dev = getattr(render_fn, "device", None)
    if dev is not None:
        return _render_image_device(render_fn, dev, rng, rays, config, passes, verbose, resample, num_repeats,
                                    compute_variance)
    height, width = rays.origins.shape[:2]
    num_rays = height * width
    rays = rays.tree_map(lambda r: np.asarray(r).reshape((num_rays, -1)) if np.size(r) >= num_rays else np.asarray(r))
    rendering = None
    chunk = config.render_chunk_size
    idx0s = range(0, num_rays, chunk)
    start = time.time()
    for i_chunk, idx0 in enumerate(idx0s):
        if verbose and i_chunk % max(1, len(idx0s) // 10) == 0:
            print(f"Rendering chunk {i_chunk}/{len(idx0s)-1}")
        chunk_size = min(chunk, num_rays - idx0)
        chunk_rays = rays.tree_map(lambda r: r[idx0: idx0 + chunk_size])
        padding = 0
        if chunk_size % chunk != 0:
            padding = chunk - (chunk_size % chunk)
            chunk_rays = chunk_rays.tree_map(lambda r: np.pad(r, ((0, padding), (0, 0)), mode="edge"))
        chunk_rays = shard(chunk_rays)
        means: Dict[str, np.ndarray] = {}
        m2: Dict[str, np.ndarray] = {}
        for i_repeat in range(num_repeats):
            cur, rng = render_fn(rng, chunk_rays, passes, resample)
            cur = {k: np.array(unshard(v[0].cpu().numpy() if hasattr(v, "cpu") else np.asarray(v[0]), padding))
                   for k, v in cur.items()}
            if rendering is None:
                rendering = {}
                for k, v in cur.items():
                    if _skip_key(k):
                        continue
                    rendering[k] = np.zeros((height, width) + v.shape[1:], dtype=v.dtype)
                    if compute_variance and k in _VAR_KEYS:
                        rendering[f"{k}_variance"] = np.zeros_like(rendering[k])
            for k, v in cur.items():
                if _skip_key(k):
                    continue
                if k not in means:
                    means[k] = v.copy()
                    if compute_variance and num_repeats > 1 and k in _VAR_KEYS:
                        m2[k] = np.zeros_like(v)
                elif k in _STAT_KEYS:
                    delta = v - means[k]
                    means[k] += delta / (i_repeat + 1)
                    if compute_variance and num_repeats > 1 and k in _VAR_KEYS:
                        m2[k] += delta * (v - means[k])
        ind = np.arange(chunk_size)
        ys, xs = (idx0 + ind) // width, (idx0 + ind) % width
        for k, v in means.items():
            rendering[k][ys, xs] = v[:chunk_size]
            if compute_variance and num_repeats > 1 and k in _VAR_KEYS and k in m2:
                rendering[f"{k}_variance"][ys, xs] = ((m2[k] / (num_repeats - 1)) * num_repeats)[:chunk_size]
    if verbose:
        print("Milliseconds per ray", (time.time() - start) * 1000 / (height * width))
    return rendering, rng


class _ImageSink:
    """Per-chunk results of the device loop -> [H, W, ...] numpy arrays with ONE device-to-host copy.

    Chunks whose `render` is a RenderDict over the same layout are kept as their flat buffers only; the image is
    assembled on the device (one gather per distinct region) into a key-major buffer that is copied to the host once;
    aliased keys share a numpy array.  Anything else (plain dicts, Welford means) goes key by key."""

    def __init__(self, height, width, chunk, compute_variance):
        self.h, self.w, self.chunk, self.var = height, width, chunk, compute_variance
        self.flats, self.layout, self.consts, self.extras = [], None, None, []
        self.generic = []                  # per chunk {key: [chunk, ...] tensor}
        self.keys = None

    def add_flat(self, rd: RenderDict):
        if self.layout is None:
            self.layout, self.consts = rd.layout, rd.consts
        if rd.layout is not self.layout or self.generic:
            return self.add_dict({k: rd[k][0, 0] if rd.lead == 2 else rd[k] for k in rd})
        self.flats.append(rd.flat)
        self.extras.append(rd.extras)

    def add_dict(self, d):
        if self.flats:                     # mixed: fall back to key-by-key for everything
            raise RuntimeError("render_fn changed its result type between chunks")
        self.generic.append(d)

    def finish(self, num_rays):
        import torch
        H, W = self.h, self.w
        out: Dict[str, np.ndarray] = {}
        if self.flats:
            A = torch.stack(self.flats)                                # [n_chunks, flat_size]
            regions, order = {}, []
            for k, ent in self.layout.items():
                if _skip_key(k):
                    continue
                if ent not in regions:
                    regions[ent] = None
                    order.append(ent)
            sizes, total = {}, 0
            for ent in order:
                kind, a, shape = ent
                width = int(np.prod(shape[1:])) if len(shape) > 1 else 1
                if kind == "const":
                    continue
                sizes[ent] = (total, width)
                total += num_rays * width
            packed = torch.empty(total, dtype=torch.float32, device=A.device)
            m = self.chunk
            for ent, (off, width) in sizes.items():
                kind, a, shape = ent
                dst = packed[off: off + num_rays * width].view(num_rays, width)
                if kind == "flat":
                    dst.copy_(A[:, a: a + m * width].reshape(-1, width)[:num_rays])
                else:                                                  # per-chunk extra tensors (lossmult)
                    dst.copy_(torch.cat([e[a].reshape(m, width) for e in self.extras])[:num_rays])
            host = np.empty(total, dtype=np.float32)
            torch.from_numpy(host).copy_(packed)                       # the one device-to-host copy (synchronises)
            arrays = {}
            for ent in order:
                kind, a, shape = ent
                tail = tuple(shape[1:])
                if kind == "const":
                    arrays[ent] = np.full((H, W) + tail, a, dtype=np.float32)
                else:
                    off, width = sizes[ent]
                    arrays[ent] = host[off: off + num_rays * width].reshape((H, W) + tail)
            for k, ent in self.layout.items():
                if not _skip_key(k):
                    out[k] = arrays[ent]
            return out
        keys = [k for k in self.generic[0] if not _skip_key(k)] if self.generic else []
        for k in keys:
            v = torch.cat([c[k] for c in self.generic])[:num_rays]
            out[k] = v.cpu().numpy().reshape((H, W) + tuple(v.shape[1:]))
        return out


def _render_image_device(render_fn, dev, rng, rays, config, passes, verbose, resample, num_repeats, compute_variance):
    """The loop of models.render_image (internal/models.py:2412-2514) with everything between the upload of the rays
    and the final copy of the image kept on GPU `dev`."""
    import torch

    height, width = rays.origins.shape[:2]
    num_rays = height * width
    device = torch.device("cuda", dev)
    chunk = config.render_chunk_size
    n_chunks = -(-num_rays // chunk)
    padding = n_chunks * chunk - num_rays

    def upload(r):
        t = r if isinstance(r, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(r))
        if t.numel() < num_rays:
            return t.to(device)
        t = t.to(device).reshape(num_rays, -1)
        if padding:                                                    # np.pad(mode="edge") of the last chunk
            t = torch.cat([t, t[-1:].expand(padding, -1)])
        return t.reshape(n_chunks, 1, chunk, -1)                       # [chunk index][shard = 1][m][.]: utils.shard done

    start = time.time()
    main = torch.cuda.current_stream(device)
    drays = rays.tree_map(upload)
    names = [k for k, v in vars(drays).items() if v is not None and v.dim() == 4]
    cols = {k: getattr(drays, k).unbind(0) for k in names}
    rest = {k: v for k, v in vars(drays).items() if k not in names}
    pool = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)] if n_chunks > 1 else [main]
    # Host time per chunk: the variable tree is checked for updates on the first chunk only, and the chunks write their
    # outputs into rows of ONE zero-filled arena made here, on the main stream the pool streams fork from (no allocation
    # + fill launch per chunk)
    mdl = getattr(render_fn, "model", None)
    arena = None
    if (mdl is not None and num_repeats == 1 and n_chunks > 1 and "material" not in passes
            and getattr(mdl.config, "transient", None) is None and hasattr(mdl, "_plan")):
        total = mdl._plan(chunk, "is_secondary" in passes, "lossmult" in names)[0][0]
        arena = torch.zeros((n_chunks, total), dtype=torch.float32, device=device)
    ready = torch.cuda.Event()
    ready.record(main)
    for s in pool:
        s.wait_event(ready)
    sink = _ImageSink(height, width, chunk, compute_variance)
    var_out: Dict[str, list] = {}
    # Direct path for the plain deterministic cache pass of this package's own model (rng None: nothing to thread from
    # chunk to chunk): exactly the calls render_fn -> render_eval_pfn -> Model.apply would make for the chunk, without
    # rebuilding a Rays / RenderDict per chunk (host time per chunk 38 -> ~15 us; the results are the same buffers).
    direct = (arena is not None and rng is None and tuple(passes) == ("cache",) and not resample
              and getattr(render_fn, "direct_ok", False))
    try:
      if arena is not None:
          mdl._out_arena = (arena, 0)
      if direct:
          hot = [k for k in names if k in ("origins", "directions", "viewdirs", "near", "far", "lights", "lossmult")]
          rc_plan, layout = mdl._plan(chunk, False, "lossmult" in names)
          mdl._ensure_variables(getattr(render_fn, "variables", None))
          handles = [s_.cuda_stream for s_ in pool]
          # the loop over the chunks itself runs in the library (rc_render_chunks: chunk i on stream i % 2, outputs into
          # row i of the arena); what is left per chunk here is the bookkeeping of the sink
          full = {k: getattr(drays, k) for k in hot if k != "lossmult"}
          mdl.rc.render_chunks(full, chunk, n_chunks, rc_ext.RC_PASS_CACHE, rc_plan, arena, handles)
          for i_chunk in range(n_chunks):
              if verbose and i_chunk % max(1, n_chunks // 10) == 0:
                  print(f"Rendering chunk {i_chunk}/{n_chunks-1}")
              row = arena[i_chunk]
              extras = (cols["lossmult"][i_chunk][0].reshape(-1, 1).expand(-1, 3),) if "lossmult" in hot else ()
              sink.add_flat(RenderDict(row, layout, mdl._consts, extras))
      for i_chunk in range(0 if not direct else n_chunks, n_chunks):
          if verbose and i_chunk % max(1, n_chunks // 10) == 0:
              print(f"Rendering chunk {i_chunk}/{n_chunks-1}")
          chunk_rays = Rays(**rest, **{k: cols[k][i_chunk] for k in names})
          with torch.cuda.stream(pool[i_chunk % len(pool)]):
              if num_repeats == 1:
                  cur, rng = render_fn(rng, chunk_rays, passes, resample)
                  if isinstance(cur, RenderDict):
                      sink.add_flat(cur)
                      if mdl is not None:
                          mdl._variables_checked = True
                  else:
                      sink.add_dict({k: v[0].reshape((-1,) + tuple(v.shape[3:])) for k, v in cur.items()})
                  continue
              means: Dict[str, Any] = {}
              m2: Dict[str, Any] = {}
              for i_repeat in range(num_repeats):                        # Welford on the device (models.py:2483-2490)
                  cur, rng = render_fn(rng, chunk_rays, passes, resample)
                  for k in cur:
                      if _skip_key(k):
                          continue
                      v = cur[k][0].reshape((-1,) + tuple(cur[k].shape[3:]))
                      if k not in means:
                          means[k] = v.clone() if k in _STAT_KEYS else v
                          if compute_variance and k in _VAR_KEYS:
                              m2[k] = torch.zeros_like(v)
                      elif k in _STAT_KEYS:
                          delta = v - means[k]
                          means[k] += delta / (i_repeat + 1)
                          if compute_variance and k in _VAR_KEYS:
                              m2[k] += delta * (v - means[k])
              sink.add_dict(means)
              for k, v in m2.items():
                  var_out.setdefault(k, []).append((v / (num_repeats - 1)) * num_repeats)
    finally:
        if mdl is not None:
            mdl._variables_checked = False
            mdl._out_arena = None
    done = [torch.cuda.Event() for _ in pool]
    for s, e in zip(pool, done):
        e.record(s)
        main.wait_event(e)
    rendering = sink.finish(num_rays)
    for k, parts in var_out.items():
        v = torch.cat(parts)[:num_rays]
        rendering[f"{k}_variance"] = v.cpu().numpy().reshape((height, width) + tuple(v.shape[1:]))
    if compute_variance and num_repeats == 1:
        for k in _VAR_KEYS:
            if k in rendering:
                rendering[f"{k}_variance"] = np.zeros_like(rendering[k])
    if verbose:
        print("Milliseconds per ray", (time.time() - start) * 1000 / (height * width))
    return rendering, rng


# ------------------------------------------------------------------------------------------------
# Multi-GPU: one process per GPU, rays sharded at image granularity, one all-gather per image
# ------------------------------------------------------------------------------------------------
GATHER_KEYS = ("rgb", "acc", "distance_median", "normals_pred")


def shard_bounds(num_rays: int, rank: int, world: int) -> Tuple[int, int]:
    """Static contiguous split: rank r owns rays [r*ceil(N/G), ...)."""
    per = -(-num_rays // world)
    lo = min(rank * per, num_rays)
    return lo, min(lo + per, num_rays)


def _advance_rng(rng, rank: int, first: bool):
    """Per-chunk random stream of a rank: the key is folded with the rank once (ranks draw different numbers, as the
    per-device keys of the reference's pmap do), then threaded from chunk to chunk like render_image threads the key
    render_fn hands back (internal/models.py:2445; train_utils.py:3794, 3817-3818).  Returns (key for this chunk,
    carry for the next one).  Seeds / Generators advance by themselves; dicts of explicit tensors and None pass."""
    if prng.is_key(rng):
        if first:
            rng = prng.fold_in(rng, rank)
        apply_key, key = prng.random_split(rng)
        next_key, _ = prng.random_split(key)
        return apply_key, next_key
    if first and isinstance(rng, (int, np.integer)):
        rng = np.random.Generator(np.random.PCG64([int(rng), rank]))
    return rng, rng


_PINNED: Dict[Tuple[int, ...], Any] = {}


def _to_host_once(t):
    """One device-to-host copy of a [rays, columns] float32 tensor into a (cached) pinned buffer; returns numpy."""
    import torch
    if not t.is_cuda:
        return t.numpy()
    shape = tuple(t.shape)
    host = _PINNED.get(shape)
    if host is None:
        _PINNED.clear()                                   # one image size at a time: do not hoard pinned memory
        host = _PINNED[shape] = torch.empty(shape, dtype=torch.float32, pin_memory=True)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy().copy()                            # the caller owns its arrays; the staging buffer is reused


def render_image_distributed(model_apply, rng, rays: Rays, config, passes=("cache",), keys=GATHER_KEYS,
                             group=None, device=None, key_widths: Optional[Dict[str, int]] = None,
                             num_repeats: int = 1, to_host: bool = False):
    """Each rank renders its contiguous share of the image in `render_chunk_size` batches, keeps the
    results on its device, packs the consumed keys into one [rays_per_rank, sum(widths)] buffer and
    issues ONE all_gather per image (the reference all-gathers the whole ~45-key dict per chunk per
    repeat, internal/train_utils.py:3795-3815).  With num_repeats > 1 the repeats of a chunk are averaged on the
    device BEFORE the gather (running mean, the update of internal/models.py:2483-2490) -- for the
    reference's stat_keys only (internal/models.py:2398-2401, `_STAT_KEYS`); every other key keeps the first repeat's
    value, exactly as render_image / _render_image_device do (a mean of unit normals is not a unit normal).

    model_apply(rng, rays) -> {"render": {key: tensor[n, ...]}}; runs on "nccl" (= RCCL over xGMI)
    with the HIP model and on "gloo" with any CPU callable (tests).  `device`: where a rank with an empty shard
    allocates its (all-padding) contribution; defaults to the current cuda device under the nccl backend.
    `to_host=True` returns float32 numpy arrays, as `render_image` does (internal/models.py:2448-2450 copies every chunk of
    every key; here the gathered keys cross PCIe as ONE device-to-host copy per image, into a pinned staging buffer)."""
    import torch
    import torch.distributed as dist

    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    widths = dict(rc_ext.OUTPUTS)
    if key_widths:
        widths.update(key_widths)
    cols = np.cumsum([0] + [widths[k] for k in keys])
    height, width = rays.origins.shape[:2]
    num_rays = height * width
    numel = lambda r: r.numel() if hasattr(r, "numel") else np.size(r)
    flat = rays.tree_map(lambda r: (r if hasattr(r, "reshape") else np.asarray(r)).reshape((num_rays, -1))
                         if numel(r) >= num_rays else r)
    lo, hi = shard_bounds(num_rays, rank, world)
    per = -(-num_rays // world)
    chunk = config.render_chunk_size
    buf = None
    first = True
    stat_cols = None                   # [sum(widths)] bool: columns whose key the reference averages over repeats
    for idx0 in range(lo, hi, chunk):
        sub = flat.tree_map(lambda r: r[idx0: min(idx0 + chunk, hi)])
        m = min(idx0 + chunk, hi) - idx0
        mean = None
        for i_repeat in range(num_repeats):
            key, rng = _advance_rng(rng, rank, first)
            first = False
            out = model_apply(key, sub)["render"]
            cur = torch.cat([out[k].reshape(m, -1) for k in keys], dim=1)
            if mean is None:
                mean = cur.clone() if num_repeats > 1 else cur
                continue
            if stat_cols is None:
                mask = np.concatenate([np.full(widths[k], k in _STAT_KEYS) for k in keys])
                stat_cols = torch.from_numpy(mask).to(cur.device)
            delta = cur - mean                                          # models.py:2485-2486, stat_keys only
            mean += torch.where(stat_cols, delta / (i_repeat + 1), torch.zeros_like(delta))
        if buf is None:
            dev = device or mean.device
            buf = torch.zeros((per, int(cols[-1])), dtype=torch.float32, device=dev)
        buf[idx0 - lo: idx0 - lo + m] = mean
    if buf is None:
        if device is None:
            nccl = on and dist.get_backend(group) == "nccl"
            device = torch.device("cuda", torch.cuda.current_device()) if nccl else "cpu"
        buf = torch.zeros((per, int(cols[-1])), dtype=torch.float32, device=device)
    if on:        # also with one rank: the collective (RCCL under "nccl") then runs the way it does with eight
        gathered = torch.empty((world * per, int(cols[-1])), dtype=torch.float32, device=buf.device)
        if buf.is_cuda:
            dist.all_gather_into_tensor(gathered, buf, group=group)
        else:
            dist.all_gather(list(gathered.chunk(world, dim=0)), buf, group=group)
    else:
        gathered = buf
    if to_host:
        gathered = _to_host_once(gathered[:num_rays])
    result = {}
    for i, k in enumerate(keys):
        v = gathered[:num_rays, cols[i]: cols[i + 1]]
        result[k] = v.reshape((height, width) + ((widths[k],) if widths[k] > 1 else ()))
    return result
