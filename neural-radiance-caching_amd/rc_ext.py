"""ctypes binding of the C ABI in include/rc_abi.h (librc_hip.so).

This is the reference-side binding a maintainer would add: plain pointers and sizes,
no torch types cross the boundary.  torch is used only for device memory and streams.
The product path fails loudly when the HIP library is missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, Optional

import numpy as np

from .config import GridConfig, RenderConfig

RC_ABI_VERSION = 4
RC_MAX_LEVELS = 3

RC_PASS_CACHE = 0x1
RC_PASS_SECONDARY = 0x2
RC_PASS_RESAMPLE = 0x4
RC_PASS_NO_ENVMAP = 0x8

# rc_output_id -> (name, width); order must match include/rc_abi.h
OUTPUTS = (
    ("rgb", 3), ("acc", 1), ("distance_mean", 1), ("distance_percentile_5", 1), ("distance_median", 1),
    ("distance_percentile_95", 1), ("diffuse_rgb", 3), ("specular_rgb", 3), ("direct_rgb", 3),
    ("indirect_rgb", 3), ("albedo_rgb", 3), ("indirect_diffuse_rgb", 3), ("indirect_specular_rgb", 3),
    ("indirect_occ", 3), ("means", 3), ("normals", 3), ("normals_pred", 3), ("ray_dists", 1),
    ("light_dists", 1), ("env_map_rgb", 3), ("rgb_no_env", 3),
)
OUTPUT_ID = {name: i for i, (name, _) in enumerate(OUTPUTS)}
RC_OUT_COUNT = len(OUTPUTS)


class rc_grid_config(C.Structure):
    _fields_ = [("hash_map_size", C.c_int32), ("max_grid_size", C.c_int32), ("min_grid_size", C.c_int32),
                ("num_features", C.c_int32), ("bbox", C.c_float), ("precondition_scaling", C.c_float)]


class rc_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("num_levels", C.c_int32), ("num_samples", C.c_int32 * RC_MAX_LEVELS),
        ("proposal_grids", rc_grid_config * RC_MAX_LEVELS), ("appearance_grid", rc_grid_config),
        ("material_grid", rc_grid_config), ("light_grid", rc_grid_config),
        ("anneal", C.c_float), ("resample_padding", C.c_float), ("raydist_p", C.c_float),
        ("raydist_premult", C.c_float), ("shadow_normal_eps_dot_min", C.c_float), ("density_bias", C.c_float),
        ("contract_radius", C.c_float), ("roughness_bias", C.c_float), ("irradiance_bias", C.c_float),
        ("ambient_irradiance_bias", C.c_float), ("rgb_max", C.c_float), ("slf_ambient_bias", C.c_float),
        ("env_rgb_bias", C.c_float), ("env_map_distance", C.c_float), ("bg_intensity", C.c_float),
        ("percentiles", C.c_float * 3), ("num_resample", C.c_int32),
        ("diffuse_sample_fraction", C.c_float), ("secondary_normal_eps", C.c_float), ("secondary_near", C.c_float),
        ("secondary_far", C.c_float), ("min_roughness", C.c_float), ("default_F_0", C.c_float),
        ("vmf_scale", C.c_float), ("num_vmf", C.c_int32),
    ]


class rc_tensor_desc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("on_device", C.c_int32)]


class rc_rays(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("origins", "directions", "viewdirs", "near", "far", "lights", "normals")]


class rc_grad_segment(C.Structure):
    _fields_ = [("name", C.c_char * 160), ("offset", C.c_int64), ("size", C.c_int64), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


class rc_randoms(C.Structure):
    _fields_ = [("jitter", C.c_void_p * RC_MAX_LEVELS), ("gumbel", C.c_void_p), ("resample_inds", C.c_void_p)]


class rc_outputs(C.Structure):
    _fields_ = [("ptr", C.c_void_p * RC_OUT_COUNT)]


# rc_mat_output_id -> (name, width); order must match include/rc_abi.h
MAT_OUTPUTS = (
    ("rgb", 3), ("acc", 1), ("direct_rgb", 3), ("indirect_rgb", 3), ("diffuse_rgb", 3), ("specular_rgb", 3),
    ("direct_diffuse_rgb", 3), ("direct_specular_rgb", 3), ("indirect_diffuse_rgb", 3), ("indirect_specular_rgb", 3),
    ("indirect_occ", 1), ("lighting_irradiance", 3), ("material_albedo", 3), ("material_roughness", 1),
    ("material_metalness", 1), ("material_F_0", 1), ("means", 3), ("normals_to_use", 3), ("ray_dists", 1),
    ("light_dists", 1),
)
MAT_OUTPUT_ID = {name: i for i, (name, _) in enumerate(MAT_OUTPUTS)}
RC_MOUT_COUNT = len(MAT_OUTPUTS)


class rc_material_randoms(C.Structure):
    _fields_ = [("gumbel", C.c_void_p), ("vmf_noise", C.c_void_p), ("spec_u1", C.c_void_p), ("spec_u2", C.c_void_p),
                ("cos_u1", C.c_void_p), ("cos_u2", C.c_void_p), ("vmf_lobe", C.c_void_p), ("vmf_v", C.c_void_p),
                ("vmf_tmp", C.c_void_p), ("sec_jitter", C.c_void_p * RC_MAX_LEVELS), ("sec_gumbel", C.c_void_p),
                ("resample_inds", C.c_void_p), ("sec_resample_inds", C.c_void_p), ("vmf_lobe_gumbel", C.c_void_p)]


class rc_mat_outputs(C.Structure):
    _fields_ = [("ptr", C.c_void_p * RC_MOUT_COUNT)]


class rc_transient_config(C.Structure):
    _fields_ = [("n_bins", C.c_int32), ("exposure_time", C.c_float), ("tfilter_sigma", C.c_float),
                ("transient_shift", C.c_float), ("bin_zero_threshold_light", C.c_int32), ("light_near", C.c_float),
                ("light_zero", C.c_int32), ("use_falloff", C.c_int32), ("indirect_scale", C.c_float),
                ("rgb_max", C.c_float), ("albedo_bias", C.c_float), ("brdf_bias", C.c_float),
                ("irradiance_bias", C.c_float), ("slf_rgb_bias", C.c_float), ("use_occlusions", C.c_int32),
                ("occ_threshold", C.c_float), ("shadow_near", C.c_float), ("shadow_far", C.c_float),
                ("reserved", C.c_int32 * 6)]


# rc_transient_output_id -> (name, trailing shape); order must match include/rc_abi.h
TRANSIENT_OUTPUTS = (
    ("rgb", "bins"), ("transient_direct_viz", "bins"), ("transient_indirect_viz", "bins"),
    ("transient_indirect_diffuse", "bins"), ("transient_indirect_specular", "bins"), ("integrated_rgb", 3),
    ("direct_rgb", 3), ("indirect_rgb", 3), ("diffuse_rgb", 3), ("specular_rgb", 3), ("albedo_rgb", 3), ("occ", 3),
    ("indirect_occ", 3), ("irradiance_rgb", 3), ("light_radiance_rgb", 3), ("n_dot_l_rgb", 3),
    ("direct_diffuse_rgb", 3), ("direct_specular_rgb", 3), ("indirect_diffuse_rgb", 3), ("indirect_specular_rgb", 3),
    ("direct_rgb_viz", 3), ("acc", 1), ("distance_mean", 1), ("distance_median", 1), ("distance_percentile_5", 1),
    ("distance_percentile_95", 1), ("means", 3), ("normals", 3), ("normals_pred", 3), ("ray_dists", 1),
    ("light_dists", 1),
)
TRANSIENT_OUTPUT_ID = {name: i for i, (name, _) in enumerate(TRANSIENT_OUTPUTS)}
RC_TOUT_COUNT = len(TRANSIENT_OUTPUTS)


class rc_transient_outputs(C.Structure):
    _fields_ = [("ptr", C.c_void_p * RC_TOUT_COUNT)]


class rc_camera(C.Structure):
    _fields_ = [("pixtocam", C.c_float * 9), ("camtoworld", C.c_float * 12), ("light", C.c_float * 3),
                ("near", C.c_float), ("far", C.c_float), ("camtype", C.c_int32),
                ("has_distortion", C.c_int32), ("distortion", C.c_float * 6),
                ("has_ndc", C.c_int32), ("pixtocam_ndc", C.c_float * 9),
                ("has_z_range", C.c_int32), ("z_range", C.c_float * 2), ("pix_dx", C.c_void_p), ("pix_dy", C.c_void_p)]


CAST_OUTPUTS = (("origins", 3), ("directions", 3), ("viewdirs", 3), ("radii", 1), ("imageplane", 2), ("look", 3), ("up", 3),
                ("lights", 3), ("near", 1), ("far", 1))


class rc_cast_outputs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k, _ in CAST_OUTPUTS]


EXPORTS = (
    "rc_create", "rc_destroy", "rc_last_error", "rc_abi_version", "rc_mlp_arithmetic", "rc_load_weights", "rc_render_rays", "rc_render_chunks",
    "rc_hashgrid_lookup", "rc_sample_intervals", "rc_workspace_ptr", "rc_set_profiling", "rc_stage_count",
    "rc_stage_name", "rc_stage_times_ms", "rc_set_graph_mode", "rc_set_fused", "rc_render_material", "rc_set_transient", "rc_render_transient", "rc_cast_rays",
    "rc_prng_fill", "rc_density_grad_size", "rc_density_grad_layout", "rc_density_backward",
    "rc_hashgrid_grad_layout", "rc_hashgrid_backward", "rc_allgather_outputs",
)

_LIB = None
_RAY_FIELDS = ("origins", "directions", "viewdirs", "near", "far", "lights", "normals")


def library_path() -> str:
    # RC_HIP_LIBRARY: a diagnostic / A-B build of the same ABI (tools/README.md); the product is the in-tree library
    return os.environ.get("RC_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librc_hip.so")


def mlp_arithmetic() -> str:
    """'f32-mfma' or 'bf16x3-split' (include/rc_abi.h rc_mlp_arithmetic): how the loaded library multiplies in the shader MLPs."""
    return "bf16x3-split" if load_library().rc_mlp_arithmetic() == 1 else "f32-mfma"


def source_hash() -> str:
    """sha256 over the kernel / ABI sources the library is built from: what measurement files kept under profiles/
    (PMC traffic, in-kernel phase stamps) are tagged with, so that a bench run can tell a stale file from a current one."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".inc")))
    files.append(os.path.join(os.path.dirname(here), "include", "rc_abi.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_library():
    """dlopen librc_hip.so.  Raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the render path.")
    lib = C.CDLL(path)
    lib.rc_create.argtypes = [C.POINTER(rc_config), C.c_int, C.POINTER(C.c_void_p)]
    lib.rc_create.restype = C.c_int
    lib.rc_destroy.argtypes = [C.c_void_p]
    lib.rc_destroy.restype = None
    lib.rc_last_error.argtypes = [C.c_void_p]
    lib.rc_last_error.restype = C.c_char_p
    lib.rc_abi_version.restype = C.c_int
    lib.rc_mlp_arithmetic.restype = C.c_int
    lib.rc_load_weights.argtypes = [C.c_void_p, C.POINTER(rc_tensor_desc), C.c_int32]
    lib.rc_load_weights.restype = C.c_int
    lib.rc_render_rays.argtypes = [C.c_void_p, C.POINTER(rc_rays), C.c_int64, C.POINTER(rc_randoms), C.c_uint32,
                                   C.POINTER(rc_outputs), C.c_void_p]
    lib.rc_render_rays.restype = C.c_int
    lib.rc_render_chunks.argtypes = [C.c_void_p, C.POINTER(rc_rays), C.c_int64, C.c_int64, C.c_uint32,
                                     C.POINTER(rc_outputs), C.c_int64, C.POINTER(C.c_void_p), C.c_int32]
    lib.rc_render_chunks.restype = C.c_int
    lib.rc_render_material.argtypes = [C.c_void_p, C.POINTER(rc_rays), C.c_int64, C.POINTER(rc_randoms),
                                       C.POINTER(rc_material_randoms), C.c_int32, C.POINTER(rc_outputs),
                                       C.POINTER(rc_mat_outputs), C.c_void_p]
    lib.rc_render_material.restype = C.c_int
    lib.rc_hashgrid_lookup.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rc_hashgrid_lookup.restype = C.c_int
    lib.rc_sample_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rc_sample_intervals.restype = C.c_int
    lib.rc_workspace_ptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.rc_workspace_ptr.restype = C.c_int
    lib.rc_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    lib.rc_set_profiling.restype = C.c_int
    lib.rc_set_graph_mode.argtypes = [C.c_void_p, C.c_int32]
    lib.rc_set_graph_mode.restype = C.c_int
    lib.rc_set_fused.argtypes = [C.c_void_p, C.c_int32]
    lib.rc_set_fused.restype = C.c_int
    lib.rc_set_transient.argtypes = [C.c_void_p, C.c_void_p]
    lib.rc_set_transient.restype = C.c_int
    lib.rc_render_transient.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
    lib.rc_render_transient.restype = C.c_int
    lib.rc_cast_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_void_p, C.c_void_p]
    lib.rc_cast_rays.restype = C.c_int
    lib.rc_prng_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_int64, C.c_void_p, C.c_void_p]
    lib.rc_prng_fill.restype = C.c_int
    lib.rc_density_grad_size.argtypes = [C.c_void_p, C.c_int32]
    lib.rc_density_grad_size.restype = C.c_int64
    lib.rc_density_grad_layout.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    lib.rc_density_grad_layout.restype = C.c_int
    lib.rc_density_backward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    lib.rc_density_backward.restype = C.c_int
    lib.rc_hashgrid_grad_layout.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.rc_hashgrid_grad_layout.restype = C.c_int
    lib.rc_hashgrid_backward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rc_hashgrid_backward.restype = C.c_int
    lib.rc_allgather_outputs.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(rc_outputs), C.c_int64, C.POINTER(rc_outputs),
                                         C.c_void_p]
    lib.rc_allgather_outputs.restype = C.c_int
    lib.rc_stage_count.restype = C.c_int
    lib.rc_stage_name.argtypes = [C.c_int32]
    lib.rc_stage_name.restype = C.c_char_p
    lib.rc_stage_times_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int32]
    lib.rc_stage_times_ms.restype = C.c_int
    _LIB = lib
    return lib


def _grid_c(g: GridConfig) -> rc_grid_config:
    return rc_grid_config(g.hash_map_size, g.max_grid_size, g.min_grid_size, g.num_features, g.bbox,
                          g.precondition_scaling)


def config_to_c(cfg: RenderConfig) -> rc_config:
    c = rc_config()
    c.abi_version = RC_ABI_VERSION
    c.num_levels = cfg.num_levels
    for i, (_, _, n) in enumerate(cfg.sampling_strategy):
        c.num_samples[i] = n
        c.proposal_grids[i] = _grid_c(cfg.proposal_grids[i])
    c.appearance_grid = _grid_c(cfg.appearance_grid)
    c.material_grid = _grid_c(cfg.material_grid)
    c.light_grid = _grid_c(cfg.light_grid)
    for k in ("anneal", "resample_padding", "raydist_p", "raydist_premult", "shadow_normal_eps_dot_min",
              "density_bias", "contract_radius", "roughness_bias", "irradiance_bias", "ambient_irradiance_bias",
              "rgb_max", "slf_ambient_bias", "env_rgb_bias", "env_map_distance"):
        setattr(c, k, float(getattr(cfg, k)))
    c.bg_intensity = float(cfg.bg_intensity)
    for i, p in enumerate(cfg.percentiles):
        c.percentiles[i] = float(p)
    c.num_resample = cfg.num_resample
    for k in ("diffuse_sample_fraction", "secondary_normal_eps", "secondary_near", "secondary_far", "min_roughness",
              "default_F_0", "vmf_scale"):
        setattr(c, k, float(getattr(cfg, k)))
    c.num_vmf = cfg.num_vmf
    return c


def transient_config_to_c(t) -> rc_transient_config:
    c = rc_transient_config()
    for k in ("n_bins", "bin_zero_threshold_light"):
        setattr(c, k, int(getattr(t, k)))
    for k in ("light_zero", "use_falloff", "use_occlusions"):
        setattr(c, k, 1 if getattr(t, k) else 0)
    for k in ("exposure_time", "tfilter_sigma", "transient_shift", "light_near", "indirect_scale", "rgb_max", "albedo_bias",
              "brdf_bias", "irradiance_bias", "slf_rgb_bias", "occ_threshold", "shadow_near", "shadow_far"):
        setattr(c, k, float(getattr(t, k)))
    return c


class RcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rc error {code}: {msg}")
        self.code = code


class RadianceCache:
    """One rc_handle: the cache renderer bound to one GPU."""

    def __init__(self, cfg: RenderConfig, device: int = 0):
        import torch  # device memory / streams only

        self._torch = torch
        self.lib = load_library()
        self.cfg = cfg
        self.device = device
        self._h = C.c_void_p()
        ccfg = config_to_c(cfg)
        rc = self.lib.rc_create(C.byref(ccfg), device, C.byref(self._h))
        if rc != 0:
            raise RcError(rc, (self.lib.rc_last_error(None) or b"").decode())
        self._keep = []
        if cfg.transient is not None:
            self._check(self.lib.rc_set_transient(self._h, C.byref(transient_config_to_c(cfg.transient))))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.rc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise RcError(rc, (self.lib.rc_last_error(self._h) or b"").decode())

    # -- weights ----------------------------------------------------------------------------
    def load_weights(self, weights: Dict[str, object]):
        """weights: flat dict 'params/...' -> numpy array or torch tensor (float32)."""
        torch = self._torch
        descs = (rc_tensor_desc * len(weights))()
        keep = []
        for i, (name, w) in enumerate(weights.items()):
            if isinstance(w, torch.Tensor):
                t = w.detach().to(torch.float32).contiguous()
                keep.append(t)
                ptr, on_dev, shape = t.data_ptr(), int(t.is_cuda), tuple(t.shape)
            else:
                a = np.ascontiguousarray(w, dtype=np.float32)
                keep.append(a)
                ptr, on_dev, shape = a.ctypes.data, 0, a.shape
            bname = name.encode()
            keep.append(bname)
            descs[i].name = bname
            descs[i].data = ptr
            descs[i].ndim = len(shape)
            for k, s in enumerate(shape):
                descs[i].shape[k] = s
            descs[i].on_device = on_dev
        # rc_load_weights copies with blocking hipMemcpy on the null stream, which is NOT ordered against torch's
        # non-blocking side streams: finish whatever produced device-resident tensors first (an optimizer step)
        if any(isinstance(w, torch.Tensor) and w.is_cuda for w in weights.values()):
            torch.cuda.synchronize(self.device)
        self._check(self.lib.rc_load_weights(self._h, descs, len(weights)))

    # -- hot path ---------------------------------------------------------------------------
    def _zeros_like_many(self, shapes):
        """Zero-filled float32 cuda tensors of the given shapes as views of ONE allocation (one fill kernel instead of
        one per output; every view starts on a 256-byte boundary)."""
        torch = self._torch
        offs, total = [], 0
        for shp in shapes:
            offs.append(total)
            total += (int(np.prod(shp)) + 63) // 64 * 64
        flat = torch.zeros(max(total, 1), dtype=torch.float32, device=f"cuda:{self.device}")
        return [flat[o: o + int(np.prod(shp))].view(shp) for o, shp in zip(offs, shapes)]

    def _dev(self, x, dtype=None):
        torch = self._torch
        dtype = dtype or torch.float32
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(x))
        elif x.is_cuda and x.dtype == dtype and x.is_contiguous() and x.device.index == self.device:
            return x                    # already where and how the library wants it (the per-call path of a training loop)
        return x.to(device=f"cuda:{self.device}", dtype=dtype).contiguous()

    def render_rays(self, rays: Dict[str, object], randoms: Optional[Dict[str, object]] = None,
                    pass_mask: int = RC_PASS_CACHE, outputs: Optional[Iterable[str]] = None,
                    out: Optional[Dict[str, object]] = None):
        """rays: dict with origins, directions, viewdirs [n,3], near, far [n] or [n,1], optional lights,
        normals.  Returns dict name -> torch cuda tensor ([n,3] or [n]).  Passing the dict returned by
        an earlier call as `out` reuses its buffers (same pointers -> the captured hipGraph is replayed)."""
        torch = self._torch
        r = rc_rays()
        held = {}
        n = None
        for k in ("origins", "directions", "viewdirs", "near", "far", "lights", "normals"):
            v = rays.get(k)
            if v is None:
                continue
            t = self._dev(v)
            t = t.reshape(-1, 3) if k not in ("near", "far") else t.reshape(-1)
            held[k] = t
            setattr(r, k, t.data_ptr())
            n = t.shape[0] if n is None else n
            if t.shape[0] != n:
                raise ValueError(f"ray field {k} has {t.shape[0]} rows, expected {n}")
        rnd_p = None
        if randoms is not None:
            rnd = rc_randoms()
            jit = randoms.get("jitter")
            if jit is not None:
                for l, j in enumerate(jit):
                    if j is not None:
                        t = self._dev(j).reshape(-1)
                        held[f"jit{l}"] = t
                        rnd.jitter[l] = t.data_ptr()
            if randoms.get("gumbel") is not None:
                held["gumbel"] = self._dev(randoms["gumbel"])
                rnd.gumbel = held["gumbel"].data_ptr()
            if randoms.get("resample_inds") is not None:
                held["inds"] = self._dev(randoms["resample_inds"], torch.int32).reshape(-1)
                rnd.resample_inds = held["inds"].data_ptr()
            rnd_p = C.byref(rnd)
        names = [nm for nm, _ in OUTPUTS] if outputs is None else list(outputs)
        if out is not None:
            names = list(out.keys())
        cout = rc_outputs()
        res = {}
        dev = f"cuda:{self.device}"
        shapes = {nm: ((n, 3) if OUTPUTS[OUTPUT_ID[nm]][1] == 3 else (n,)) for nm in names}
        fresh = dict(zip(names, self._zeros_like_many([shapes[nm] for nm in names]))) if out is None else None
        for nm in names:
            shape = shapes[nm]
            t = out[nm] if out is not None else fresh[nm]
            if tuple(t.shape) != shape or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"output buffer {nm}: expected contiguous float32 cuda tensor of shape {shape}")
            res[nm] = t
            cout.ptr[OUTPUT_ID[nm]] = t.data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_render_rays(self._h, C.byref(r), n, rnd_p, pass_mask, C.byref(cout), stream))
        self._keep = [held]   # keep inputs alive until the next call (async enqueue)
        return res

    # -- lean per-chunk launch (models.render_image's hot loop at render_chunk_size = 1024) ------------
    def output_plan(self, names, n: int):
        """Layout of ONE flat float32 allocation holding the outputs `names` of an n-ray batch (every output starts on a
        256-byte boundary): (total floats, {name: (offset, shape)}, [(rc_output_id, offset)])."""
        offs, total, ids = {}, 0, []
        for nm in names:
            width = OUTPUTS[OUTPUT_ID[nm]][1]
            offs[nm] = (total, (n, 3) if width == 3 else (n,))
            ids.append((OUTPUT_ID[nm], total))
            total += (n * width + 63) // 64 * 64
        return max(total, 1), offs, ids

    def render_chunk(self, rays: Dict[str, object], randoms, pass_mask: int, plan, out_flat=None, stream_handle=None):
        """rc_render_rays into one fresh flat buffer laid out by `plan` (output_plan).  The hot-loop variant of
        render_rays: device-resident float32 ray fields are passed by pointer as they are (no reshape / copy), the
        outputs are not wrapped into per-key tensors.  Returns (flat tensor, n)."""
        torch = self._torch
        r = rc_rays()
        held = []
        n = None
        for k in _RAY_FIELDS:
            v = rays.get(k)
            if v is None:
                continue
            if not (isinstance(v, torch.Tensor) and v.is_cuda and v.dtype is torch.float32 and v.is_contiguous()):
                v = self._dev(v)
            held.append(v)
            setattr(r, k, v.data_ptr())
            if k == "near":
                n = v.numel()
        rnd_p = None
        if randoms is not None:
            rnd = rc_randoms()
            jit = randoms.get("jitter")
            if jit is not None:
                for l, j in enumerate(jit):
                    if j is not None:
                        t = self._dev(j)
                        held.append(t)
                        rnd.jitter[l] = t.data_ptr()
            if randoms.get("gumbel") is not None:
                t = self._dev(randoms["gumbel"])
                held.append(t)
                rnd.gumbel = t.data_ptr()
            if randoms.get("resample_inds") is not None:
                t = self._dev(randoms["resample_inds"], torch.int32)
                held.append(t)
                rnd.resample_inds = t.data_ptr()
            rnd_p = C.byref(rnd)
        total, _, ids = plan
        # out_flat: a zero-filled float32 cuda buffer of `total` elements the caller provides (a row of its arena)
        flat = torch.zeros(total, dtype=torch.float32, device=held[0].device) if out_flat is None else out_flat
        base = flat.data_ptr()
        cout = rc_outputs()
        for oid, off in ids:
            cout.ptr[oid] = base + 4 * off
        # stream_handle: the raw hipStream_t to enqueue on (a caller that alternates streams skips torch's context manager)
        stream = torch.cuda.current_stream(self.device).cuda_stream if stream_handle is None else stream_handle
        self._check(self.lib.rc_render_rays(self._h, C.byref(r), n, rnd_p, pass_mask, C.byref(cout), stream))
        self._keep = held          # inputs stay alive until the next call (async enqueue)
        return flat, n

    def render_chunks(self, rays: Dict[str, object], chunk: int, n_chunks: int, pass_mask: int, plan, arena, stream_handles):
        """rc_render_chunks: the chunk loop of render_image in native code.  `rays`: float32 cuda tensors holding
        n_chunks * chunk rays each (contiguous, the last chunk edge-padded by the caller); `arena`: zero-filled
        [n_chunks, total] float32 cuda tensor whose rows are laid out by `plan`; chunk i goes to
        stream_handles[i % len] (raw hipStream_t values).  One ABI call for the whole image."""
        torch = self._torch
        r = rc_rays()
        held = []
        for k in _RAY_FIELDS:
            v = rays.get(k)
            if v is None:
                continue
            assert isinstance(v, torch.Tensor) and v.is_cuda and v.dtype is torch.float32 and v.is_contiguous(), k
            assert v.numel() == n_chunks * chunk * (1 if k in ("near", "far") else 3), (k, tuple(v.shape))
            held.append(v)
            setattr(r, k, v.data_ptr())
        total, _, ids = plan
        assert arena.is_contiguous() and tuple(arena.shape) == (n_chunks, total)
        base = arena.data_ptr()
        cout = rc_outputs()
        for oid, off in ids:
            cout.ptr[oid] = base + 4 * off
        hs = (C.c_void_p * len(stream_handles))(*stream_handles)
        self._check(self.lib.rc_render_chunks(self._h, C.byref(r), chunk, n_chunks, pass_mask, C.byref(cout), total, hs,
                                              len(stream_handles)))
        self._keep = held

    def _rays_struct(self, rays):
        r = rc_rays()
        held = {}
        n = None
        for k in ("origins", "directions", "viewdirs", "near", "far", "lights", "normals"):
            v = rays.get(k)
            if v is None:
                continue
            t = self._dev(v)
            t = t.reshape(-1, 3) if k not in ("near", "far") else t.reshape(-1)
            held[k] = t
            setattr(r, k, t.data_ptr())
            n = t.shape[0] if n is None else n
            if t.shape[0] != n:
                raise ValueError(f"ray field {k} has {t.shape[0]} rows, expected {n}")
        return r, held, n

    def density_grad_layout(self, level: int):
        """rc_density_grad_layout: [(tensor name, offset, shape)] of the gradient buffer of proposal level `level`
        (the reference's parameter-tree names), and its total size in floats."""
        cnt = C.c_int32()
        self._check(self.lib.rc_density_grad_layout(self._h, level, None, 0, C.byref(cnt)))
        segs = (rc_grad_segment * cnt.value)()
        self._check(self.lib.rc_density_grad_layout(self._h, level, segs, cnt.value, C.byref(cnt)))
        out = [(s.name.decode(), int(s.offset), tuple(int(v) for v in s.shape[: s.ndim])) for s in segs]
        total = int(self.lib.rc_density_grad_size(self._h, level))
        if total < 0:
            self._check(total)
        return out, total

    def hashgrid_grad_layout(self, grid_id: int):
        """rc_hashgrid_grad_layout: [(tensor name, offset, shape)] of the table-gradient buffer of a grid, and its size."""
        cnt, total = C.c_int32(), C.c_int64()
        self._check(self.lib.rc_hashgrid_grad_layout(self._h, grid_id, None, 0, C.byref(cnt), C.byref(total)))
        segs = (rc_grad_segment * cnt.value)()
        self._check(self.lib.rc_hashgrid_grad_layout(self._h, grid_id, segs, cnt.value, C.byref(cnt), C.byref(total)))
        return [(s.name.decode(), int(s.offset), tuple(int(v) for v in s.shape[: s.ndim])) for s in segs], int(total.value)

    def hashgrid_backward(self, grid_id: int, points, d_features, grads=None, apply_contraction: bool = True):
        """rc_hashgrid_backward: scatter d L / d features [n, L*F] (the layout hashgrid_lookup returns) into the tables'
        gradient buffer (flat float32 cuda tensor of hashgrid_grad_layout(grid_id)[1] elements; accumulated into when given)."""
        torch = self._torch
        pts = self._dev(points).reshape(-1, 3).contiguous()
        n = pts.shape[0]
        df = self._dev(d_features).reshape(n, -1).contiguous()
        _, total = self.hashgrid_grad_layout(grid_id)
        if grads is None:
            grads = torch.zeros(total, dtype=torch.float32, device=f"cuda:{self.device}")
        elif grads.numel() != total or grads.dtype != torch.float32 or not grads.is_cuda or not grads.is_contiguous():
            raise ValueError(f"grads must be a contiguous float32 cuda tensor of {total} elements")
        g = self.cfg_grid(grid_id)
        if df.shape[1] != g.out_dim:
            raise ValueError(f"d_features must have {g.out_dim} columns")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_hashgrid_backward(self._h, grid_id, pts.data_ptr(), n, df.data_ptr(), grads.data_ptr(),
                                                  1 if apply_contraction else 0, stream))
        self._keep = [pts, df]
        return grads

    def density_backward(self, level: int, points, d_density, d_feature=None, grads=None):
        """rc_density_backward: gradients of L w.r.t. the hash-grid tables and the density MLP of proposal level `level`
        given d L / d density [n] (and d L / d feature [n, 64]) at the world-space sample means `points` [n, 3].
        Returns (grads, density): `grads` is the flat float32 cuda buffer of density_grad_layout(level), accumulated
        into when passed in (zeroed and allocated otherwise)."""
        torch = self._torch
        pts = self._dev(points).reshape(-1, 3).contiguous()
        n = pts.shape[0]
        dd = self._dev(d_density).reshape(-1).contiguous()
        if dd.shape[0] != n:
            raise ValueError("d_density must have one value per point")
        df = None
        if d_feature is not None:
            df = self._dev(d_feature).reshape(n, 64).contiguous()
        sizes = self.__dict__.setdefault("_grad_sizes", {})        # the layout is fixed by the config: asked once per level
        total = sizes.get(level)
        if total is None:
            total = int(self.lib.rc_density_grad_size(self._h, level))
            if total < 0:
                self._check(total)
            sizes[level] = total
        if grads is None:
            grads = torch.zeros(total, dtype=torch.float32, device=f"cuda:{self.device}")
        elif grads.numel() != total or grads.dtype != torch.float32 or not grads.is_cuda or not grads.is_contiguous():
            raise ValueError(f"grads must be a contiguous float32 cuda tensor of {total} elements")
        dens = torch.empty(n, dtype=torch.float32, device=f"cuda:{self.device}")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_density_backward(self._h, level, pts.data_ptr(), n, dd.data_ptr(),
                                                 None if df is None else df.data_ptr(), grads.data_ptr(), dens.data_ptr(),
                                                 stream))
        self._keep = [pts, dd, df]
        return grads, dens

    def prng_fill(self, key, shape, mode: str = "uniform", minval: float = 0.0, maxval: float = 1.0):
        """rc_prng_fill: the tensor jax.random.{bits,uniform,normal,gumbel}(key, shape) of the reference's pinned jax
        holds, generated in HBM.  key: uint32[2] (prng.PRNGKey / prng.split)."""
        from . import prng
        torch = self._torch
        modes = {"bits": prng.MODE_BITS, "uniform": prng.MODE_UNIFORM, "normal": prng.MODE_NORMAL, "gumbel": prng.MODE_GUMBEL}
        if mode not in modes:
            raise ValueError(f"unknown mode {mode!r}")
        k = (C.c_uint32 * 2)(*[int(v) for v in prng.as_key(key)])
        shape = tuple(int(v) for v in shape)
        n = int(np.prod(shape)) if shape else 1
        out = torch.empty(shape, dtype=torch.int32 if mode == "bits" else torch.float32, device=f"cuda:{self.device}")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_prng_fill(self._h, k, modes[mode], float(minval), float(maxval), n, out.data_ptr(), stream))
        return out

    def cast_rays(self, camera, pix_x_int=None, pix_y_int=None, rect=None, pix_jitter=None):
        """rc_cast_rays: rays of `camera` (pixtocam [3,3], camtoworld [3,4], light, near, far; optional camtype,
        distortion_params, pixtocam_ndc, z_range as in camera_utils.pixels_to_rays / cast_ray_batch) for an explicit
        pixel batch (two int arrays of one shape) or for rect = (x0, y0, width, height), as a Rays of cuda tensors
        with the batch shape of the pixels ([h, w, .] for a rectangle).  pix_jitter = (dx, dy): the sub-pixel offsets
        the reference draws when jitter > 0 (camera_utils.py:943-957), float32 arrays of the pixels' shape."""
        from .rays import Rays
        torch = self._torch
        cam = rc_camera()
        p2c = np.asarray(camera.pixtocam, np.float32).reshape(9)
        c2w = np.asarray(camera.camtoworld, np.float32).reshape(12)
        light = c2w.reshape(3, 4)[:, 3] if camera.light is None else np.asarray(camera.light, np.float32).reshape(3)
        for i in range(9):
            cam.pixtocam[i] = float(p2c[i])
        for i in range(12):
            cam.camtoworld[i] = float(c2w[i])
        for i in range(3):
            cam.light[i] = float(light[i])
        cam.near, cam.far = float(camera.near), float(camera.far)
        cam.camtype = {"perspective": 0, "pano": 1, "fisheye": 2, "fisheye_equisolid": 3}[getattr(camera, "camtype", "perspective")]
        dist = getattr(camera, "distortion_params", None)
        if dist is not None:              # dict of floats like the reference's distortion_params (k1..k4, p1, p2; missing = 0)
            cam.has_distortion = 1
            for i, k in enumerate(("k1", "k2", "k3", "k4", "p1", "p2")):
                cam.distortion[i] = float(dist.get(k, 0.0))
        ndc = getattr(camera, "pixtocam_ndc", None)
        if ndc is not None:
            cam.has_ndc = 1
            for i, v in enumerate(np.asarray(ndc, np.float32).reshape(9)):
                cam.pixtocam_ndc[i] = float(v)
        zr = getattr(camera, "z_range", None)
        if zr is not None:
            cam.has_z_range = 1
            cam.z_range[0], cam.z_range[1] = float(zr[0]), float(zr[1])
        dev = f"cuda:{self.device}"
        if rect is not None:
            x0, y0, w, hgt = (int(v) for v in rect)
            shape, n, px, py = (hgt, w), w * hgt, None, None
        else:
            px = self._dev(np.ascontiguousarray(pix_x_int), torch.int32)
            py = self._dev(np.ascontiguousarray(pix_y_int), torch.int32)
            if px.shape != py.shape:
                raise ValueError("pix_x_int and pix_y_int must have the same shape")
            shape, n, x0, y0, w, hgt = tuple(px.shape), px.numel(), 0, 0, 0, 0
        jit = None
        if pix_jitter is not None:
            jit = [self._dev(np.ascontiguousarray(j, dtype=np.float32) if not isinstance(j, torch.Tensor) else j).reshape(-1) for j in pix_jitter]
            if jit[0].numel() != n or jit[1].numel() != n:
                raise ValueError("pix_jitter: two arrays with one value per pixel")
            cam.pix_dx, cam.pix_dy = jit[0].data_ptr(), jit[1].data_ptr()
        out = rc_cast_outputs()
        t = {}
        for k, width in CAST_OUTPUTS:
            t[k] = torch.empty(shape + (width,), dtype=torch.float32, device=dev)
            setattr(out, k, t[k].data_ptr())
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_cast_rays(self._h, C.byref(cam), px.data_ptr() if px is not None else None,
                                          py.data_ptr() if py is not None else None, n, x0, y0, w, hgt, C.byref(out), stream))
        self._keep = [px, py, jit]
        ones = torch.ones(shape + (1,), dtype=torch.float32, device=dev)
        zi = torch.zeros(shape + (1,), dtype=torch.int32, device=dev)
        return Rays(origins=t["origins"], lights=t["lights"], directions=t["directions"], viewdirs=t["viewdirs"],
                    radii=t["radii"], imageplane=t["imageplane"], look=t["look"], up=t["up"], cam_origins=t["origins"],
                    vcam_look=t["look"], vcam_up=t["up"], vcam_origins=t["origins"], lossmult=ones, near=t["near"],
                    far=t["far"], cam_idx=zi, light_idx=zi)

    def render_transient(self, rays: Dict[str, object], randoms: Optional[Dict[str, object]] = None,
                         outputs: Optional[Iterable[str]] = None):
        """Time-resolved cache (rc_render_transient).  rays needs `lights` and `cam_origins` besides the usual
        fields; randoms: {"jitter": [u0, u1, u2], "shadow_jitter": [v0, v1, v2]} or None (shadow_jitter: per-level jitter of
        the n * 32 shadow rays when the config has use_occlusions).  Returns dict name -> cuda tensor: [n, n_bins, 3] for the
        histograms, [n, 3] / [n] otherwise."""
        torch = self._torch
        r, held, n = self._rays_struct(rays)
        cam = self._dev(rays["cam_origins"]).reshape(-1, 3)
        held["cam_origins"] = cam
        def jitter_struct(key):
            if randoms is None or randoms.get(key) is None:
                return None
            rnd = rc_randoms()
            for l, j in enumerate(randoms[key]):
                if j is not None:
                    t = self._dev(j).reshape(-1)
                    held[f"{key}{l}"] = t
                    rnd.jitter[l] = t.data_ptr()
            held[key + "_struct"] = rnd
            return C.byref(rnd)

        rnd_p = jitter_struct("jitter")
        shadow_p = jitter_struct("shadow_jitter")      # [3][n * 32], use_occlusions only
        names = [nm for nm, _ in TRANSIENT_OUTPUTS] if outputs is None else list(outputs)
        cout = rc_transient_outputs()
        res = {}
        dev = f"cuda:{self.device}"
        nb = self.cfg.transient.n_bins
        def tshape(nm):
            kind = TRANSIENT_OUTPUTS[TRANSIENT_OUTPUT_ID[nm]][1]
            return (n, nb, 3) if kind == "bins" else ((n, 3) if kind == 3 else (n,))
        for nm, t in zip(names, self._zeros_like_many([tshape(nm) for nm in names])):
            res[nm] = t
            cout.ptr[TRANSIENT_OUTPUT_ID[nm]] = t.data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_render_transient(self._h, C.byref(r), cam.data_ptr(), n, rnd_p, shadow_p, C.byref(cout), stream))
        self._keep = [held]
        return res

    def render_material(self, rays: Dict[str, object], randoms: Dict[str, object], num_secondary_samples: int = None):
        """Material stage (rc_render_material).  randoms: dict with the keys of
        oracle-compatible `draw_randoms` (jitter[3], gumbel, vmf_noise, spec_u1/u2, cos_u1/u2, vmf_lobe, vmf_v,
        vmf_tmp, spec_jitter[3], spec_gumbel, diff_jitter[3], diff_gumbel) and optionally the categorical picks
        themselves (resample_inds [n]; spec_resample_inds [n*Ks] + diff_resample_inds [n*Kd]), which replace the draws.
        Returns (cache_outputs, material_outputs) as dicts of cuda tensors."""
        torch = self._torch
        K = num_secondary_samples or self.cfg.num_secondary_samples
        r, held, n = self._rays_struct(rays)
        dev = f"cuda:{self.device}"
        rnd = rc_randoms()
        if randoms.get("jitter") is not None:
            for l, j in enumerate(randoms["jitter"]):
                held[f"jit{l}"] = self._dev(j).reshape(-1)
                rnd.jitter[l] = held[f"jit{l}"].data_ptr()
        mr = rc_material_randoms()
        for k in ("gumbel", "vmf_noise", "spec_u1", "spec_u2", "cos_u1", "cos_u2", "vmf_v", "vmf_tmp"):
            if k == "gumbel" and randoms.get(k) is None:
                continue                      # the primary pick is handed over as resample_inds
            held["m_" + k] = self._dev(randoms[k])
            setattr(mr, k, held["m_" + k].data_ptr())
        if randoms.get("vmf_lobe") is not None:
            held["m_lobe"] = self._dev(randoms["vmf_lobe"], torch.int32).reshape(-1)
            mr.vmf_lobe = held["m_lobe"].data_ptr()
        else:                                 # the lobe is drawn on the device: argmax(logits + Gumbel noise)
            held["m_lobe_g"] = self._dev(randoms["vmf_lobe_gumbel"]).reshape(n, -1)
            mr.vmf_lobe_gumbel = held["m_lobe_g"].data_ptr()
        # secondary trace randoms: [specular block | diffuse block]
        # (a caller that keeps them in the ABI's layout passes sec_jitter[3] [n*K] / sec_gumbel [n*K, S] and skips the copies)
        for l in range(RC_MAX_LEVELS):
            if randoms.get("sec_jitter") is not None:
                held[f"sj{l}"] = self._dev(randoms["sec_jitter"][l]).reshape(-1)
                if held[f"sj{l}"].numel() != n * K:
                    raise ValueError(f"sec_jitter[{l}]: expected {n * K} values")
            else:
                held[f"sj{l}"] = torch.cat([self._dev(randoms["spec_jitter"][l]).reshape(-1),
                                            self._dev(randoms["diff_jitter"][l]).reshape(-1)])
            mr.sec_jitter[l] = held[f"sj{l}"].data_ptr()
        if randoms.get("sec_gumbel") is not None:
            held["sg"] = self._dev(randoms["sec_gumbel"])
            if held["sg"].shape[0] != n * K:
                raise ValueError(f"sec_gumbel: expected {n * K} rows")
            mr.sec_gumbel = held["sg"].data_ptr()
        elif randoms.get("spec_gumbel") is not None and randoms.get("diff_gumbel") is not None:
            held["sg"] = torch.cat([self._dev(randoms["spec_gumbel"]), self._dev(randoms["diff_gumbel"])], dim=0).contiguous()
            mr.sec_gumbel = held["sg"].data_ptr()
        # explicit categorical picks (filtered_sampler_inds) instead of the Gumbel draws
        if randoms.get("resample_inds") is not None:
            held["m_inds"] = self._dev(randoms["resample_inds"], torch.int32).reshape(-1)
            mr.resample_inds = held["m_inds"].data_ptr()
        if randoms.get("spec_resample_inds") is not None and randoms.get("diff_resample_inds") is not None:
            held["s_inds"] = torch.cat([self._dev(randoms["spec_resample_inds"], torch.int32).reshape(-1),
                                        self._dev(randoms["diff_resample_inds"], torch.int32).reshape(-1)])
            mr.sec_resample_inds = held["s_inds"].data_ptr()
        cout, mout = rc_outputs(), rc_mat_outputs()
        cres, mres = {}, {}
        c_items = [(i, nm, width) for i, (nm, width) in enumerate(OUTPUTS) if nm not in ("env_map_rgb", "rgb_no_env")]
        m_items = [(i, nm, width) for i, (nm, width) in enumerate(MAT_OUTPUTS)]
        bufs = self._zeros_like_many([((n, 3) if width == 3 else (n,)) for _, _, width in c_items + m_items])
        for (i, nm, _), t in zip(c_items, bufs[: len(c_items)]):
            cres[nm] = t
            cout.ptr[i] = t.data_ptr()
        for (i, nm, _), t in zip(m_items, bufs[len(c_items):]):
            mres[nm] = t
            mout.ptr[i] = t.data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_render_material(self._h, C.byref(r), n, C.byref(rnd), C.byref(mr), K, C.byref(cout),
                                                C.byref(mout), stream))
        self._keep = [held]
        return cres, mres

    def allgather_outputs(self, nccl_comm: int, local: Dict[str, object], world: int):
        """rc_allgather_outputs: gather this rank's outputs (dict name -> [n, .] cuda tensor, as render_rays returns) over
        an RCCL communicator the caller owns (ncclComm_t as an integer address).  Returns name -> [world * n, .]."""
        torch = self._torch
        lo, fu, res = rc_outputs(), rc_outputs(), {}
        n = None
        for nm, t in local.items():
            n = t.shape[0] if n is None else n
            if t.shape[0] != n or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError(f"output {nm}: expected contiguous float32 cuda tensors with one row count")
            res[nm] = torch.empty((world * n,) + tuple(t.shape[1:]), dtype=torch.float32, device=t.device)
            lo.ptr[OUTPUT_ID[nm]] = t.data_ptr()
            fu.ptr[OUTPUT_ID[nm]] = res[nm].data_ptr()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_allgather_outputs(self._h, C.c_void_p(nccl_comm), C.byref(lo), n, C.byref(fu), stream))
        self._keep = [local]
        return res

    # -- single operators ---------------------------------------------------------------------
    def cfg_grid(self, grid_id: int):
        """GridConfig of grid 0-2 (proposal density grids), 3 (appearance), 4 (material), 5 (light)."""
        return (list(self.cfg.proposal_grids) + [self.cfg.appearance_grid, self.cfg.material_grid, self.cfg.light_grid])[grid_id]

    def hashgrid_lookup(self, grid_id: int, points, apply_contraction: bool = True):
        torch = self._torch
        g = self.cfg_grid(grid_id)
        p = self._dev(points).reshape(-1, 3)
        out = torch.empty((p.shape[0], g.out_dim), dtype=torch.float32, device=p.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_hashgrid_lookup(self._h, grid_id, p.data_ptr(), p.shape[0], out.data_ptr(),
                                                int(apply_contraction), stream))
        self._keep = [p]
        return out

    def sample_intervals(self, t, logits, num_samples: int, jitter=None):
        torch = self._torch
        t = self._dev(t)
        logits = self._dev(logits)
        n, P = logits.shape
        out = torch.empty((n, num_samples + 1), dtype=torch.float32, device=t.device)
        j = None if jitter is None else self._dev(jitter).reshape(-1)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.rc_sample_intervals(self._h, t.data_ptr(), logits.data_ptr(), n, P, num_samples,
                                                 None if j is None else j.data_ptr(), out.data_ptr(), stream))
        self._keep = [t, logits, j]
        return out

    # -- introspection ------------------------------------------------------------------------
    def workspace(self, name: str, dtype=None):
        """Copy of an internal buffer of the last render (tests / debugging)."""
        torch = self._torch
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._check(self.lib.rc_workspace_ptr(self._h, name.encode(), C.byref(ptr), C.byref(cnt)))
        torch.cuda.synchronize(self.device)
        host = np.empty(cnt.value, dtype=np.float32)
        _memcpy_d2h(host.ctypes.data, ptr.value, cnt.value * 4)
        return host.view(np.int32) if dtype == np.int32 else host

    def set_profiling(self, enabled):
        self._check(self.lib.rc_set_profiling(self._h, int(enabled)))

    def set_graph_mode(self, mode: int):
        """0 eager launches, 1 capture a hipGraph when a call repeats (default), 2 capture at once."""
        self._check(self.lib.rc_set_graph_mode(self._h, int(mode)))

    def set_fused(self, on: bool):
        """Plain cache pass: True (default) one fused launch per batch with every intermediate on chip,
        False one launch per stage (fills the workspace that `workspace()` shows)."""
        self._check(self.lib.rc_set_fused(self._h, int(on) if not isinstance(on, bool) else (1 if on else 0)))

    def stage_times_ms(self) -> Dict[str, float]:
        n = self.lib.rc_stage_count()
        arr = (C.c_float * n)()
        self._check(self.lib.rc_stage_times_ms(self._h, arr, n))
        return {self.lib.rc_stage_name(i).decode(): float(arr[i]) for i in range(n)}


def _memcpy_d2h(dst: int, src: int, nbytes: int):
    """hipMemcpy device->host through the HIP runtime torch already loaded."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    rc = hip.hipMemcpy(dst, src, nbytes, 2)  # hipMemcpyDeviceToHost
    if rc != 0:
        raise RuntimeError(f"hipMemcpy failed: {rc}")
