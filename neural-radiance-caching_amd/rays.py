"""`Rays` batch struct (mirror of internal/utils.py:142-169) and synthetic ray batches.

Field names, order and shapes follow the reference dataclass so that a host which
builds `utils.Rays` can hand the same arrays over.  Arrays may be numpy arrays or
torch tensors; the path reads origins, directions, viewdirs, near, far, lights,
lossmult and (secondary rays) normals -- the rest is carried through untouched.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Optional

import numpy as np

_Array = Any


@dataclasses.dataclass
class Rays:
    origins: _Array
    lights: _Array
    directions: _Array
    viewdirs: _Array
    radii: _Array
    imageplane: _Array
    look: _Array
    up: _Array
    cam_origins: _Array
    vcam_look: _Array
    vcam_up: _Array
    vcam_origins: _Array
    lossmult: _Array
    near: _Array
    far: _Array
    cam_idx: _Array
    light_idx: _Array
    normals: Optional[_Array] = None
    pix_x_int: Optional[_Array] = None
    pix_y_int: Optional[_Array] = None
    exposure_idx: Optional[_Array] = None
    exposure_values: Optional[_Array] = None
    device_idx: Optional[_Array] = None
    impulse_response: Optional[_Array] = None

    def replace(self, **kw) -> "Rays":
        return dataclasses.replace(self, **kw)

    def tree_map(self, fn) -> "Rays":
        """Apply fn to every non-None field (jax.tree_util.tree_map over the pytree)."""
        return Rays(**{f.name: (None if getattr(self, f.name) is None else fn(getattr(self, f.name)))
                       for f in dataclasses.fields(self)})

    def hot_fields(self) -> dict:
        """The fields the hot path consumes, as a plain dict."""
        d = dict(origins=self.origins, directions=self.directions, viewdirs=self.viewdirs,
                 near=self.near, far=self.far, lights=self.lights, lossmult=self.lossmult)
        if self.cam_origins is not None:
            d["cam_origins"] = self.cam_origins      # transient only (render_utils.py:1733-1740)
        if self.normals is not None:
            d["normals"] = self.normals
        return d


@dataclasses.dataclass
class Pixels:
    """Mirror of utils.Pixels (internal/utils.py:126-139): integer pixel coordinates + per-ray metadata; what
    create_render_fn's device function casts into Rays itself (internal/train_utils.py:3762-3792)."""
    pix_x_int: _Array
    pix_y_int: _Array
    lossmult: _Array
    near: _Array
    far: _Array
    cam_idx: _Array
    light_idx: _Array
    exposure_idx: Optional[_Array] = None
    exposure_values: Optional[_Array] = None
    device_idx: Optional[_Array] = None

    def tree_map(self, fn) -> "Pixels":
        return Pixels(**{f.name: (None if getattr(self, f.name) is None else fn(getattr(self, f.name)))
                         for f in dataclasses.fields(self)})


def _normalize(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _lookat_frame(origins):
    look = _normalize(-origins)
    world_up = np.broadcast_to(np.array([0.0, 0.0, 1.0]), origins.shape)
    right = _normalize(np.cross(look, world_up))
    up = np.cross(right, look)
    return look, up


def synthetic_rays(n_rays: int, seed: int = 20200823, near: float = 2.0, far: float = 6.0) -> Rays:
    """Random primary rays of a TensoIR/Blender style capture:
    origins on the radius-4.03 upper shell, looking at random points of a radius-0.8 ball,
    un-normalised pinhole `directions`, radii of an 800 px / focal 1111 camera."""
    rng = np.random.Generator(np.random.PCG64(seed))
    o = rng.normal(size=(n_rays, 3))
    o[:, 2] = np.abs(o[:, 2])
    o = _normalize(o)
    o[:, 2] = np.maximum(o[:, 2], 0.1)
    o = 4.03 * _normalize(o)
    p = _normalize(rng.normal(size=(n_rays, 3))) * (0.8 * rng.uniform(size=(n_rays, 1)) ** (1 / 3))
    viewdirs = _normalize(p - o)
    directions = viewdirs * (1.0 + np.abs(rng.normal(scale=0.05, size=(n_rays, 1))))
    look, up = _lookat_frame(o)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    ones = np.ones((n_rays, 1))
    zi = np.zeros((n_rays, 1), dtype=np.int32)
    return Rays(
        origins=f32(o), lights=f32(o), directions=f32(directions), viewdirs=f32(viewdirs),
        radii=f32(ones * 5.2e-4), imageplane=f32(np.zeros((n_rays, 2))), look=f32(look), up=f32(up),
        cam_origins=f32(o), vcam_look=f32(look), vcam_up=f32(up), vcam_origins=f32(o),
        lossmult=f32(ones), near=f32(ones * near), far=f32(ones * far), cam_idx=zi, light_idx=zi.copy())


def synthetic_transient_rays(n_rays: int, seed: int = 20200823, near: float = 0.7, far: float = 4.0) -> Rays:
    """Random primary rays of a simulated transient capture (cornell scale, Config.near/far of
    transient_simulation_ngp_yobo_cornell.gin:28-32): cameras on a radius-2.5 shell looking at a radius-0.6
    ball, a point light next to each camera (`lights`), `cam_origins` = the camera centre."""
    r = synthetic_rays(n_rays, seed, near, far)
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    scale = np.float32(2.5 / 4.03)
    o = r.origins * scale
    lights = (o + rng.normal(scale=0.05, size=o.shape)).astype(np.float32)
    return dataclasses.replace(r, origins=o, cam_origins=o.copy(), vcam_origins=o.copy(), lights=lights)


def synthetic_camera_rays(height: int, width: int, focal: float = 1111.0, cam_origin=(0.0, -3.5, 2.0),
                          near: float = 2.0, far: float = 6.0) -> Rays:
    """A look-at pinhole camera cast the way the Blender loader does (pixel centres,
    OpenGL camera axes; internal/camera_utils.py:896-1072): rays shaped [H, W, .]."""
    o = np.asarray(cam_origin, dtype=np.float64)
    look = _normalize(-o)
    right = _normalize(np.cross(look, np.array([0.0, 0.0, 1.0])))
    up = np.cross(right, look)
    ys, xs = np.meshgrid(np.arange(height) + 0.5, np.arange(width) + 0.5, indexing="ij")
    cx = (xs - width * 0.5) / focal
    cy = -(ys - height * 0.5) / focal
    d = cx[..., None] * right + cy[..., None] * up + look
    viewdirs = _normalize(d)
    dx = np.linalg.norm(d[:, 1:] - d[:, :-1], axis=-1)
    dx = np.concatenate([dx, dx[:, -1:]], axis=1)
    dy = np.linalg.norm(d[1:] - d[:-1], axis=-1)
    dy = np.concatenate([dy, dy[-1:]], axis=0)
    radii = (0.5 * (dx + dy))[..., None] * 2 / np.sqrt(12)     # camera_utils.py:1070
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    bc = lambda v: f32(np.broadcast_to(v, (height, width, 3)))
    ones = np.ones((height, width, 1))
    zi = np.zeros((height, width, 1), dtype=np.int32)
    return Rays(
        origins=bc(o), lights=bc(o), directions=f32(d), viewdirs=f32(viewdirs), radii=f32(radii),
        imageplane=f32(np.stack([cx, cy], -1)), look=bc(look), up=bc(up), cam_origins=bc(o),
        vcam_look=bc(look), vcam_up=bc(up), vcam_origins=bc(o), lossmult=f32(ones),
        near=f32(ones * near), far=f32(ones * far), cam_idx=zi, light_idx=zi.copy())
