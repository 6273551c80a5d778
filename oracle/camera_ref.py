"""Pinhole ray generation (oracle; see oracle/__init__.py -- PARITY UNPINNED like the rest).

Restates camera_utils.pixels_to_rays (internal/camera_utils.py:896-1072) and the parts of cast_ray_batch
(:1225-1329) that apply to the BASELINE scenes: ProjectionType.PERSPECTIVE, distortion_params=None,
pixtocam_ndc=None, z_range=None, jitter=0, xnp=numpy (the dataset / eval path casts with numpy).
"""
from __future__ import annotations

import numpy as np


def get_pixtocam(focal, width, height):
    """camera_utils.get_pixtocam (:760-763): inverse of intrinsic_matrix(f, f, w/2, h/2)."""
    camtopix = np.array([[focal, 0, width * 0.5], [0, focal, height * 0.5], [0, 0, 1.0]])
    return np.linalg.inv(camtopix)


def pixels_to_rays(pix_x_int, pix_y_int, pixtocam, camtoworld, dtype=np.float32, camtype="perspective"):
    pix_x_int = np.asarray(pix_x_int)
    pix_y_int = np.asarray(pix_y_int)
    pixtocam = np.asarray(pixtocam, dtype)
    camtoworld = np.asarray(camtoworld, dtype)

    def pix_to_dir(x, y):
        return np.stack([x + 0.5, y + 0.5, np.ones_like(x)], axis=-1)

    px = pix_x_int.astype(dtype)
    py = pix_y_int.astype(dtype)
    stacked = np.stack([pix_to_dir(px, py), pix_to_dir(px + 1, py), pix_to_dir(px, py + 1)], axis=0)
    mat_vec_mul = lambda A, b: np.matmul(A, b[..., None])[..., 0]
    cam_dirs = mat_vec_mul(pixtocam, stacked)
    if camtype == "pano":
        # ProjectionType.PANORAMIC (:1013-1024)
        theta, phi = cam_dirs[..., 0], cam_dirs[..., 1]
        cam_dirs = np.stack([-np.sin(phi) * np.sin(theta), -np.cos(phi), -np.sin(phi) * np.cos(theta)], axis=-1).astype(dtype)
    cam_dirs = np.matmul(cam_dirs, np.diag(np.array([1.0, -1.0, -1.0], dtype)))      # OpenCV -> OpenGL
    imageplane = cam_dirs[0, ..., :2]
    dirs = mat_vec_mul(camtoworld[..., :3, :3], cam_dirs)
    directions, dx, dy = dirs
    origins = np.broadcast_to(camtoworld[..., :3, -1], directions.shape)
    viewdirs = directions / np.linalg.norm(directions, axis=-1, keepdims=True)
    look = np.broadcast_to(-camtoworld[..., :3, 2], directions.shape)
    up = np.broadcast_to(camtoworld[..., :3, 1], directions.shape)
    dx_norm = np.linalg.norm(dx - directions, axis=-1)
    dy_norm = np.linalg.norm(dy - directions, axis=-1)
    radii = (0.5 * (dx_norm + dy_norm))[..., None] * 2 / np.sqrt(12)
    return dict(origins=origins, directions=directions, viewdirs=viewdirs, radii=radii.astype(dtype), imageplane=imageplane,
                look=look, up=up)


def cast_ray_batch(pixtocam, camtoworld, light, pix_x_int, pix_y_int, near, far, dtype=np.float32, camtype="perspective"):
    """cast_ray_batch for one camera: rays + lights = lights[cam_idx], cam_origins = origins, near / far from Pixels."""
    r = pixels_to_rays(pix_x_int, pix_y_int, pixtocam, camtoworld, dtype, camtype)
    shape = r["directions"].shape
    r["lights"] = np.broadcast_to(np.asarray(light, dtype), shape)
    r["cam_origins"] = r["origins"]
    r["near"] = np.full(shape[:-1] + (1,), near, dtype)
    r["far"] = np.full(shape[:-1] + (1,), far, dtype)
    return r


def cast_spherical_rays(camtoworld, height, width, near, far, light=None, dtype=np.float32):
    """camera_utils.cast_spherical_rays (:1415-1443) -> cast_general_rays with ProjectionType.PANORAMIC."""
    pixtocam = np.diag(np.array([2.0 * np.pi / width, np.pi / height, 1.0]))
    py, px = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    camtoworld = np.asarray(camtoworld)[:3, :4]
    light = camtoworld[:3, 3] if light is None else light
    return cast_ray_batch(pixtocam, camtoworld, light, px, py, near, far, dtype, camtype="pano")
