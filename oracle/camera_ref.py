"""Pinhole ray generation (oracle; see oracle/__init__.py -- PARITY UNPINNED like the rest).

Restates camera_utils.pixels_to_rays (internal/camera_utils.py:896-1072) and cast_ray_batch (:1225-1329):
ProjectionType.PERSPECTIVE (the BASELINE scenes), PANORAMIC, FISHEYE, FISHEYE_EQUISOLID, distortion_params
(_radial_and_tangential_undistort, :795-890), pixtocam_ndc (convert_to_ndc, :50-111), z_range (rays_planes_intersection,
:1143-1164, 1291-1299), jitter with the offsets (dx, dy) handed over instead of drawn from a key (:943-957);
xnp=numpy (the dataset / eval path casts with numpy).
"""
from __future__ import annotations

import numpy as np


def get_pixtocam(focal, width, height):
    """camera_utils.get_pixtocam (:760-763): inverse of intrinsic_matrix(f, f, w/2, h/2)."""
    camtopix = np.array([[focal, 0, width * 0.5], [0, focal, height * 0.5], [0, 0, 1.0]])
    return np.linalg.inv(camtopix)


def _residual_and_jacobian(x, y, xd, yd, k1, k2, k3, k4, p1, p2):
    """camera_utils._compute_residual_and_jacobian (:795-841)."""
    r = x * x + y * y
    d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
    fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
    fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
    d_r = k1 + r * (2.0 * k2 + r * (3.0 * k3 + r * 4.0 * k4))
    d_x = 2.0 * x * d_r
    d_y = 2.0 * y * d_r
    fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
    fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
    fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
    fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
    return fx, fy, fx_x, fx_y, fy_x, fy_y


def radial_and_tangential_undistort(xd, yd, k1=0, k2=0, k3=0, k4=0, p1=0, p2=0, eps=1e-9, max_iterations=10):
    """camera_utils._radial_and_tangential_undistort (:844-890): Newton steps from the distorted point.  The
    coefficients take the dtype of the coordinates (the C ABI carries them as float32)."""
    dt = np.asarray(xd).dtype.type
    k1, k2, k3, k4, p1, p2 = (dt(v) for v in (k1, k2, k3, k4, p1, p2))
    x, y = np.copy(xd), np.copy(yd)
    for _ in range(max_iterations):
        fx, fy, fx_x, fx_y, fy_x, fy_y = _residual_and_jacobian(x, y, xd, yd, k1, k2, k3, k4, p1, p2)
        denominator = fy_x * fx_y - fx_x * fy_y
        x_numerator = fx * fy_y - fy * fx_y
        y_numerator = fy * fx_x - fx * fy_x
        ok = np.abs(denominator) > eps
        safe = np.where(ok, denominator, np.ones_like(denominator))
        x = x + np.where(ok, x_numerator / safe, np.zeros_like(denominator))
        y = y + np.where(ok, y_numerator / safe, np.zeros_like(denominator))
    return x, y


def convert_to_ndc(origins, directions, pixtocam, near=1.0):
    """camera_utils.convert_to_ndc (:50-111)."""
    t = -(near + origins[..., 2]) / directions[..., 2]
    origins = origins + t[..., None] * directions
    dx, dy, dz = np.moveaxis(directions, -1, 0)
    ox, oy, oz = np.moveaxis(origins, -1, 0)
    xmult = 1.0 / pixtocam[0, 2]
    ymult = 1.0 / pixtocam[1, 2]
    origins_ndc = np.stack([xmult * ox / oz, ymult * oy / oz, -np.ones_like(oz)], axis=-1)
    infinity_ndc = np.stack([xmult * dx / dz, ymult * dy / dz, np.ones_like(oz)], axis=-1)
    return origins_ndc, infinity_ndc - origins_ndc


def pixels_to_rays(pix_x_int, pix_y_int, pixtocam, camtoworld, dtype=np.float32, camtype="perspective",
                   distortion_params=None, pixtocam_ndc=None, pix_jitter=None):
    pix_x_int = np.asarray(pix_x_int)
    pix_y_int = np.asarray(pix_y_int)
    pixtocam = np.asarray(pixtocam, dtype)
    camtoworld = np.asarray(camtoworld, dtype)

    def pix_to_dir(x, y):
        return np.stack([x + 0.5, y + 0.5, np.ones_like(x)], axis=-1)

    # (:943-966) dx = dy = 0.0 without jitter; with it the reference adds the offsets to the INTEGER coordinates (+ 1 for
    # the neighbours first), the sum is float32
    dxj, dyj = (dtype(0.0), dtype(0.0)) if pix_jitter is None else (np.asarray(pix_jitter[0], dtype), np.asarray(pix_jitter[1], dtype))
    px0, px1 = pix_x_int.astype(dtype) + dxj, (pix_x_int + 1).astype(dtype) + dxj
    py0, py1 = pix_y_int.astype(dtype) + dyj, (pix_y_int + 1).astype(dtype) + dyj
    stacked = np.stack([pix_to_dir(px0, py0), pix_to_dir(px1, py0), pix_to_dir(px0, py1)], axis=0)
    mat_vec_mul = lambda A, b: np.matmul(A, b[..., None])[..., 0]
    cam_dirs = mat_vec_mul(pixtocam, stacked)
    if distortion_params is not None:
        x, y = radial_and_tangential_undistort(cam_dirs[..., 0], cam_dirs[..., 1], **distortion_params)
        cam_dirs = np.stack([x, y, np.ones_like(x)], -1)
    if camtype in ("fisheye", "fisheye_equisolid"):
        r = np.sqrt(np.sum(np.square(cam_dirs[..., :2]), axis=-1))
        theta = np.minimum(np.pi, r).astype(dtype) if camtype == "fisheye" else 2.0 * np.arcsin(r / 2.0)
        s_over_r = np.sin(theta) / r
        cam_dirs = np.stack([cam_dirs[..., 0] * s_over_r, cam_dirs[..., 1] * s_over_r, np.cos(theta)], axis=-1).astype(dtype)
    elif camtype == "pano":
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
    if pixtocam_ndc is None:
        dx_norm = np.linalg.norm(dx - directions, axis=-1)
        dy_norm = np.linalg.norm(dy - directions, axis=-1)
    else:
        p_ndc = np.asarray(pixtocam_ndc, dtype)
        origins_dx, _ = convert_to_ndc(origins, dx, p_ndc)
        origins_dy, _ = convert_to_ndc(origins, dy, p_ndc)
        origins, directions = convert_to_ndc(origins, directions, p_ndc)
        dx_norm = np.linalg.norm(origins_dx - origins, axis=-1)
        dy_norm = np.linalg.norm(origins_dy - origins, axis=-1)
    radii = (0.5 * (dx_norm + dy_norm))[..., None] * 2 / np.sqrt(12)
    return dict(origins=origins, directions=directions, viewdirs=viewdirs, radii=radii.astype(dtype), imageplane=imageplane,
                look=look, up=up)


def rays_planes_intersection(z_min, z_max, origins, directions):
    """camera_utils.rays_planes_intersection (:1143-1164)."""
    t1 = (z_min - origins[..., 2]) / directions[..., 2]
    t2 = (z_max - origins[..., 2]) / directions[..., 2]
    return np.minimum(t1, t2), np.maximum(t1, t2)


def cast_ray_batch(pixtocam, camtoworld, light, pix_x_int, pix_y_int, near, far, dtype=np.float32, camtype="perspective",
                   distortion_params=None, pixtocam_ndc=None, z_range=None, pix_jitter=None):
    """cast_ray_batch for one camera: rays + lights = lights[cam_idx], cam_origins = origins, near / far from Pixels."""
    r = pixels_to_rays(pix_x_int, pix_y_int, pixtocam, camtoworld, dtype, camtype, distortion_params, pixtocam_ndc, pix_jitter)
    shape = r["directions"].shape
    if z_range is not None:                                  # :1291-1299
        origins, directions = r["origins"], r["directions"]
        t_min, t_max = rays_planes_intersection(dtype(z_range[0]), dtype(z_range[1]), origins, directions)
        t_min = np.broadcast_to(t_min[..., None], origins.shape)
        t_max = np.broadcast_to(t_max[..., None], origins.shape)
        hit_mask = t_max < t_min
        r["origins"] = np.where(hit_mask, origins, origins + directions * t_min)
        r["directions"] = np.where(hit_mask, directions, directions * (t_max - t_min))
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
