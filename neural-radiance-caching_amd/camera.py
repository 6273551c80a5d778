"""Host mirror of the reference's ray generation for pinhole cameras, running on the GPU.

    cast_ray_batch(rc, camera, pixels) -> Rays   camera_utils.cast_ray_batch   internal/camera_utils.py:1225-1329
    get_pixtocam(focal, width, height)           camera_utils.get_pixtocam     internal/camera_utils.py:760-763
    render_camera(model, camera, height, width)  trainer.render_primary_rays   engine/trainer.py:812-846 (pose in, image out)

ProjectionType.PERSPECTIVE (the BASELINE scenes), PANORAMIC, FISHEYE, FISHEYE_EQUISOLID; optional radial + tangential
distortion and NDC rays (camera_utils.py:795-890, 50-111), z_range cropping (:1291-1299) and pixel jitter offsets handed
over as arrays (:943-957).
The returned Rays hold torch cuda tensors (nothing crosses PCIe but the 3x3 + 3x4 matrices).
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np

from .rays import Rays


def get_pixtocam(focal: float, width: int, height: int) -> np.ndarray:
    camtopix = np.array([[focal, 0, width * 0.5], [0, focal, height * 0.5], [0, 0, 1.0]])
    return np.linalg.inv(camtopix)


@dataclasses.dataclass
class Camera:
    """One entry of the reference's `cameras` tuple (pixtocams[i], camtoworlds[i]) + its light and depth range."""

    pixtocam: np.ndarray          # [3, 3]
    camtoworld: np.ndarray        # [3, 4]
    light: Optional[np.ndarray] = None   # [3]; default: the camera centre (datasets.py:1348)
    near: float = 2.0
    far: float = 6.0
    camtype: str = "perspective"       # ProjectionType value: "perspective", "pano", "fisheye" or "fisheye_equisolid"
    distortion_params: Optional[dict] = None   # {"k1", "k2", "k3", "k4", "p1", "p2"} (camera_utils.py:981-989)
    pixtocam_ndc: Optional[np.ndarray] = None  # [3, 3]: rays in NDC space (camera_utils.py:1052-1066)
    z_range: Optional[tuple] = None            # (z_min, z_max): rays cropped to that slab (camera_utils.py:1291-1299)


def cast_spherical_rays(rc, camtoworld, height: int, width: int, near: float, far: float, light=None) -> Rays:
    """camera_utils.cast_spherical_rays (internal/camera_utils.py:1415-1443): the full [height, width] panorama around
    `camtoworld`, pixtocam = diag(2 pi / width, pi / height, 1), ProjectionType.PANORAMIC."""
    p2c = np.diag(np.array([2.0 * np.pi / width, np.pi / height, 1.0]))
    cam = Camera(pixtocam=p2c, camtoworld=np.asarray(camtoworld)[:3, :4], light=light, near=near, far=far, camtype="pano")
    return rc.cast_rays(cam, rect=(0, 0, width, height))


def cast_ray_batch(rc, camera: Camera, pix_x_int=None, pix_y_int=None, rect=None, pix_jitter=None) -> Rays:
    """Rays for an explicit pixel batch (int arrays of one shape) or for rect = (x0, y0, width, height).  pix_jitter =
    (dx, dy): the sub-pixel offsets camera_utils.pixels_to_rays draws for jitter > 0 (explicit random inputs)."""
    return rc.cast_rays(camera, pix_x_int, pix_y_int, rect, pix_jitter)


def render_camera(model, camera: Camera, height: int, width: int, passes=("cache",), rows_per_chunk: Optional[int] = None,
                  keys=("rgb", "acc", "distance_median"), streams: int = 1, to_host: bool = True, rng=None):
    """Image of one camera without a host-side ray batch: rows of pixels are cast on the device and rendered
    chunk by chunk; returns {key: [H, W, ...]} (numpy, or cuda tensors with to_host=False).  Default chunk: whole rows
    covering >= 16 384 rays (one fused launch each; an 800 x 800 image takes 78-83 ms on one MI355X, 7.7-8.2 M rays/s).
    Chunks are independent: with streams > 1 they alternate between HIP streams (pays off for small chunks only).
    rng: None (deterministic sampling branch) or a uint32[2] key (prng.PRNGKey): the per-ray jitter is then generated
    in HBM (rc_prng_fill) from the keys the reference's render_image -> render_eval_fn -> model chain would derive
    for each chunk (models.py:2445, train_utils.py:3794-3818; prng.py), nothing is uploaded."""
    import torch

    from . import prng
    if rng is not None:
        rng = prng.as_key(rng)
    levels = [lvl[2] for lvl in model.config.sampling_strategy]

    chunk = model.config.render_chunk_size
    rows = rows_per_chunk or max(1, max(chunk, 16384) // width)
    dev = torch.device(f"cuda:{model.device}")
    pool = [torch.cuda.Stream(device=dev) for _ in range(max(1, streams))] if streams > 1 else [torch.cuda.current_stream(dev)]
    start = torch.cuda.Event()
    start.record(torch.cuda.current_stream(dev))
    out = {}
    for i, y0 in enumerate(range(0, height, rows)):
        hgt = min(rows, height - y0)
        st = pool[i % len(pool)]
        if i < len(pool):
            st.wait_event(start)
        with torch.cuda.stream(st):
            rays = cast_ray_batch(model.rc, camera, rect=(0, y0, width, hgt))
            randoms = None
            if rng is not None:
                apply_key, rng = prng.random_split(rng)
                rng, _ = prng.random_split(rng)                 # the key render_eval_fn hands back for the next chunk
                s_key = prng.cache_keys(prng.model_cache_rng(apply_key))["sampler"]
                jit = []
                for _ in levels:                                # sampling.py:341 / :408
                    k, s_key = prng.random_split(s_key)
                    jit.append(model.rc.prng_fill(k, (hgt * width,), "uniform"))
                    _, s_key = prng.random_split(s_key)
                randoms = {"jitter": jit}
            r = model.apply(None, randoms, rays, passes=passes)["render"]
            for k in keys:
                v = r[k]
                if k not in out:
                    # first chunk: allocate the image on this stream; later chunks on other streams only write rows
                    out[k] = torch.empty((height, width) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
                    if len(pool) > 1:
                        done = torch.cuda.Event()
                        done.record(st)
                        for other in pool:
                            other.wait_event(done)
                out[k][y0:y0 + hgt] = v.reshape((hgt, width) + tuple(v.shape[1:]))
    torch.cuda.synchronize(dev)
    return {k: v.cpu().numpy() for k, v in out.items()} if to_host else out
