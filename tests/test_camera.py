"""On-device ray generation: oracle KATs on the CPU, parity of rc_cast_rays on the GPU."""
import numpy as np
import pytest
import torch

import nrc_amd
from oracle import camera_ref


def _lookat(origin):
    o = np.asarray(origin, np.float64)
    look = -o / np.linalg.norm(o)
    right = np.cross(look, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
    up = np.cross(right, look)
    # OpenGL camera: columns = right, up, -look
    return np.concatenate([np.stack([right, up, -look], axis=1), o[:, None]], axis=1)


def test_pixels_to_rays_known_answers():
    H, W, f = 8, 10, 12.5
    p2c = camera_ref.get_pixtocam(f, W, H)
    assert np.allclose(p2c @ np.array([W / 2, H / 2, 1.0]), [0, 0, 1])
    c2w = _lookat([0.0, -3.0, 4.0])
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    r = camera_ref.pixels_to_rays(xs, ys, p2c, c2w, np.float64)
    # a pixel whose centre is the principal point does not exist for even sizes; the four central ones straddle `look`
    centre = r["directions"][H // 2 - 1:H // 2 + 1, W // 2 - 1:W // 2 + 1].mean((0, 1))
    assert np.allclose(centre / np.linalg.norm(centre), -np.asarray([0.0, -3.0, 4.0]) / 5.0, atol=1e-12)
    assert np.allclose(np.linalg.norm(r["viewdirs"], axis=-1), 1.0)
    assert np.allclose(r["origins"], [0.0, -3.0, 4.0])
    assert np.allclose(r["look"][0, 0], c2w[:, 2] * -1) and np.allclose(r["up"][0, 0], c2w[:, 1])
    # image plane: x right, y up (OpenGL), one pixel = 1 / f
    assert np.allclose(r["imageplane"][0, 0], [(0.5 - W / 2) / f, -(0.5 - H / 2) / f])
    assert np.allclose(np.diff(r["imageplane"][..., 0], axis=1), 1 / f) and np.allclose(np.diff(r["imageplane"][..., 1], axis=0), -1 / f)
    # radii: neighbours are exactly one pixel (1 / f) away on the image plane, rotation keeps lengths
    assert np.allclose(r["radii"], (1 / f) * 2 / np.sqrt(12))
    # un-normalised pinhole directions: z (camera) component is -1 before the rotation
    cam_dirs = r["directions"] @ c2w[:, :3]
    assert np.allclose(cam_dirs[..., 2], -1.0)


def test_cast_ray_batch_fields():
    p2c = camera_ref.get_pixtocam(50.0, 6, 4)
    c2w = _lookat([1.0, 2.0, 2.0])
    r = camera_ref.cast_ray_batch(p2c, c2w, [0.5, 0.5, 3.0], np.array([0, 5]), np.array([3, 0]), 0.7, 4.0)
    assert r["origins"].shape == (2, 3) and r["near"].shape == (2, 1) and r["radii"].shape == (2, 1)
    assert np.allclose(r["lights"], [0.5, 0.5, 3.0]) and np.all(r["near"] == np.float32(0.7)) and r["cam_origins"] is r["origins"]
    assert r["directions"].dtype == np.float32


@pytest.mark.gpu
def test_cast_rays_on_device_matches_oracle():
    from nrc_amd import rc_ext
    rc = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
    H, W, f = 37, 53, 61.0
    cam = nrc_amd.Camera(nrc_amd.get_pixtocam(f, W, H), _lookat([2.0, -3.0, 1.5]), light=[2.1, -3.0, 1.6], near=2.0, far=6.0)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    ref = camera_ref.cast_ray_batch(cam.pixtocam, cam.camtoworld, cam.light, xs, ys, 2.0, 6.0)
    rect = rc.cast_rays(cam, rect=(0, 0, W, H))
    torch.cuda.synchronize()
    for k in ("origins", "directions", "viewdirs", "radii", "imageplane", "look", "up", "lights", "near", "far"):
        got = getattr(rect, k).cpu().numpy()
        assert got.shape == ref[k].shape, k
        assert np.abs(got - ref[k]).max() <= 2e-6 * max(1.0, np.abs(ref[k]).max()), k
    # explicit, unordered pixel batch and a sub-rectangle
    rng = np.random.default_rng(0)
    px, py = rng.integers(0, W, size=(5, 7)), rng.integers(0, H, size=(5, 7))
    batch = rc.cast_rays(cam, px, py)
    sub = rc.cast_rays(cam, rect=(11, 5, 20, 9))
    torch.cuda.synchronize()
    assert torch.equal(batch.directions, rect.directions[torch.from_numpy(py), torch.from_numpy(px)])
    assert torch.equal(sub.directions, rect.directions[5:14, 11:31]) and torch.equal(sub.radii, rect.radii[5:14, 11:31])


@pytest.mark.gpu
def test_render_camera_equals_render_of_host_rays():
    """Pose in, image out: rays cast on the device give the image of the same rays uploaded from the host."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import common
    from nrc_amd import model as M
    H, W, f = 20, 24, 30.0
    cfg = nrc_amd.hotdog_config(render_chunk_size=128)
    m = M.Model(cfg, 0)
    m.load_variables(common.weights_np())
    cam = nrc_amd.Camera(nrc_amd.get_pixtocam(f, W, H), _lookat([0.0, -3.5, 2.0]), near=2.0, far=6.0)
    img = nrc_amd.render_camera(m, cam, H, W)
    assert img["rgb"].shape == (H, W, 3) and img["acc"].shape == (H, W)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    ref = camera_ref.cast_ray_batch(cam.pixtocam, cam.camtoworld, cam.camtoworld[:, 3], xs, ys, 2.0, 6.0)
    fields = {k: np.ascontiguousarray(ref[k].reshape(H * W, -1)) for k in ("origins", "directions", "viewdirs", "near", "far", "lights")}
    out = m.rc.render_rays(fields, None, outputs=["rgb", "acc"])
    torch.cuda.synchronize()
    assert np.abs(img["rgb"].reshape(-1, 3) - out["rgb"].cpu().numpy()).max() <= 1e-4      # ray fields agree to 2e-6
    assert np.abs(img["acc"].reshape(-1) - out["acc"].cpu().numpy()).max() <= 1e-4


def test_spherical_rays_known_answers():
    """cast_spherical_rays (camera_utils.py:1415-1443, :1013-1024): a full panorama of unit directions."""
    H, W = 6, 12
    c2w = np.concatenate([np.eye(3), [[1.0], [2.0], [3.0]]], axis=1)
    r = camera_ref.cast_spherical_rays(c2w, H, W, 0.1, 5.0, dtype=np.float64)
    d = r["directions"]
    assert d.shape == (H, W, 3) and np.allclose(np.linalg.norm(d, axis=-1), 1.0)
    # pixel (x, y): theta = 2 pi (x + 0.5) / W, phi = pi (y + 0.5) / H; camera (OpenCV) direction
    # (-sin phi sin theta, -cos phi, -sin phi cos theta), then diag(1, -1, -1) and the identity rotation
    x, y = 4, 1
    th, ph = 2 * np.pi * (x + 0.5) / W, np.pi * (y + 0.5) / H
    assert np.allclose(d[y, x], [-np.sin(ph) * np.sin(th), np.cos(ph), np.sin(ph) * np.cos(th)])
    # rows run from the +y pole to the -y pole, columns once around it; opposite columns are mirrored in x and z
    assert d[0, :, 1].min() > 0.9 and d[-1, :, 1].max() < -0.9
    assert np.allclose(d[:, :W // 2, 0], -d[:, W // 2:, 0]) and np.allclose(d[:, :W // 2, 2], -d[:, W // 2:, 2])
    assert np.allclose(r["origins"], [1.0, 2.0, 3.0]) and np.allclose(r["lights"], [1.0, 2.0, 3.0])
    assert np.all(r["near"] == 0.1) and np.all(r["far"] == 5.0)


@pytest.mark.gpu
def test_spherical_rays_on_device_match_oracle():
    from nrc_amd import rc_ext
    rc = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
    H, W = 33, 70
    c2w = _lookat([0.3, -0.2, 0.4])
    ref = camera_ref.cast_spherical_rays(c2w, H, W, 0.05, 2.0)
    got = nrc_amd.cast_spherical_rays(rc, c2w, H, W, 0.05, 2.0)
    torch.cuda.synchronize()
    for k in ("origins", "directions", "viewdirs", "radii", "imageplane", "lights", "near", "far"):
        g = getattr(got, k).cpu().numpy()
        assert g.shape == ref[k].shape, k
        assert np.abs(g - ref[k]).max() <= 4e-6 * max(1.0, np.abs(ref[k]).max()), k
    with pytest.raises(KeyError):
        rc.cast_rays(nrc_amd.Camera(np.eye(3), c2w, camtype="orthographic"), rect=(0, 0, 2, 2))


# ---- the rest of pixels_to_rays: distortion, fisheye, NDC (camera_utils.py:795-890, 991-1011, 50-111, 1052-1066)
DIST = dict(k1=0.08, k2=-0.03, k3=0.004, k4=0.0, p1=0.002, p2=-0.0015)


def test_undistortion_inverts_the_distortion_model():
    """The forward model (:808-816): xd = x d + 2 p1 x y + p2 (r + 2 x^2), yd likewise; the Newton undistortion takes
    the distorted point back."""
    rng = np.random.default_rng(3)
    x, y = rng.uniform(-0.6, 0.6, size=(2, 200))
    k1, k2, k3, k4, p1, p2 = (DIST[k] for k in ("k1", "k2", "k3", "k4", "p1", "p2"))
    r = x * x + y * y
    d = 1 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
    xd = x * d + 2 * p1 * x * y + p2 * (r + 2 * x * x)
    yd = y * d + 2 * p2 * x * y + p1 * (r + 2 * y * y)
    ux, uy = camera_ref.radial_and_tangential_undistort(xd, yd, **DIST)
    assert np.abs(ux - x).max() <= 1e-12 and np.abs(uy - y).max() <= 1e-12
    # no distortion: the identity, whatever the iteration count
    ux, uy = camera_ref.radial_and_tangential_undistort(xd, yd)
    assert np.array_equal(ux, xd) and np.array_equal(uy, yd)


def test_fisheye_and_ndc_known_answers():
    H, W, f = 8, 10, 7.0          # even sizes: no pixel centre on the optical axis (sin(theta) / r is 0 / 0 there, as in the reference)
    p2c = camera_ref.get_pixtocam(f, W, H)
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], axis=1)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    # equidistant fisheye: the angle between a ray and the optical axis (-z after the OpenGL flip) is the image-plane
    # radius over the focal length; equisolid: r = 2 sin(theta / 2)
    rad = np.hypot((xs + 0.5 - W / 2) / f, (ys + 0.5 - H / 2) / f)
    for camtype, theta in (("fisheye", np.minimum(np.pi, rad)), ("fisheye_equisolid", 2 * np.arcsin(rad / 2))):
        r = camera_ref.pixels_to_rays(xs, ys, p2c, c2w, np.float64, camtype=camtype)
        assert np.allclose(np.linalg.norm(r["directions"], axis=-1), 1.0)
        assert np.allclose(np.arccos(-r["directions"][..., 2]), theta, atol=1e-12), camtype
    # NDC (NeRF appendix C): origins land on the near plane z = -1, directions have z = 2, and
    # origin + direction = the projection of the point at infinity (xmult dx / dz, ymult dy / dz, 1)
    c2w = _lookat([0.2, -0.1, 3.0])
    c2w[:, :3] = np.eye(3)                                           # forward facing: the camera looks down -z
    r = camera_ref.pixels_to_rays(xs, ys, p2c, c2w, np.float64, pixtocam_ndc=p2c)
    plain = camera_ref.pixels_to_rays(xs, ys, p2c, c2w, np.float64)
    assert np.allclose(r["origins"][..., 2], -1.0) and np.allclose(r["directions"][..., 2], 2.0)
    d = plain["directions"]
    inf = np.stack([d[..., 0] / d[..., 2] / p2c[0, 2], d[..., 1] / d[..., 2] / p2c[1, 2], np.ones_like(d[..., 2])], -1)
    assert np.allclose(r["origins"] + r["directions"], inf)
    assert np.array_equal(r["viewdirs"], plain["viewdirs"])           # viewdirs are taken before the conversion
    assert np.all(r["radii"] > 0) and r["radii"].shape == (H, W, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["distortion", "fisheye", "fisheye_equisolid", "ndc", "distortion+ndc"])
def test_cast_rays_distortion_fisheye_ndc_on_device(case):
    from nrc_amd import rc_ext
    rc = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
    H, W, f = 30, 44, 40.0
    p2c = nrc_amd.get_pixtocam(f, W, H)
    c2w = _lookat([2.0, -3.0, 1.5])
    kw = {}
    if "fisheye" in case:
        kw["camtype"] = case
    if "distortion" in case:
        kw["distortion_params"] = DIST
    if "ndc" in case:
        c2w = np.concatenate([np.eye(3), [[0.2], [-0.1], [3.0]]], axis=1)
        kw["pixtocam_ndc"] = p2c
    cam = nrc_amd.Camera(p2c, c2w, light=[2.1, -3.0, 1.6], near=0.0, far=1.0, **kw)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    ref = camera_ref.cast_ray_batch(cam.pixtocam, cam.camtoworld, cam.light, xs, ys, 0.0, 1.0, camtype=kw.get("camtype", "perspective"),
                                    distortion_params=kw.get("distortion_params"), pixtocam_ndc=kw.get("pixtocam_ndc"))
    got = rc.cast_rays(cam, rect=(0, 0, W, H))
    torch.cuda.synchronize()
    for k in ("origins", "directions", "viewdirs", "radii", "imageplane", "look", "up", "lights", "near", "far"):
        g = getattr(got, k).cpu().numpy()
        assert g.shape == ref[k].shape, k
        # 2e-6 like the pinhole test; the radii of the NDC rays are differences of two NDC origins ~1 apart (cancellation)
        tol = 2e-5 if ("ndc" in case and k == "radii") else 4e-6
        assert np.abs(g - ref[k]).max() <= tol * max(1.0, np.abs(ref[k]).max()), (k, np.abs(g - ref[k]).max())


def test_z_range_and_pixel_jitter_known_answers():
    """cast_ray_batch(z_range=...) crops a ray to the slab between two z planes (camera_utils.py:1143-1164, 1291-1299):
    the new origin lies on the nearer plane, origin + direction on the farther one; viewdirs and radii are untouched.
    Pixel jitter offsets (:943-957) move a ray exactly like a fractional pixel coordinate would."""
    H, W, f = 6, 8, 9.0
    p2c = camera_ref.get_pixtocam(f, W, H)
    c2w = _lookat([0.5, -2.0, 5.0])
    xs, ys = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    plain = camera_ref.cast_ray_batch(p2c, c2w, [0, 0, 5.0], xs, ys, 2.0, 6.0, np.float64)
    crop = camera_ref.cast_ray_batch(p2c, c2w, [0, 0, 5.0], xs, ys, 2.0, 6.0, np.float64, z_range=(-1.0, 1.5))
    # every ray of this camera points downwards: it enters at z = 1.5 and leaves at z = -1
    assert (plain["directions"][..., 2] < 0).all()
    assert np.allclose(crop["origins"][..., 2], 1.5) and np.allclose((crop["origins"] + crop["directions"])[..., 2], -1.0)
    # the cropped ray is the same line: origin' - origin and direction' are parallel to direction
    for v in (crop["origins"] - plain["origins"], crop["directions"]):
        assert np.allclose(np.cross(v, plain["directions"]), 0.0, atol=1e-12)
    assert np.array_equal(crop["viewdirs"], plain["viewdirs"]) and np.array_equal(crop["radii"], plain["radii"])
    assert crop["cam_origins"] is crop["origins"]
    # jitter: (dx, dy) = (0.25, -0.5) on integer pixel (3, 2) = the direction through pixel coordinates (3.75, 2.0)
    dx, dy = np.full(xs.shape, 0.25), np.full(xs.shape, -0.5)
    jit = camera_ref.pixels_to_rays(xs, ys, p2c, c2w, np.float64, pix_jitter=(dx, dy))
    cam_dir = p2c @ np.array([3 + 0.25 + 0.5, 2 - 0.5 + 0.5, 1.0])
    want = c2w[:, :3] @ (cam_dir * np.array([1.0, -1.0, -1.0]))
    assert np.allclose(jit["directions"][2, 3], want, atol=1e-12)
    assert np.allclose(jit["radii"], plain["radii"])              # a pinhole's pixel footprint does not depend on the offset


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["z_range", "jitter", "jitter+z_range+distortion"])
def test_cast_rays_z_range_and_jitter_on_device(case):
    from nrc_amd import rc_ext
    rc = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
    H, W, f = 26, 38, 33.0
    p2c = nrc_amd.get_pixtocam(f, W, H)
    c2w = _lookat([1.0, -2.5, 4.0])
    kw = {}
    if "z_range" in case:
        kw["z_range"] = (-0.75, 1.25)
    if "distortion" in case:
        kw["distortion_params"] = DIST
    cam = nrc_amd.Camera(p2c, c2w, light=[1.0, -2.4, 4.1], near=0.0, far=1.0, **kw)
    rng = np.random.default_rng(12)
    ys, xs = rng.integers(0, H, size=(5, 61)), rng.integers(0, W, size=(5, 61))
    jit = (rng.uniform(-0.5, 0.5, xs.shape).astype(np.float32), (rng.normal(size=xs.shape) * 0.5).astype(np.float32)) if "jitter" in case else None
    ref = camera_ref.cast_ray_batch(cam.pixtocam, cam.camtoworld, cam.light, xs, ys, 0.0, 1.0, distortion_params=kw.get("distortion_params"),
                                    z_range=kw.get("z_range"), pix_jitter=jit)
    got = nrc_amd.cast_ray_batch(rc, cam, xs, ys, pix_jitter=jit)
    torch.cuda.synchronize()
    for k in ("origins", "directions", "viewdirs", "radii", "imageplane", "look", "up", "lights", "near", "far"):
        g = getattr(got, k).cpu().numpy()
        assert g.shape == ref[k].shape, k
        assert np.abs(g - ref[k]).max() <= 4e-6 * max(1.0, np.abs(ref[k]).max()), (k, np.abs(g - ref[k]).max())
    if jit is None:
        with pytest.raises(ValueError):
            rc.cast_rays(cam, xs, ys, pix_jitter=(np.zeros(3, np.float32), np.zeros(3, np.float32)))
