"""PSNR protocol of SURVEY 8(d): one synthetic look-at camera rendered by the HIP path (render_camera: rays cast on the
device, fused kernel) and by the fp32 / fp64 oracle on the same pixels; PSNR = -10 log10(mean((a - b)^2)), max |d rgb|.
The oracle does ~3 k rays/s on 16 host threads: the default is a 160 x 160 crop-free image (25 600 rays)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import common, nrc_amd
from nrc_amd import model as M
from oracle import cache_ref

side = int(sys.argv[1]) if len(sys.argv) > 1 else 160
torch.set_num_threads(min(16, os.cpu_count() or 1))
cfg = nrc_amd.hotdog_config()
m = M.Model(cfg, 0)
m.load_variables(common.weights_np())
o = np.array([0.0, -3.5, 2.0]); look = -o / np.linalg.norm(o); right = np.cross(look, [0, 0, 1.0]); right /= np.linalg.norm(right)
up = np.cross(right, look)
c2w = np.concatenate([np.stack([right, up, -look], 1), o[:, None]], 1)
cam = nrc_amd.Camera(nrc_amd.get_pixtocam(1111.0 * side / 800.0, side, side), c2w, near=2.0, far=6.0)
img = nrc_amd.render_camera(m, cam, side, side)
rays = m.rc.cast_rays(cam, rect=(0, 0, side, side)).tree_map(lambda t: t.cpu().numpy().reshape(side * side, -1))
refs = {}
for name, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
    t0 = time.time()
    ref = cache_ref.cache_forward(common.weights_torch(dtype=dt), cfg, common.rays_torch(rays, dt), None, want_grad_normals=False)["render"]
    a = img["rgb"].reshape(-1, 3).astype(np.float64); b = ref["rgb"].double().numpy()
    refs[name] = b
    mse = float(np.mean((a - b) ** 2))
    print(f"{side}x{side} image, HIP vs {name} oracle: PSNR {-10 * np.log10(max(mse, 1e-300)):.1f} dB, max|d rgb| {np.abs(a - b).max():.3e}, "
          f"mean acc {float(img['acc'].mean()):.3f}  (oracle {time.time() - t0:.1f} s)", flush=True)
d = refs["fp32"] - refs["fp64"]
print(f"for scale, fp32 oracle vs fp64 oracle: PSNR {-10 * np.log10(max(float(np.mean(d ** 2)), 1e-300)):.1f} dB, max|d rgb| {np.abs(d).max():.3e}; "
      f"pixels with |HIP - fp64| > 1e-4: {int((np.abs(img['rgb'].reshape(-1, 3) - refs['fp64']).max(-1) > 1e-4).sum())} of {side * side}")
