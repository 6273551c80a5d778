"""800x800 image from a pose: fused plan against the launch-per-stage plan at large chunk sizes."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import nrc_amd
from nrc_amd import model as M
cfg = nrc_amd.hotdog_config()
m = M.Model(cfg, 0); m.load_variables(nrc_amd.synthetic_weights(cfg))
o = np.array([0.0, -3.5, 2.0]); look = -o / np.linalg.norm(o); right = np.cross(look, [0, 0, 1.0]); right /= np.linalg.norm(right); up = np.cross(right, look)
c2w = np.concatenate([np.stack([right, up, -look], 1), o[:, None]], 1)
cam = nrc_amd.Camera(nrc_amd.get_pixtocam(1111.0, 800, 800), c2w, near=2.0, far=6.0)
for fused in (True, False):
    m.rc.set_fused(fused)
    for rows in (20, 80, 200, 800):
        nrc_amd.render_camera(m, cam, 800, 800, rows_per_chunk=rows, to_host=False)
        t0 = time.perf_counter()
        img = nrc_amd.render_camera(m, cam, 800, 800, rows_per_chunk=rows, to_host=False)
        dt = time.perf_counter() - t0
        print(f"fused {fused} rows/chunk {rows:4d} ({rows*800:6d} rays): {dt*1e3:7.1f} ms  {640000/dt/1e6:5.2f} M rays/s  acc {float(img['acc'].mean()):.3f}", flush=True)
