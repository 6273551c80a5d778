"""models.render_image on the 800x800 image (BASELINE configs[3], one GPU) at the reference's documented chunk size and
at large chunks: device-complete time (all chunks enqueued and finished) and time including the unpack + D2H of all keys."""
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
rays = m.rc.cast_rays(cam, rect=(0, 0, 800, 800)).tree_map(lambda t: t.cpu().numpy())
fn = M.bind_render_fn(M.create_render_fn(m))
marks = {}
orig = M._ImageSink.finish
def finish(self, n):
    marks["enq"] = time.perf_counter()
    torch.cuda.synchronize(); marks["dev"] = time.perf_counter()
    r = orig(self, n); marks["fin"] = time.perf_counter(); return r
M._ImageSink.finish = finish
for chunk in (1024, 4096, 16000, 64000, 640000):
    c = nrc_amd.hotdog_config(render_chunk_size=chunk)
    for rep in range(3):
        t0 = time.perf_counter()
        img, _ = M.render_image(fn, None, rays, c, ("cache",), verbose=False)
        t1 = time.perf_counter()
    print(f"chunk {chunk:7d}: enqueue {1e3*(marks['enq']-t0):7.1f} ms  device-complete {1e3*(marks['dev']-t0):7.1f} ms "
          f"({640000/(marks['dev']-t0)/1e6:5.2f} M rays/s)  unpack+D2H {1e3*(marks['fin']-marks['dev']):6.1f} ms  total {1e3*(t1-t0):7.1f} ms", flush=True)
# device-resident rays (render_camera path) for comparison
for rows in (20, 800):
    nrc_amd.render_camera(m, cam, 800, 800, rows_per_chunk=rows, to_host=False)
    t0 = time.perf_counter(); nrc_amd.render_camera(m, cam, 800, 800, rows_per_chunk=rows, to_host=False); t1 = time.perf_counter()
    print(f"render_camera rows {rows}: {1e3*(t1-t0):.1f} ms", flush=True)
