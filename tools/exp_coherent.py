"""Throughput of the fused cache kernel on coherent camera rays (scanline chunk vs 32x32 tile) vs random rays."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(nrc_amd.synthetic_weights(cfg))
cam = nrc_amd.synthetic_camera_rays(800, 800)
def fields(sel):
    f = {k: np.asarray(v)[sel].reshape(1024, -1) for k, v in cam.hot_fields().items()}
    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in f.items()}
cases = {
    "random (bench)": {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in nrc_amd.synthetic_rays(1024).hot_fields().items()},
    "scanline 1024 px": fields((slice(400, 402), slice(0, 512))),
    "tile 32x32 px": fields((slice(384, 416), slice(384, 416))),
}
for name, f in cases.items():
    out = rc.render_rays(f, None)
    for _ in range(20): rc.render_rays(f, None, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): rc.render_rays(f, None, out=out)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 200 * 1e3
    print(f"{name:20s} {ms*1e3:8.1f} us/batch  {1024/ms/1e3:6.2f} M rays/s  acc mean {float(out['acc'].mean()):.3f}")
