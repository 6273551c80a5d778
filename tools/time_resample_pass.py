"""Time the launch-per-stage resampling pass (RC_PASS_CACHE | RC_PASS_RESAMPLE, rgb + acc) per launch plan and batch size:
rc_set_fused 2 = level kernels behind k_sample_level, 1 = sampling inside the level kernels (k_level_ray)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(nrc_amd.synthetic_weights(cfg))
for n in (256, 1024, 4096, 16384, 65536):
    r = nrc_amd.synthetic_rays(n, seed=3)
    f = {k: torch.from_numpy(np.asarray(v)).cuda().contiguous() for k, v in r.hot_fields().items()}
    rng = np.random.default_rng(1)
    rnd = {"jitter": [torch.from_numpy(rng.uniform(size=n).astype(np.float32)).cuda() for _ in range(3)],
           "gumbel": torch.from_numpy(rng.gumbel(size=(n, 32)).astype(np.float32)).cuda()}
    line = f"n={n:6d}"
    for mode in (2, 1):
        rc.set_fused(mode)
        out = rc.render_rays(f, rnd, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_RESAMPLE, outputs=["rgb", "acc"])
        for _ in range(10): rc.render_rays(f, rnd, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_RESAMPLE, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps): rc.render_rays(f, rnd, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_RESAMPLE, out=out)
        e1.record(); torch.cuda.synchronize()
        line += f"   mode {mode}: {e0.elapsed_time(e1) / reps * 1e3:8.1f} us"
    print(line, flush=True)
