import os, sys
sys.path.insert(0, "."); 
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
from nrc_amd.model import _CACHE_DEVICE_KEYS
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(nrc_amd.synthetic_weights(cfg))
keys = list(_CACHE_DEVICE_KEYS)
def batch(n, seed):
    r = nrc_amd.synthetic_rays(n, seed=seed)
    f = {k: torch.from_numpy(np.asarray(v)).cuda().contiguous() for k, v in r.hot_fields().items()}
    f.pop("lossmult", None); return f
for n in (128, 256, 512, 1024, 2048, 4096, 16384):
    B = [batch(n, 100 + i) for i in range(16)]
    for mode in (3, 1):
        rc.set_fused(mode)
        out = rc.render_rays(B[0], None, outputs=keys)
        for i in range(30): rc.render_rays(B[i % 16], None, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        e0.record()
        for i in range(reps): rc.render_rays(B[i % 16], None, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"n={n:6d} mode {mode}: {us:9.2f} us/launch  {n/us:6.2f} M rays/s", flush=True)
