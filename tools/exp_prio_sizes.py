"""RC_FUSED_PRIO (priority scheme of k_cache_fused_team) by batch size: us per launch."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
from nrc_amd.model import _CACHE_DEVICE_KEYS
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(nrc_amd.synthetic_weights(cfg))
keys = list(_CACHE_DEVICE_KEYS)
for n in (512, 1024, 2048, 4096, 16384):
    B = []
    for i in range(8):
        r = nrc_amd.synthetic_rays(n, seed=100 + i)
        f = {k: torch.from_numpy(np.asarray(v)).cuda().contiguous() for k, v in r.hot_fields().items()}
        f.pop("lossmult", None); B.append(f)
    out = rc.render_rays(B[0], None, outputs=keys)
    for i in range(20): rc.render_rays(B[i % 8], None, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for i in range(reps): rc.render_rays(B[i % 8], None, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"prio={os.environ.get('RC_FUSED_PRIO', 'default')} n={n:6d}: {e0.elapsed_time(e1) / reps * 1e3:9.2f} us/launch", flush=True)
