"""Time k_cache_fused on 1024-ray batches (64 distinct batches, HIP events over many launches) for a few output sets.
RC_HIP_LIBRARY selects another build of the library (A/B)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
from nrc_amd.model import _CACHE_DEVICE_KEYS
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(nrc_amd.synthetic_weights(cfg))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
def batch(seed):
    r = nrc_amd.synthetic_rays(n, seed=seed)
    f = {k: torch.from_numpy(np.asarray(v)).cuda().contiguous() for k, v in r.hot_fields().items()}
    f.pop("lossmult", None); return f
B = [batch(100 + i) for i in range(64)]
sets = {"all": list(_CACHE_DEVICE_KEYS), "no_normals": [k for k in _CACHE_DEVICE_KEYS if k != "normals"], "rgb_only": ["rgb"]}
for name, keys in sets.items():
    out = rc.render_rays(B[0], None, outputs=keys)
    for i in range(50): rc.render_rays(B[i % 64], None, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): rc.render_rays(B[i % 64], None, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('RC_HIP_LIBRARY','product')[-40:]:40s} {name:12s} {e0.elapsed_time(e1) / reps * 1e3:8.2f} us/launch  ({n / (e0.elapsed_time(e1) / reps * 1e-3) / 1e6:.2f} M rays/s)", flush=True)
