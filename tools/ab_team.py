"""The two forms of the fused cache kernel side by side: rc_set_fused 1 (two wavefronts per ray, rc_fused2.hip) against
3 (one wavefront per ray, rc_fused.hip) and 2 (launch-per-stage): bitwise comparison of every output on a few batch
sizes, then the time per 1024-ray launch (64 distinct batches, HIP events).  RC_FUSED_STAGGER=<cycles> starts the second
half of the team kernel's grid late (experiment)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
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
    f.pop("lossmult", None)
    return f


if "--time-only" not in sys.argv:
    for n in (1, 2, 3, 64, 257, 1024, 2049):
        f = batch(n, 7 + n)
        jit = {"jitter": [torch.rand(n, device="cuda") * 0.01 for _ in range(3)]} if n == 257 else None
        res = {}
        for mode in (1, 3, 2):
            rc.set_fused(mode)
            out = rc.render_rays(f, jit, outputs=keys)
            torch.cuda.synchronize()
            res[mode] = {k: v.clone() for k, v in out.items()}
        bad13 = [k for k in keys if not torch.equal(res[1][k], res[3][k])]
        bad12 = [k for k in keys if not torch.equal(res[1][k], res[2][k])]
        worst = max((float((res[1][k] - res[3][k]).abs().max()) for k in keys), default=0.0)
        nan = [k for k in keys if not bool(torch.isfinite(res[1][k]).all())]
        print(f"n={n:5d}: team vs one-wave differs on {bad13} (max abs {worst:.3e}); vs staged differs on {bad12}; non-finite {nan}", flush=True)

n = 1024
reps = 400
B = [batch(n, 100 + i) for i in range(64)]
for mode in (3, 1, 3, 1):
    rc.set_fused(mode)
    out = rc.render_rays(B[0], None, outputs=keys)
    for i in range(50): rc.render_rays(B[i % 64], None, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): rc.render_rays(B[i % 64], None, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"mode {mode} ({'team, 2 waves/ray' if mode == 1 else 'one wave/ray'}): {us:8.2f} us/launch  ({n / us:.2f} M rays/s)  stagger={os.environ.get('RC_FUSED_STAGGER', '0')}", flush=True)
