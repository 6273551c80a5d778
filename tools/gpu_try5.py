import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import common, nrc_amd
from nrc_amd import rc_ext
rc = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
rc.load_weights(common.weights_np())
for n in (1, 2, 4, 5, 256):
    rays = nrc_amd.synthetic_rays(n)
    for outs in (None, ["rgb", "acc"]):
        res = {}
        for fused in (True, False):
            rc.set_fused(fused)
            o = rc.render_rays(rays.hot_fields(), None, outputs=outs) if outs else rc.render_rays(rays.hot_fields(), None)
            torch.cuda.synchronize()
            res[fused] = o["rgb"].cpu().numpy()
        print(n, "all" if outs is None else "rgb", np.abs(res[True] - res[False]).max(), flush=True)
