import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import nrc_amd, common
from nrc_amd import rc_ext
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(common.weights_np())
n = 256
rays = nrc_amd.synthetic_rays(n)
out = rc.render_rays(rays.hot_fields(), None)
torch.cuda.synchronize()
ref32 = common.oracle_cache(n)
ref64 = common.oracle_cache(n, dtype=torch.float64)
ng = rc.workspace("normals_grad").reshape(3, n, 32).transpose(1, 2, 0)
for nm, ref in (("fp32", ref32), ("fp64", ref64)):
    rn = ref["sampler"][2]["normals"].numpy()
    d = np.abs(ng - rn)
    print(nm, "per-sample normals: max", d.max(), "mean", d.mean(), "frac>1e-2", (d > 1e-2).mean())
    print(nm, "rendered normals maxdiff", np.abs(out["normals"].cpu().numpy() - ref["render"]["normals"].numpy()).max())
rn32, rn64 = ref32["sampler"][2]["normals"].numpy(), ref64["sampler"][2]["normals"].numpy()
print("oracle fp32 vs fp64 per-sample: max", np.abs(rn32 - rn64).max(), "mean", np.abs(rn32-rn64).mean())
print("rgb", np.abs(out["rgb"].cpu().numpy() - ref32["render"]["rgb"].numpy()).max())
