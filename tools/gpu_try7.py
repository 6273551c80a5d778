import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import nrc_amd, common
from nrc_amd import rc_ext
n = 24
cfg = nrc_amd.cornell_transient_config()
h = rc_ext.RadianceCache(cfg, 0)
h.load_weights(common.weights_transient_np(False, 6.0))
rays = nrc_amd.synthetic_transient_rays(n)
rnd = {"jitter": common.jitters(n, seed=5)}
out = {k: v.cpu().numpy() for k, v in h.render_transient(rays.hot_fields(), rnd).items()}
ref = common.oracle_transient(n, jitter_seed=5, density_shift=6.0)
r = {k: v.numpy() for k, v in ref["render"].items()}
sh = ref["shader"]
for k in ("transient_indirect_diffuse", "transient_indirect_specular", "transient_indirect_viz", "indirect_diffuse_rgb", "indirect_specular_rgb"):
    d = np.abs(out[k] - r[k].reshape(out[k].shape))
    print(f"{k:28s} max ref {np.abs(r[k]).max():9.4g} max diff {d.max():9.3g} worst ray {np.unravel_index(d.argmax(), d.shape)[0]}")
ray = 9
w = sh["weights"].numpy()[ray]
print("weights", w.round(3))
print("rdist", sh["ray_dists"].numpy()[ray, :, 0].round(3))
print("ldist", sh["light_dists"].numpy()[ray, :, 0].round(4))
tshade = h.workspace("tshade").reshape(19, n, 32)
print("hip ldist", tshade[16, ray].round(4))
a, b = out["transient_indirect_viz"][ray, :, 0], r["transient_indirect_viz"][ray, :, 0]
nz = np.nonzero(np.abs(a - b) > 1e-4)[0]
print("bins differing", nz[:10], nz[-10:] if len(nz) else None, "hip", a[nz[:5]], "ref", b[nz[:5]])
