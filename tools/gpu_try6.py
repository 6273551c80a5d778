"""Scratch: transient HIP path vs oracle."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
from oracle import transient_ref
cfg = nrc_amd.cornell_transient_config()
w = nrc_amd.synthetic_weights(cfg)
rc = rc_ext.RadianceCache(cfg, 0)
rc.load_weights(w)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rays = nrc_amd.synthetic_transient_rays(n)
f = rays.hot_fields()
out = rc.render_transient(f, None)
torch.cuda.synchronize()
wt = {k: torch.from_numpy(v) for k, v in w.items()}
rt = {k: torch.from_numpy(np.asarray(v)) for k, v in f.items()}
ref = transient_ref.transient_forward(wt, cfg, rt, None)["render"]
alias = {"transient_indirect_diffuse": "transient_indirect_diffuse", "transient_indirect_specular": "transient_indirect_specular"}
for k, v in out.items():
    if k == "normals":
        continue
    r = ref[k].numpy()
    g = v.cpu().numpy().reshape(r.shape) if r.ndim != v.dim() or True else v.cpu().numpy()
    if r.shape != g.shape:
        r = np.broadcast_to(r, g.shape) if r.size != g.size else r.reshape(g.shape)
    d = np.abs(g - r)
    print(f"{k:32s} max|ref| {np.abs(r).max():10.4g}  max diff {d.max():10.3g}  mean diff {d.mean():10.3g}", flush=True)
t0 = time.time()
for _ in range(5):
    out = rc.render_transient(f, None, outputs=["rgb", "integrated_rgb"])
torch.cuda.synchronize()
print("ms per call", (time.time() - t0) / 5 * 1e3)
