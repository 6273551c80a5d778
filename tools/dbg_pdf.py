import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import nrc_amd, common
from nrc_amd import rc_ext
from oracle import material_ref
cfg = nrc_amd.hotdog_config()
wn = common.weights_material_np(True)
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(wn)
n = 128
rays = nrc_amd.synthetic_rays(n, seed=77)
rnd = material_ref.draw_randoms(cfg, n, seed=3)
ref = material_ref.material_forward(common.to_torch(wn), cfg, common.rays_torch(rays), rnd)
picks = dict(resample_inds=ref["inds"][:, 0].numpy(), spec_resample_inds=ref["debug"]["specular"]["inds"].numpy(), diff_resample_inds=ref["debug"]["diffuse"]["inds"].numpy())
rnd_p = dict(rnd, **picks)
ref64 = material_ref.material_forward(common.to_torch(wn, torch.float64), cfg, common.rays_torch(rays, torch.float64), rnd_p)
rnd2 = dict(rnd_p, gumbel=None, spec_gumbel=None, diff_gumbel=None)
rnd2 = {k: (v.astype(np.int32) if "inds" in k else v) for k, v in rnd2.items()}
rc.render_material(rays.hot_fields(), rnd2); torch.cuda.synchronize()
smp = rc.workspace("sec_samples").reshape(n, 32, 5)
out = dict(smp=smp)
for nm in ("specular", "diffuse"):
    for k in ("pdf", "weight", "local_lightdirs"):
        out[f"{nm}_{k}_32"] = ref["debug"][nm][k].numpy(); out[f"{nm}_{k}_64"] = ref64["debug"][nm][k].numpy()
out["rough"] = ref["material"]["roughness"].numpy()
np.savez("gpurun_out/dbg_pdf.npz", **out)
