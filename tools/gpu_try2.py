import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import nrc_amd, common
from nrc_amd import rc_ext
from oracle import cache_ref
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(common.weights_np())
n = 512
rays, rnd = common.secondary_case(n, seed=5)
ref = cache_ref.cache_forward(common.weights_torch(), cfg, common.rays_dict_torch(rays), [torch.from_numpy(j)[:, None] for j in rnd["jitter"]],
                              is_secondary=True, gumbel=torch.from_numpy(rnd["gumbel"]), want_grad_normals=False)
out = rc.render_rays(rays, rnd, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_SECONDARY)
torch.cuda.synchronize()
inds = rc.workspace("inds", np.int32)[:n]
print("inds mismatch", int((inds != ref["filtered_sampler_inds"][:, 0].numpy()).sum()), "of", n)
for l in range(3):
    a = rc.workspace(f"tdist{l}").reshape(n, -1); b = ref["sampler"][l]["tdist"].numpy()
    print(l, "tdist", np.abs(a-b).max(), "tmax", b.max())
R_ = ref["render"]
for k in ("rgb", "acc", "env_map_rgb", "diffuse_rgb", "specular_rgb", "distance_median", "means"):
    v = out[k].cpu().numpy(); b = R_[k].numpy().reshape(v.shape)
    print(f"{k:20s} maxdiff {np.abs(v-b).max():.3e} refmax {np.abs(b).max():.3e}")
print("rgb_no_env", np.abs(out["rgb_no_env"].cpu().numpy() - (R_["rgb_no_stopgrad"] - R_["env_map_rgb"]*(1-R_["acc"][:,None])).numpy()).max())
