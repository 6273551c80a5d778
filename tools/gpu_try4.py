import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import nrc_amd, common
from nrc_amd import rc_ext
from oracle import material_ref
cfg = nrc_amd.hotdog_config()
Wn = nrc_amd.synthetic_weights(cfg, passes=("cache", "material"), level_decay=float(os.environ.get("DECAY", "1.0")), table_range=float(os.environ.get("TR", "0.05")))
Wt = {k: torch.from_numpy(v) for k, v in Wn.items()}
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(Wn)
n = 128
rays = nrc_amd.synthetic_rays(n)
rnd = material_ref.draw_randoms(cfg, n, seed=3)
ref = material_ref.material_forward(Wt, cfg, common.rays_torch(rays), rnd)
cres, mres = rc.render_material(rays.hot_fields(), rnd)
torch.cuda.synchronize()
inds = rc.workspace("inds", np.int32)[:n]
same = inds == ref["inds"][:, 0].numpy()
print("inds same", same.mean())
mat = rc.workspace("m_mat").reshape(n, 5)
print("mat albedo", np.abs(mat[:, :3] - ref["material"]["albedo"][:, :].numpy().reshape(n, 3))[same].max(), "rough", np.abs(mat[:, 3] - ref["material"]["roughness"].numpy().reshape(n))[same].max(), "metal", np.abs(mat[:, 4] - ref["material"]["metalness"].numpy().reshape(n))[same].max())
vmf = rc.workspace("l_vmf").reshape(n, 128, 5)
rv = ref["vmfs"]
from oracle import mathx
rm = mathx.l2_normalize(rv["vmf_means"]).numpy()
print("vmf mean", np.abs(vmf[..., :3] - rm)[same].max(), "kappa", np.abs(vmf[..., 3] - rv["vmf_kappas"][..., 0].numpy())[same].max(), "w", np.abs(vmf[..., 4] - torch.softmax(rv["vmf_logits"][..., 0], -1).numpy())[same].max())
K = 32
smp = rc.workspace("sec_samples").reshape(n, K, 5)
for nm, sl in (("specular", slice(0, 16)), ("diffuse", slice(16, 32))):
    d = ref["debug"][nm]
    print(nm, "local dirs", np.abs(smp[:, sl, :3] - d["local_lightdirs"].numpy())[same].max(), "pdf", np.abs(smp[:, sl, 3] - d["pdf"][..., 0].numpy())[same].max(), "pdf max", d["pdf"].max().item(), "weight", np.abs(smp[:, sl, 4] - d["weight"][..., 0].numpy())[same].max())
sec_rgb = rc.workspace("sec_rgb").reshape(-1, 3); sec_acc = rc.workspace("sec_acc")
r_rgb = torch.cat([ref["debug"]["specular"]["rgb"], ref["debug"]["diffuse"]["rgb"]]).numpy()
r_acc = torch.cat([ref["debug"]["specular"]["acc"], ref["debug"]["diffuse"]["acc"]]).numpy()
sm2 = np.concatenate([np.repeat(same, 16), np.repeat(same, 16)])
print("sec rgb: max", np.abs(sec_rgb - r_rgb)[sm2].max(), "mean", np.abs(sec_rgb - r_rgb)[sm2].mean(), "acc", np.abs(sec_acc - r_acc)[sm2].max())
Rr = ref["render"]
for k, v in mres.items():
    if k in Rr:
        a = v.cpu().numpy(); b = Rr[k].numpy().reshape(a.shape)
        print(f"{k:26s} max {np.abs(a-b)[same].max():.3e} mean {np.abs(a-b)[same].mean():.3e} refmax {np.abs(b).max():.3e}")
print("cache rgb", np.abs(cres["rgb"].cpu().numpy() - Rr["cache_rgb"].numpy()).max())
