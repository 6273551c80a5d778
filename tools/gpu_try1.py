import sys, time
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
from oracle import cache_ref, hashgrid_ref, mathx, stepfun_ref
import common
cfg = nrc_amd.hotdog_config()
W = common.weights_np()
rc = rc_ext.RadianceCache(cfg, 0)
t=time.time(); rc.load_weights(W); print("load", time.time()-t)
Wt = common.weights_torch()
# 1. hashgrid
rng = np.random.default_rng(0)
pts = rng.uniform(-5, 5, size=(4096, 3)).astype(np.float32)
for gid, (pref, g) in enumerate([("params/Cache/Sampler/MLP_0/density_grid", cfg.proposal_grids[0]), ("params/Cache/Sampler/MLP_1/density_grid", cfg.proposal_grids[1]), ("params/Cache/Sampler/MLP_2/density_grid", cfg.proposal_grids[2]), ("params/Cache/Shader/appearance_grid", cfg.appearance_grid)]):
    ref = hashgrid_ref.hash_encoding(Wt, pref, g, mathx.contract_radius(torch.from_numpy(pts), 2.0))
    out = rc.hashgrid_lookup(gid, pts).cpu()
    print("grid", gid, "maxdiff", float((out-ref).abs().max()), "ref absmax", float(ref.abs().max()))
# 2. sample_intervals
P, S, n = 64, 32, 512
t = np.sort(rng.uniform(size=(n, P+1)).astype(np.float32), axis=-1); t[:,0]=0; t[:,-1]=1
lg = rng.normal(size=(n,P)).astype(np.float32)*3
jit = rng.uniform(size=(n,1)).astype(np.float32)
for J in (None, jit):
    ref = stepfun_ref.sample_intervals(None if J is None else torch.from_numpy(J), torch.from_numpy(t), torch.from_numpy(lg), S)
    out = rc.sample_intervals(t, lg, S, J).cpu()
    print("sample_intervals jitter", J is not None, float((out-ref).abs().max()))
# 3. full render
for js in (None, 7):
    n = 256
    rays = nrc_amd.synthetic_rays(n)
    ref = common.oracle_cache(n, jitter_seed=js)
    rnd = None if js is None else {"jitter": common.jitters(n, seed=js)}
    out = rc.render_rays(rays.hot_fields(), rnd)
    torch.cuda.synchronize()
    for l in range(3):
        for k in ("sdist", "tdist"):
            a = rc.workspace(f"{k}{l}").reshape(n, -1); b = ref["sampler"][l][k].numpy()
            print(l, k, np.abs(a-b).max())
        S = cfg.sampling_strategy[l][2]
        a = rc.workspace(f"density{l}").reshape(n, S); b = ref["sampler"][l]["density"].numpy()
        print(l, "density", np.abs(a-b).max(), "rel", (np.abs(a-b)/(np.abs(b)+1e-6)).max())
        a = rc.workspace(f"weights{l}").reshape(n, S); b = ref["sampler"][l]["weights"].numpy()
        print(l, "weights", np.abs(a-b).max())
    sh = rc.workspace("shade").reshape(15, n, 32)
    print("shade rgb", np.abs(np.moveaxis(sh[0:3],0,-1) - ref["shader"]["rgb"].numpy()).max())
    R = ref["render"]
    for k, v in out.items():
        if k in R:
            b = R[k].reshape(v.shape).numpy()
            print(f"{k:28s} maxdiff {np.abs(v.cpu().numpy()-b).max():.3e}  refmax {np.abs(b).max():.3e}")
rc.set_profiling(True)
rays = nrc_amd.synthetic_rays(1024)
for i in range(3):
    out = rc.render_rays(rays.hot_fields(), None, outputs=["rgb","acc"])
torch.cuda.synchronize()
print(rc.stage_times_ms())
t=time.time()
for i in range(20): out = rc.render_rays(rays.hot_fields(), None, outputs=["rgb","acc"])
torch.cuda.synchronize(); dt=(time.time()-t)/20
print("ms per 1024 rays (incl host overhead)", dt*1e3, "rays/s", 1024/dt)
