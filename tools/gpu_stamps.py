"""Diagnostic: per-segment cycle shares of the cache-shader kernel (tools/diag/librc_hip.so, -DRC_STAMPS)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
rc_ext.library_path = lambda: os.path.join(R, "tools", "diag", "librc_hip.so")
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(nrc_amd.synthetic_weights(cfg))
rc.set_graph_mode(0)
rays = nrc_amd.synthetic_rays(1024)
for _ in range(5): rc.render_rays(rays.hot_fields(), None, outputs=["rgb"])
torch.cuda.synchronize()
d = rc.workspace("debug").view(np.uint64)[: 1024 * 10].reshape(1024, 10)
st = d[:, :7].astype(np.int64); seg = np.diff(st, axis=1)
names = ["ws_begin+stage", "heads(245)", "IDE", "s0(808)", "ibrdf(229)", "trunk(841)"]
mf = [0, 245, 0, 808, 229, 841]
print("median cycles per segment:")
for i, nme in enumerate(names):
    med = np.median(seg[:, i]); print(f"  {nme:16s} {med:9.0f}  ideal {mf[i]*64:7d}  eff {mf[i]*64/med if med else 0:.2f}")
tot = st[:, 6] - st[:, 0]; rt = (d[:, 8].astype(np.int64) - d[:, 7].astype(np.int64))
print("total cycles median", np.median(tot), "realtime ticks(100MHz) median", np.median(rt), "=> clock GHz", np.median(tot) / (np.median(rt) * 10) )
print("kernel span (first start to last end) us:", (d[:, 8].max() - d[:, 7].min()) / 100.0)
print("start skew us:", (d[:, 7].max() - d[:, 7].min()) / 100.0)
