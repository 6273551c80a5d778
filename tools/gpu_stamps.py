"""Diagnostic: per-segment cycle shares of the stand-alone cache-shader kernel (tools/diag/librc_hip.so, make diag)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
rc_ext.library_path = lambda: os.path.join(R, "tools", "diag", "librc_hip.so")
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(nrc_amd.synthetic_weights(cfg))
rc.set_graph_mode(0); rc.set_fused(False)
rays = nrc_amd.synthetic_rays(1024)
for _ in range(5): rc.render_rays(rays.hot_fields(), None, outputs=["rgb"])
torch.cuda.synchronize()
d = rc.workspace("debug").view(np.uint64)[: 1024 * 16].reshape(1024, 16)
st = d[:, :12].astype(np.int64); seg = np.diff(st, axis=1)
names = ["ws_begin+stage", "(enter)", "heads(49)", "IDE", "s0(680)", "ibrdf(98+66+dot)", "s1(260)+park", "s2(260)+park", "sb(256)", "so dot", "combine"]
mf = [0, 0, 49, 0, 680, 164, 260, 260, 256, 0, 0]
print("median cycles per segment:")
for i, nme in enumerate(names):
    med = np.median(seg[:, i]); print(f"  {nme:18s} {med:9.0f}  ideal {mf[i]*64:7d}  eff {mf[i]*64/med if med else 0:.2f}")
tot = st[:, 11] - st[:, 0]; rt = (d[:, 13].astype(np.int64) - d[:, 12].astype(np.int64))
print("total cycles median", np.median(tot), "realtime ticks(100MHz) median", np.median(rt), "=> clock GHz", np.median(tot) / (np.median(rt) * 10))
print("kernel span (first start to last end) us:", (d[:, 13].max() - d[:, 12].min()) / 100.0)
