"""Diagnostic: cycle shares inside k_transient_bins (tools/diag/librc_hip.so, `make diag` in csrc)."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
rc_ext.library_path = lambda: os.path.join(R, "tools", "diag", *(sys.argv[1:2]), "librc_hip.so")
cfg = nrc_amd.cornell_transient_config()
rc = rc_ext.RadianceCache(cfg, 0)
rc.load_weights(nrc_amd.synthetic_weights(cfg))
n = 1024
rays = nrc_amd.synthetic_transient_rays(n)
f = {k: torch.from_numpy(np.asarray(v)).cuda().contiguous() for k, v in rays.hot_fields().items()}
keys = ["rgb", "integrated_rgb", "acc", "transient_direct_viz", "transient_indirect_viz"]
for _ in range(4):
    rc.render_transient(f, None, outputs=keys)
torch.cuda.synchronize()
rc.lib.rc_debug_bins_stamps.restype = C.c_void_p
ptr = rc.lib.rc_debug_bins_stamps()
buf = torch.empty(n * 8, dtype=torch.int64, device="cuda")
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(ptr), C.c_size_t(n * 8 * 8), 3)
d = buf.cpu().numpy().reshape(n, 8)
med = lambda x: float(np.median(x))
total = med(d[:, 4] - d[:, 5])
print(f"per-wave cycles (median over {n} rays): total {total:.0f}")
print(f"  prologue        {med(d[:, 0] - d[:, 5]):10.0f}")
print(f"  tiles: MFMA     {med(d[:, 1]):10.0f}   ({med(d[:, 1]) / 66:.0f} per tile; 98 MFMAs = 6272 cycles of matrix pipe)")
print(f"  tiles: epilogue {med(d[:, 2]):10.0f}   ({med(d[:, 2]) / 66:.0f} per tile)")
print(f"  tiles: other    {med(d[:, 3] - d[:, 0] - d[:, 1] - d[:, 2]):10.0f}")
print(f"  tail            {med(d[:, 4] - d[:, 3]):10.0f}   (bin sums {med(d[:, 6] - d[:, 3]):.0f}, direct scatter {med(d[:, 7] - d[:, 6]):.0f}, filter + outputs + extras {med(d[:, 4] - d[:, 7]):.0f})")
