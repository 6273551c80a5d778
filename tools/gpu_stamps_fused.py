"""Diagnostic: per-phase cycle shares of the fused cache kernel (tools/diag/librc_hip.so, make diag)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
rc_ext.library_path = lambda: os.path.join(R, "tools", "diag", "librc_hip.so")
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(nrc_amd.synthetic_weights(cfg))
rc.set_graph_mode(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
outs = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else None
# RC_STAMP_RAYS: random (the bench's incoherent rays, default) | tile (32x32 pixels of the 800x800 camera) | strip (a
# 1024-pixel row strip of it, what render_image's 1024-ray chunks are)
mode = os.environ.get("RC_STAMP_RAYS", "random")
if mode == "random":
    rays = nrc_amd.synthetic_rays(n)
    f = {k: torch.from_numpy(np.asarray(v)).cuda() for k, v in rays.hot_fields().items()}
else:
    assert n == 1024
    cam = nrc_amd.synthetic_camera_rays(800, 800)
    sel = (slice(384, 416), slice(384, 416)) if mode == "tile" else (slice(400, 402), slice(0, 512))
    f = {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v)[sel].reshape(1024, -1))).cuda() for k, v in cam.hot_fields().items()}
print("rays:", mode)
for _ in range(5): rc.render_rays(f, None, outputs=outs) if outs else rc.render_rays(f, None)
torch.cuda.synchronize()
rc.lib.rc_debug_fused_stamps.restype = ctypes.c_void_p
ptr = rc.lib.rc_debug_fused_stamps()
buf = torch.empty(n * 16, dtype=torch.int64, device="cuda")
import ctypes as C
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(ptr), C.c_size_t(n * 16 * 8), 3)
torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(n, 16)
seg = np.diff(d[:, :12], axis=1)
names = ["begin+resample0", "gather0", "mlp0+w", "resample1", "gather1", "mlp1+w", "resample2", "gather2", "mlp2(+bwd)", "shader", "composite"]
tot = d[:, 11] - d[:, 0]
rt = d[:, 15] - d[:, 14]
ghz = np.median(tot) / (np.median(rt) * 10)
print("clock GHz", ghz)
for i, nm in enumerate(names[:11]):
    print(f"  {nm:12s} median {np.median(seg[:, i]):9.0f} cyc = {np.median(seg[:, i]) / ghz / 1e3:7.2f} us   p95 {np.percentile(seg[:, i], 95) / ghz / 1e3:7.2f} us")
if os.environ.get("RC_FUSED_STAGGER", "0") not in ("", "0"):      # two-wavefront kernel with a late half: phases per half
    t0 = d[:, 0].astype(np.float64); base = np.percentile(t0, 2); late = (t0 - base) > 0.5 * float(os.environ["RC_FUSED_STAGGER"])
    t0 = t0 - base + t0.min()
    for nm2, sel in (("early", ~late), ("late", late)):
        if sel.sum() == 0: continue
        print(f"  [{nm2}: {int(sel.sum())} rays] start {np.median(t0[sel] - t0.min()) / ghz / 1e3:6.1f} us; " +
              " ".join(f"{nm.split('+')[0][:9]} {np.median(seg[sel, i]) / ghz / 1e3:5.1f}" for i, nm in enumerate(names)) +
              f"; total {np.median(tot[sel]) / ghz / 1e3:6.1f} us; end {np.median(d[sel, 11] - t0.min()) / ghz / 1e3:6.1f} us")
print("total median us", np.median(tot) / ghz / 1e3, "kernel span us", (d[:, 15].max() - d[:, 14].min()) / 100.0, "start skew us", (d[:, 14].max() - d[:, 14].min()) / 100.0)
if os.environ.get("RC_STAMP_DETAIL"):
    # who finishes last?  launch-relative start / end (100 MHz realtime counter) by workgroup (2 rays each) and XCD (block % 8)
    rt_start = (d[:, 14] - d[:, 14].min()) / 100.0
    rt_end = (d[:, 15] - d[:, 14].min()) / 100.0
    blk = np.arange(n) // 2
    print("end time us: min %.1f  p5 %.1f  median %.1f  p95 %.1f  max %.1f" % (rt_end.min(), np.percentile(rt_end, 5), np.median(rt_end), np.percentile(rt_end, 95), rt_end.max()))
    for x in range(8):
        sel = (blk % 8) == x
        print(f"  block % 8 == {x}: start {np.median(rt_start[sel]):5.2f}  end median {np.median(rt_end[sel]):6.1f}  max {rt_end[sel].max():6.1f}  total(cyc)/ghz median {np.median(tot[sel]) / ghz / 1e3:6.1f}")
    order = np.argsort(rt_end)
    print("  10 last rays:", [(int(i), round(float(rt_end[i]), 1)) for i in order[-10:]])
    half = blk >= (n // 4)
    hw = d[:, 13].astype(np.int64) & 0xffffffff
    print("  hand-off barriers of wave 0: count", int(np.median(d[:, 13].astype(np.int64) >> 32)), " cycles waited: median %.0f = %.2f us, p95 %.2f us" % (np.median(d[:, 12]), np.median(d[:, 12]) / ghz / 1e3, np.percentile(d[:, 12], 95) / ghz / 1e3))
    for nm2, sel in (("first half", blk < (n // 4)), ("second half", blk >= (n // 4))):
        print(f"    [{nm2}] waited median {np.median(d[sel, 12]) / ghz / 1e3:.2f} us")
    if hw.any():       # HW_REG_HW_ID of wave 0 of each ray: SIMD_ID bits 5:4, CU_ID 11:8, SH_ID 12, SE_ID 15:13
        simd = (hw >> 4) & 3
        cu = (hw >> 8) & 0xff
        print("  SIMD of the stamping wave: first half", np.bincount(simd[~half], minlength=4), "second half", np.bincount(simd[half], minlength=4),
              "; even ray slots", np.bincount(simd[0::2], minlength=4), "odd", np.bincount(simd[1::2], minlength=4))
    for nm2, sel in (("first half", ~half), ("second half", half)):
        print(f"  [{nm2}] " + " ".join(f"{nm.split('+')[0][:9]} {np.median(seg[sel, i]) / ghz / 1e3:5.1f}" for i, nm in enumerate(names)) +
              f"; shader start {np.median((d[sel, 9] - d[sel, 0])) / ghz / 1e3:6.1f} end {np.median((d[sel, 10] - d[sel, 0])) / ghz / 1e3:6.1f} us after its own start")
    print("  first half of the grid: end median %.1f max %.1f; second half: end median %.1f max %.1f" % (np.median(rt_end[~half]), rt_end[~half].max(), np.median(rt_end[half]), rt_end[half].max()))
if len(sys.argv) > 3:      # JSON for profiles/fused_phase_stamps.json (bench.py's hashgrid.in_fused_kernel block)
    import json
    json.dump({"source_hash": rc_ext.source_hash(), "n_rays": n, "rays": mode, "clock_ghz": float(ghz),
               "phases_us": {nm.split("+")[0].split("(")[0] if nm.startswith("gather") else nm: float(np.median(seg[:, i]) / ghz / 1e3) for i, nm in enumerate(names)},
               "total_us": float(np.median(tot) / ghz / 1e3),
               "how": "tools/gpu_stamps_fused.py on the -DRC_STAMPS build (make diag): s_memtime at the phase boundaries of "
                      "k_cache_fused, median over the rays of one 1024-ray batch; the stamps add ~3 % to the kernel"},
              open(sys.argv[3], "w"), indent=1)
# Slots 12 / 13 hold detail stamps INSIDE the first lookup only in the one-wave kernel (rc_fused.hip); the two-wave kernel
# (rc_fused2.hip, the default) keeps its hand-off barrier wait / count + HW_ID there (RC_STAMP_DETAIL above).  Print the
# detail line only when the slots really are time stamps between phase 1 and phase 2 of (nearly) every ray.
is_detail = np.mean((d[:, 1] <= d[:, 12]) & (d[:, 12] <= d[:, 13]) & (d[:, 13] <= d[:, 2])) > 0.95
if is_detail:
    print("gather0 detail us: pos+contract", np.median(d[:, 12] - d[:, 1]) / ghz / 1e3, "issue", np.median(d[:, 13] - d[:, 12]) / ghz / 1e3, "wait+combine", np.median(d[:, 2] - d[:, 13]) / ghz / 1e3)
else:
    print("gather0 detail: not stamped by this kernel (two-wave kernel: slots 12 / 13 = hand-off wait, count | HW_ID)")
