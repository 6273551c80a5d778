"""Where the time of the fused cache kernel goes: per phase (and per wave role, for the two-wave kernel of fp32-MFMA
builds), measured with in-kernel stamps.  A split-MFMA build (rc_ext.mlp_arithmetic() == "bf16x3-split") runs the
one-wave-per-ray kernel: one column of waves, one wave per SIMD, MFMA issue = 64 cycles per fp32 MFMA of the density MLPs
(two 32-sample tiles per wave on the proposal levels: 148 / 152, the last level with its backward pass 196) and 32 cycles
per bf16 MFMA of the shader (223 cells x 6 products + 2 flush MFMAs per layer = 1352).

Needs two diagnostic builds of the library (tools/prof_round.sh builds them):
    make -C neural-radiance-caching_amd/csrc diag                                          -> tools/diag/librc_hip.so
    make -C neural-radiance-caching_amd/csrc diag DIAG_EXTRA=-DRC_GATHER_FAKE=0 DIAG_DIR=fake  -> tools/diag/fake/librc_hip.so
The second one sends every table lookup to entry 0 (all lanes hit one line): its phases are what the SAME instruction
stream takes without memory time -- issue time of the lookups (address arithmetic, interpolation) and everything else
unchanged.  Per phase the table prints, for wave 0 and wave 1 of a ray (median over the 1024 rays of one batch):
    measured          s_memtime difference between the phase's boundary stamps (the stamps add ~3 % to the kernel)
    no-memory         the same phase of the RC_GATHER_FAKE build
    memory wait       measured - no-memory  (what the lookups wait for beyond their own issue time)
    MFMA issue        MFMAs of the phase x 64 cycles x 2 (both waves of a SIMD -- two workgroups -- run the same phase at
                      the same time, PMC: SQ_VALU_MFMA_BUSY_CYCLES) / clock: the matrix pipe's own time
    other             no-memory - MFMA issue: vector / LDS / scalar issue, LDS and MFMA-result latency, barriers
The MFMA counts per phase come from the ISA of the stamped build (v_mfma between the stamps, per wave): 74 / 76 for the
proposal levels (one 32-point tile per wave), 114 for the last density MLP with its backward pass (GRAD), 859 for the
shader (each wave half of the output tiles).  Sum 1123 per wave = 2246 per ray."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext

N = 1024
NAMES = ["begin+resample0", "gather0", "mlp0+w", "resample1", "gather1", "mlp1+w", "resample2", "gather2", "mlp2+bwd", "shader", "composite"]
MFMA = [0, 0, 74, 0, 0, 76, 0, 0, 114, 859, 0]


def run(lib):
    """stamps [2 roles, N rays, 16] of one launch with the library `lib`, in a child process (one library per process)."""
    import subprocess, json, tempfile
    out = tempfile.mktemp(suffix=".npy")
    code = f"""
import ctypes as C, os, sys
sys.path.insert(0, {R!r})
import numpy as np, torch
import nrc_amd
from nrc_amd import rc_ext
rc_ext.library_path = lambda: {lib!r}
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0); rc.load_weights(nrc_amd.synthetic_weights(cfg)); rc.set_graph_mode(0)
rays = nrc_amd.synthetic_rays({N})
f = {{k: torch.from_numpy(np.asarray(v)).cuda() for k, v in rays.hot_fields().items()}}
for _ in range(5): rc.render_rays(f, None)
torch.cuda.synchronize()
rc.lib.rc_debug_fused_stamps.restype = C.c_void_p
ptr = rc.lib.rc_debug_fused_stamps()
buf = torch.empty(2 * {N} * 16, dtype=torch.int64, device="cuda")
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(ptr), C.c_size_t(2 * {N} * 16 * 8), 3)
torch.cuda.synchronize()
np.save({out!r}, buf.cpu().numpy().reshape(2, {N}, 16))
"""
    subprocess.run([sys.executable, "-c", code], check=True, stderr=subprocess.DEVNULL)
    d = np.load(out); os.remove(out)
    return d


SPLIT = rc_ext.mlp_arithmetic() == "bf16x3-split"
MFMA_CYC = [0, 0, 148 * 64, 0, 0, 152 * 64, 0, 0, 196 * 64, 1352 * 32, 0] if SPLIT else [m * 64 * 2 for m in MFMA]
real = run(os.path.join(R, "tools", "diag", "librc_hip.so"))
fake = run(os.path.join(R, "tools", "diag", "fake", "librc_hip.so"))
tot = real[0][:, 11] - real[0][:, 0]
rt = real[0][:, 15] - real[0][:, 14]
ghz = float(np.median(tot) / (np.median(rt) * 10))            # s_memtime ticks per ns (realtime counter: 100 MHz)
us = lambda cyc: cyc / ghz / 1e3
print(f"clock {ghz:.3f} GHz; launch span (first start -> last end) {(real[0][:, 15].max() - real[0][:, 14].min()) / 100.0:.1f} us real, "
      f"{(fake[0][:, 15].max() - fake[0][:, 14].min()) / 100.0:.1f} us without memory time; source {rc_ext.source_hash()}")
print(f"{'phase':16s} | {'wave 0: measured':>16s} {'no-memory':>10s} {'mem wait':>9s} | {'wave 1: measured':>16s} {'no-memory':>10s} {'mem wait':>9s} | {'MFMA issue':>10s} {'other':>7s}")
sums = np.zeros(8)
two = bool(real[1][:, 11].any())            # the one-wave kernel leaves the second half of the stamp buffer untouched
if not two: print("(one wave per ray: the wave-1 columns repeat wave 0)")
for i, nm in enumerate(NAMES):
    row = []
    for q in ((0, 1) if two else (0, 0)):
        m = float(np.median(real[q][:, i + 1] - real[q][:, i])); f = float(np.median(fake[q][:, i + 1] - fake[q][:, i]))
        row += [us(m), us(f), us(m) - us(f)]
    mf = us(MFMA_CYC[i])
    other = row[1] - mf
    print(f"{nm:16s} | {row[0]:16.2f} {row[1]:10.2f} {row[2]:9.2f} | {row[3]:16.2f} {row[4]:10.2f} {row[5]:9.2f} | {mf:10.2f} {other:7.2f}")
    sums += np.array(row + [mf, other])
print(f"{'sum':16s} | {sums[0]:16.2f} {sums[1]:10.2f} {sums[2]:9.2f} | {sums[3]:16.2f} {sums[4]:10.2f} {sums[5]:9.2f} | {sums[6]:10.2f} {sums[7]:7.2f}")
print(f"per-ray total (stamp 0 -> 11), median: wave 0 {us(float(np.median(tot))):.2f} us; the phases above are {sums[0] / us(float(np.median(tot))) * 100:.1f} % of it")
hb = real[0][:, 12].astype(np.float64)
if two: print(f"hand-off barriers of wave 0: {int(np.median(real[0][:, 13].astype(np.int64) >> 32))} per ray, waited {us(float(np.median(hb))):.2f} us (median; arrival skew + the fence's drain)")
