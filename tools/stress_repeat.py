"""Race screen of the cache pass: the same batch rendered many times on every launch plan must come out bit for bit the
same, whatever the co-residence of workgroups (1024 rays: two workgroups per CU; 4097 / 25001: several rounds per CU and
the level kernels with three waves per SIMD).  Prints one line per (plan, size): launches that differed from the first.
The screen that found the instability of the two-wave kernel under the split-MFMA form (csrc/rc_dev_mlp.h): python tools/stress_repeat.py [launches]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import nrc_amd  # noqa: E402
from nrc_amd import rc_ext  # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    h = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
    h.load_weights(common.weights_np())
    h.set_graph_mode(0)
    total_bad = 0
    sizes = [int(x) for x in os.environ.get("STRESS_SIZES", "1024,4097,25001").split(",")]
    for n in sizes:
        rays = nrc_amd.synthetic_rays(n, seed=77)
        f = {k: torch.from_numpy(np.asarray(v)).cuda().contiguous() for k, v in rays.hot_fields().items()}
        rnd = {"jitter": [torch.from_numpy(j).cuda() for j in common.jitters(n, seed=5)]}
        firsts = {}
        for plan, name in ((1, "fused (two waves per ray)"), (3, "fused (one wave per ray)"), (0, "launch per stage")):
            h.set_fused(plan)
            first = {k: v.clone() for k, v in h.render_rays(f, rnd).items()}
            torch.cuda.synchronize()
            firsts[plan] = first
            reps = max(4, launches * 1024 // n)
            bad, worst, rays_bad = 0, 0.0, 0
            for it in range(reps):
                out = h.render_rays(f, rnd)
                torch.cuda.synchronize()
                d = max(float((out[k] - first[k]).abs().max()) for k in out)
                if d > 0:
                    if bad < 2:
                        keys = {k: float((out[k] - first[k]).abs().max()) for k in out}
                        print("      differing keys:", {k: f"{v:.1e}" for k, v in keys.items() if v > 0}, flush=True)
                    bad += 1
                    worst = max(worst, d)
                    rays_bad += int(((out["rgb"] - first["rgb"]).abs().reshape(n, -1).max(dim=1).values > 0).sum())
            total_bad += bad
            print(f"{n:6d} rays, {name:26s}: {bad} of {reps} launches differ from the first (rays {rays_bad}, max {worst:.3e})", flush=True)
        for plan in (3, 0):
            d = max(float((firsts[plan][k] - firsts[1][k]).abs().max()) for k in firsts[1])
            print(f"{n:6d} rays, plan {plan} vs the two-wave kernel: max difference {d:.3e}", flush=True)
    h.set_fused(True)
    print("stable" if total_bad == 0 else f"UNSTABLE: {total_bad} launches differed")
    return 0 if total_bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
