"""What the loose tolerances of tests/test_gpu_parity.py actually hold: L-inf of every non-RGB key of the 256-ray cache pass,
HIP vs the fp32 oracle, HIP vs the fp64 oracle, and the fp32 oracle vs the fp64 oracle (the noise floor of fp32 arithmetic
in the reference's own order).  Run on the GPU box: python tools/measure_tolerances.py > gpurun_out/tolerances.txt"""
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

KEYS = ("rgb", "acc", "means", "normals_pred", "ray_dists", "light_dists", "distance_mean", "distance_median",
        "distance_percentile_5", "distance_percentile_95")


def main():
    h = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
    h.load_weights(common.weights_np())
    print("key                      | HIP-fp32 oracle | HIP-fp64 oracle | fp32-fp64 oracle | magnitude")
    for n, js in ((256, None), (256, 7), (1024, 3)):
        rays = nrc_amd.synthetic_rays(n, seed=20200823)
        rnd = None if js is None else {"jitter": common.jitters(n, seed=js)}
        h.set_fused(False)
        out = {k: v.cpu().numpy() for k, v in h.render_rays(rays.hot_fields(), rnd).items()}
        torch.cuda.synchronize()
        r32 = common.oracle_cache(n, jitter_seed=js, want_grad_normals=False)
        r64 = common.oracle_cache(n, jitter_seed=js, want_grad_normals=False, dtype=torch.float64)
        print(f"--- {n} rays, jitter seed {js}")

        def row(name, a, b32, b64):
            b32 = np.asarray(b32, np.float64).reshape(a.shape); b64 = np.asarray(b64, np.float64).reshape(a.shape)
            print(f"{name:24s} | {np.abs(a - b32).max():15.3e} | {np.abs(a - b64).max():15.3e} | {np.abs(b32 - b64).max():16.3e} | {np.abs(b64).max():9.3g}")
        for k in KEYS:
            a = out[k].astype(np.float64)
            b32, b64 = r32["render"][k].numpy(), r64["render"][k].numpy()
            if k in ("ray_dists", "light_dists"): b32, b64 = b32[:, 0], b64[:, 0]
            row(k, a, b32, b64)
        for l, S in enumerate((64, 64, 32)):
            for w, cols in (("sdist", S + 1), ("tdist", S + 1), ("weights", S), ("density", S)):
                a = h.workspace(f"{w}{l}").reshape(n, cols).astype(np.float64)
                row(f"{w}{l}", a, r32["sampler"][l][w].numpy(), r64["sampler"][l][w].numpy())
        h.set_fused(True)


if __name__ == "__main__":
    main()
