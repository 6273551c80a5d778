"""Times rc_density_backward at the BASELINE batch (1024 rays: 65536 samples at levels 0/1, 32768 at level 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import common, nrc_amd

rc = common.make_rc()
rng = np.random.Generator(np.random.PCG64(1))
for level, n in ((0, 65536), (1, 65536), (2, 32768)):
    # ray-ordered samples like a training batch: S consecutive points march along each of 1024 rays through the scene ball
    S = n // 1024
    o = rng.normal(size=(1024, 1, 3)); o = 4.0 * o / np.linalg.norm(o, axis=-1, keepdims=True)
    tgt = rng.normal(size=(1024, 1, 3)) * 0.4
    d = (tgt - o) / np.linalg.norm(tgt - o, axis=-1, keepdims=True)
    t = np.linspace(2.0, 6.0, S)[None, :, None] + rng.uniform(0, 4.0 / S, size=(1024, 1, 1))
    pts_np = (o + d * t).reshape(n, 3).astype(np.float32)
    if os.environ.get("RC_BENCH_SHUFFLE"):
        pts_np = pts_np[rng.permutation(n)]
    pts = torch.from_numpy(pts_np).cuda()
    dd = torch.from_numpy(rng.normal(size=(n,)).astype(np.float32)).cuda()
    df = torch.from_numpy((rng.normal(size=(n, 64)) * 0.1).astype(np.float32)).cuda()
    layout, total = rc.density_grad_layout(level)
    flat = torch.zeros(total, dtype=torch.float32, device="cuda")
    for _ in range(3):
        rc.density_backward(level, pts, dd, df, grads=flat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        rc.density_backward(level, pts, dd, df, grads=flat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"level {level}: n={n} grads={total * 4 / 1e6:.1f} MB  {dt * 1e3:.3f} ms/call  {n / dt / 1e6:.1f} M samples/s", flush=True)
