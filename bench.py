#!/usr/bin/env python3
"""Headline benchmark: rays/sec of the radiance-cache forward on 1024-ray batches.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (rc_render_rays: 3 proposal rounds -> hash-grid lookups ->
density MLPs -> cache shader -> volume compositing) over one batch of 1024 synthetic rays with
the hotdog architecture and synthetic weights; rays, weights and outputs are resident in HBM.
With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank renders its own
1024-ray batch per step (ray batches shard trivially, no data-path collective; weak scaling) and
the rendered pixels are all-gathered over RCCL once at the end, outside the per-step loop, exactly
as the image renderer does once per image.

The timed loop cycles through 64 different resident ray batches (no step re-renders the rays of the step before).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fused cache kernel, MFMA bound: priced in
algorithmic fp32 flops against the fp32 MFMA peak, `roofline.issued_bf16` the split build's products on the bf16 pipe); its duration comes from HIP events recorded on the launch stream inside the timed region; `traffic` is read
from the PMC file tools/prof_round.sh leaves under profiles/ and is null when that file belongs to other kernel sources.
`hashgrid` reports the achieved algorithmic GB/s of the grid lookups twice: of the four stand-alone kernels (measured
here, separate pass on the launch-per-stage plan) and of the gather phases inside the fused kernel (in-kernel stamps
of a diagnostic build, profiles/, same staleness rule).  `parity` is max |rgb - oracle| and the PSNR of the first batch.
`image` times BASELINE configs[3]: the 800 x 800 image, rays sharded over the ranks (strong scaling), one all-gather.
`cpu_baseline` times the CPU oracle (a torch fp32 restatement of the reference path, kind "port":
the JAX reference cannot run here) on the same 1024-ray batch by BASELINE.md's protocol: 8 threads and all granted host
cores, 3 warm-up + 10 timed passes each, median (`cpu_baseline.runs` holds both, `value` the all-cores run).
`image.ms_per_image_to_host` adds the one device-to-host copy of the gathered keys to the device-complete image time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAYS_PER_BATCH = 1024
# algorithmic work per primary ray, cache-only
GRID_BYTES = {"grid0": 64 * 192, "grid1": 64 * 224, "grid2": 32 * 1024, "grid_app": 32 * 1024}   # per ray
SHADER_FLOP_PER_SAMPLE = 253824          # output-relevant (heads + IBRDF + SurfaceLightField MLP)
SHADED_SAMPLES = 32
# SURVEY.md §8(d) figure of record per primary ray, cache-only: 64 x 9 088 + 64 x 9 216 + 32 x 12 800 (density MLPs of
# the three proposal levels) + 32 x 253 824 (shader) = 9 703 424 FLOP.  The backward pass for the analytic normals
# that the fused kernel also runs (32 x 12 416 FLOP) is NOT counted.
DENSITY_FLOP_PER_SAMPLE = (9088, 9216, 12800)
SAMPLES = (64, 64, 32)
FUSED_FLOP_PER_RAY = sum(s * f for s, f in zip(SAMPLES, DENSITY_FLOP_PER_SAMPLE)) + SHADED_SAMPLES * SHADER_FLOP_PER_SAMPLE
assert FUSED_FLOP_PER_RAY == 9703424
PEAK_F32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0
N_BATCHES = 64                           # distinct ray batches the timed loop cycles through
PROFILES = os.path.join(ROOT, "profiles")


def measured_file(name, source_hash):
    """A measurement kept under profiles/ (written by tools/prof_round.sh / tools/gpu_stamps_fused.py on a GPU box):
    returned only when it was taken on the kernel sources this run uses, else None (stale numbers are not reported)."""
    path = os.path.join(PROFILES, name)
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, f"profiles/{name} missing"
    if d.get("source_hash") != source_hash:
        return None, f"profiles/{name} was measured on other kernel sources ({d.get('source_hash')} != {source_hash})"
    return d, f"profiles/{name}"


def host_cpu_share(cap=16):
    """CPUs this process may really use: scheduler affinity, the cgroup CPU quota, capped at the one-GPU share of
    the box (oversubscribing torch's intra-op pool on a quota-limited container stalls for minutes)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, cap))


def cpu_baseline(cfg, weights_np, n_rays, warmup=3, timed=10, budget_s=90.0):
    """BASELINE.md §2 / SURVEY §8(d) protocol: the torch-fp32 oracle on the same 1024-ray batch at N = 8 threads AND at
    N = all host cores this process is granted, 3 warm-up + 10 timed passes each, MEDIAN reported; `value` / `cores` are
    the all-cores run, `runs` holds both.  Bounded: a thread count stops early (and says so) once `budget_s` is spent in
    total; progress goes to stderr."""
    import numpy as np
    import torch

    import nrc_amd
    from oracle import cache_ref

    granted = host_cpu_share()
    wt = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    rays = nrc_amd.synthetic_rays(n_rays)
    rt = {k: torch.from_numpy(np.asarray(v)) for k, v in rays.hot_fields().items()}
    run = lambda: cache_ref.cache_forward(wt, cfg, rt, None, want_grad_normals=True, exec_dead_envmap=True)
    runs = []
    t_start = time.time()
    for cores in sorted({min(8, granted), granted}):
        torch.set_num_threads(cores)
        times = []
        for i in range(warmup + timed):
            if i > warmup and time.time() - t_start > budget_s:
                break
            t0 = time.time()
            run()
            dt = time.time() - t0
            if i >= warmup:
                times.append(dt)
            print(f"[bench] cpu_baseline {cores} threads, pass {i + 1}/{warmup + timed}"
                  f"{' (warm-up)' if i < warmup else ''}: {dt:.2f} s", file=sys.stderr, flush=True)
        med = sorted(times)[len(times) // 2]
        runs.append({"cores": torch.get_num_threads(), "value": n_rays / med, "unit": "rays/s", "median_ms": med * 1e3,
                     "warmup": warmup, "timed": len(times)})
    best = runs[-1]                                       # all granted cores
    return {"value": best["value"], "unit": "rays/s", "cores": best["cores"], "kind": "port", "runs": runs,
            "sample": f"{warmup} warm-up + {best['timed']} timed passes of one {n_rays}-ray batch (64,64,32 samples) per thread "
                      f"count (8 and all {granted} granted host cores), torch fp32 oracle incl. the reference's dead "
                      f"cache-EnvMap MLP and autograd normals; median {best['median_ms']:.0f} ms at {best['cores']} threads"}


def transient_line(local_rank, dev, n_rays=1024, steps=20, warmup=3):
    """Secondary measurement (not part of `value`): rc_render_transient on 1024 synthetic cornell rays,
    700 bins x 3 channels per ray out of two MFMA kernels (BASELINE configs[4])."""
    import numpy as np
    import torch

    import nrc_amd
    from nrc_amd import rc_ext

    cfg = nrc_amd.cornell_transient_config()
    rc = rc_ext.RadianceCache(cfg, local_rank)
    rc.load_weights(nrc_amd.synthetic_weights(cfg))
    rays = nrc_amd.synthetic_transient_rays(n_rays)
    f = {k: torch.from_numpy(np.asarray(v)).to(dev).contiguous() for k, v in rays.hot_fields().items()}
    keys = ["rgb", "integrated_rgb", "acc", "transient_direct_viz", "transient_indirect_viz"]
    for _ in range(warmup):
        rc.render_transient(f, None, outputs=keys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rc.render_transient(f, None, outputs=keys)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # algorithmic FLOP of the two wide heads alone: 32 samples x 2 x (128 x 2101 + 64 x 2100) per ray
    head_flop = n_rays * 32 * 2 * (128 * 2101 + 64 * 2100)
    return {"workload": "cornell time-resolved cache, 1024 rays x (64,64,32) samples x 700 bins (configs[4])",
            "rays_per_s": n_rays / (ms * 1e-3), "ms_per_step": ms, "steps": steps,
            "wide_heads_gflop_per_step": head_flop / 1e9,
            "note": "output allocation (3 x [1024,700,3] fp32 zero-fills) is inside the step"}


def material_line(local_rank, dev, n_rays=1024, steps=20, warmup=3):
    """Secondary measurement (not part of `value`): rc_render_material on 1024 synthetic hotdog rays =
    material / light heads, 32 importance-sampled secondary rays per primary ray traced through the cache
    (32 768 secondary rays per step), Monte-Carlo BRDF integration (BASELINE configs[2])."""
    import numpy as np
    import torch

    import nrc_amd
    from nrc_amd import rc_ext

    cfg = nrc_amd.hotdog_config()
    rc = rc_ext.RadianceCache(cfg, local_rank)
    rc.load_weights(nrc_amd.synthetic_weights(cfg, passes=("cache", "material")))
    rays = nrc_amd.synthetic_rays(n_rays)
    f = {k: torch.from_numpy(np.asarray(v)).to(dev).contiguous() for k, v in rays.hot_fields().items()}
    rng = np.random.Generator(np.random.PCG64(0))
    K, S = cfg.num_secondary_samples, cfg.sampling_strategy[-1][2]
    Kd = int(round(K * cfg.diffuse_sample_fraction))
    Ks, kc = K - Kd, int(round(0.5 * int(round(K * cfg.diffuse_sample_fraction))))
    u = lambda *shape: torch.from_numpy(rng.uniform(size=shape).astype(np.float32)).to(dev)
    g = lambda *shape: torch.from_numpy(rng.gumbel(size=shape).astype(np.float32)).to(dev)
    rnd = dict(jitter=[u(n_rays) for _ in range(3)], gumbel=g(n_rays, S),
               vmf_noise=torch.from_numpy(rng.normal(size=(n_rays, cfg.num_vmf, 3)).astype(np.float32)).to(dev),
               spec_u1=u(n_rays, Ks), spec_u2=u(n_rays, Ks), cos_u1=u(n_rays, kc), cos_u2=u(n_rays, kc),
               vmf_lobe=torch.from_numpy(rng.integers(0, cfg.num_vmf, size=(n_rays,)).astype(np.int32)).to(dev),
               vmf_v=torch.from_numpy(rng.normal(size=(n_rays, Kd - kc, 2)).astype(np.float32)).to(dev), vmf_tmp=u(n_rays, Kd - kc),
               spec_jitter=[u(n_rays * Ks) for _ in range(3)], spec_gumbel=g(n_rays * Ks, S),
               diff_jitter=[u(n_rays * Kd) for _ in range(3)], diff_gumbel=g(n_rays * Kd, S))
    # the secondary trace's randoms in the ABI's own layout ([specular block | diffuse block], rc_abi.h rc_material_randoms)
    rnd["sec_jitter"] = [torch.cat([rnd["spec_jitter"][l], rnd["diff_jitter"][l]]) for l in range(3)]
    rnd["sec_gumbel"] = torch.cat([rnd["spec_gumbel"], rnd["diff_gumbel"]], dim=0).contiguous()
    for _ in range(warmup):
        rc.render_material(f, rnd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rc.render_material(f, rnd)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"workload": "hotdog material stage, 1024 primary rays x 32 secondary rays x (64,64,32) samples (configs[2])",
            "primary_rays_per_s": n_rays / (ms * 1e-3), "secondary_rays_per_s": n_rays * K / (ms * 1e-3),
            "ms_per_step": ms, "steps": steps}


def train_backward_line(local_rank, dev, n_rays=1024, steps=20, warmup=3):
    """Secondary measurement (not part of `value`): rc_density_backward for the three proposal levels of one 1024-ray
    batch (64 + 64 + 32 ray-ordered samples per ray): gradients of the hash-grid tables and density MLPs
    (SURVEY 8(f) rank 4)."""
    import numpy as np
    import torch

    import nrc_amd
    from nrc_amd import rc_ext

    cfg = nrc_amd.hotdog_config()
    rc = rc_ext.RadianceCache(cfg, local_rank)
    rc.load_weights(nrc_amd.synthetic_weights(cfg))
    rng = np.random.Generator(np.random.PCG64(1))
    rays = nrc_amd.synthetic_rays(n_rays)
    o = np.asarray(rays.origins)[:, None, :]
    d = np.asarray(rays.directions)[:, None, :]
    per_level = {}
    total_ms = 0.0
    for level, S in enumerate(lvl[2] for lvl in cfg.sampling_strategy):
        t = np.linspace(2.0, 6.0, S)[None, :, None] + rng.uniform(0, 4.0 / S, size=(n_rays, 1, 1))
        n = n_rays * S
        pts = torch.from_numpy((o + d * t).reshape(n, 3).astype(np.float32)).to(dev)
        dd = torch.from_numpy(rng.normal(size=(n,)).astype(np.float32)).to(dev)
        df = torch.from_numpy((rng.normal(size=(n, 64)) * 0.1).astype(np.float32)).to(dev)
        _, total = rc.density_grad_layout(level)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        for _ in range(warmup):
            rc.density_backward(level, pts, dd, df, grads=flat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            rc.density_backward(level, pts, dd, df, grads=flat)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        total_ms += ms
        per_level[f"level{level}"] = {"samples": n, "ms": ms, "grad_mbytes": total * 4 / 1e6}
    return {"workload": "backward of the three proposal density fields for one 1024-ray batch (tables + density MLPs)",
            "rays_per_s": n_rays / (total_ms * 1e-3), "ms_per_step": total_ms, "steps": steps, "levels": per_level,
            "note": "scatter into the tables is bound by the memory-side atomic request rate (DESIGN.md 4.4)"}


def image_line(rc_model, cfg, dev, world, rank, dist, reps=3):
    """BASELINE configs[3]: the 800 x 800 image (640 000 rays) rendered by render_image_distributed -- every rank its
    contiguous share of the rays (cast on the device beforehand: inputs resident in HBM), one all-gather of the consumed
    keys per image (RCCL when world > 1).  Strong scaling: the image is fixed, the ranks split it.  Max over ranks."""
    import numpy as np
    import torch

    import nrc_amd
    from nrc_amd import model as M

    H = W = 800
    o = np.array([0.0, -3.5, 2.0])
    look = -o / np.linalg.norm(o)
    right = np.cross(look, [0, 0, 1.0]); right /= np.linalg.norm(right)
    up = np.cross(right, look)
    c2w = np.concatenate([np.stack([right, up, -look], 1), o[:, None]], 1)
    cam = nrc_amd.Camera(nrc_amd.get_pixtocam(1111.0, W, H), c2w, near=2.0, far=6.0)
    rays = rc_model.rc.cast_rays(cam, rect=(0, 0, W, H))
    icfg = nrc_amd.hotdog_config(render_chunk_size=16384)
    apply = lambda rng, r: rc_model.apply(None, rng, r)
    keys = ("rgb", "acc", "distance_median", "normals_pred")
    times = []
    for i in range(reps + 1):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = M.render_image_distributed(apply, None, rays, icfg, keys=keys)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if i:
            times.append(dt)
    ms = sorted(times)[len(times) // 2] * 1e3
    # the same image handed over the way render_image hands it over (numpy on the host, internal/models.py:2448-2450 copies
    # every chunk; here ONE device-to-host copy of the gathered keys per image): end-to-end wall clock incl. that copy
    host_times = []
    for i in range(reps + 1):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        himg = M.render_image_distributed(apply, None, rays, icfg, keys=keys, to_host=True)
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if i:
            host_times.append(dt)
    assert isinstance(himg["rgb"], np.ndarray) and himg["rgb"].shape == (H, W, 3)
    ms_host = sorted(host_times)[len(host_times) // 2] * 1e3
    return {"workload": "lego/hotdog-architecture 800x800 image, rays sharded over the ranks, one all-gather of "
                        "rgb+acc+distance_median+normals_pred per image (configs[3])",
            "scaling": "strong", "n_gpus": world, "ms_per_image": ms, "rays_per_s": H * W / (ms * 1e-3),
            "ms_per_image_to_host": ms_host, "rays_per_s_to_host": H * W / (ms_host * 1e-3),
            "to_host_note": "ms_per_image ends device-complete (results in HBM); ms_per_image_to_host adds the single "
                            "device-to-host copy of the gathered keys (20.5 MB) into numpy arrays",
            "render_chunk_size": icfg.render_chunk_size, "acc_mean": float(img["acc"].mean()),
            "collective": "all_gather_into_tensor over RCCL" if world > 1 else "none (one rank)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-transient", action="store_true",
                    help="skip the secondary measurement of the time-resolved cache (configs[4])")
    ap.add_argument("--no-material", action="store_true",
                    help="skip the secondary measurement of the material stage (configs[2])")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the secondary measurement of the density fields' backward pass (SURVEY 8(f) rank 4)")
    ap.add_argument("--no-image", action="store_true",
                    help="skip the 800x800 image line (configs[3], strong scaling over the ranks)")
    ap.add_argument("--graph-mode", type=int, default=2, help="0 eager, 1 lazy hipGraph, 2 hipGraph at once")
    ap.add_argument("--plan", choices=("fused", "fused1", "staged"), default="fused",
                    help="fused: one launch per batch, two wavefronts per ray (rc_set_fused 1, default); fused1: the same with "
                         "one wavefront per ray (rc_set_fused 3, the round-1/2 kernel); staged: one launch per stage")
    ap.add_argument("--runs", type=int, default=9,
                    help="the timed region (--steps steps between two barriers) is repeated this many times in-process; "
                         "the MEDIAN run is reported as ms_per_step / value, all runs in config.ms_per_step_runs")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent batches in flight: step i is enqueued on HIP stream i %% streams")
    ap.add_argument("--profile-mode", type=int, default=3,
                    help="HIP events in the timed region: 0 none, 1 every stage, 2 around the dominant kernel of every step, "
                         "3 around the dominant kernel of every 8th step (default: the events of mode 2 cost 2 %% of a step)")
    args = ap.parse_args()
    if args.profile_mode == 3 and args.steps < 16:
        args.profile_mode = 2            # too few steps to sample: events on every step
    # stdout carries exactly ONE line, the JSON: anything the GPU runtime or a library writes to file descriptor 1
    # meanwhile (libdrm prints a missing-file notice there at device initialisation) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch

    import nrc_amd
    from nrc_amd import model as M
    from nrc_amd import rc_ext
    from nrc_amd.model import _CACHE_DEVICE_KEYS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"

    cfg = nrc_amd.hotdog_config()
    weights = nrc_amd.synthetic_weights(cfg)
    model = M.Model(cfg, local_rank)
    model.load_variables(weights)
    rc = model.rc
    rc.set_graph_mode(args.graph_mode)
    fused = args.plan in ("fused", "fused1")
    fused_mode = {"fused": 1, "fused1": 3, "staged": 0}[args.plan]
    rc.set_fused(fused_mode)
    # how the library multiplies in the shader MLPs (include/rc_abi.h rc_mlp_arithmetic).  A bf16x3-split build runs the
    # one-wave-per-ray kernel for the default plan too (the two-wave kernel was unstable under the split form)
    arith = rc_ext.mlp_arithmetic()
    dominant = {"fused": "k_cache_fused" if arith == "bf16x3-split" else "k_cache_fused_team", "fused1": "k_cache_fused",
                "staged": "k_cache_shader"}[args.plan]

    def batch(seed):
        r = nrc_amd.synthetic_rays(RAYS_PER_BATCH, seed=seed)
        f = {k: torch.from_numpy(np.asarray(v)).to(dev).contiguous() for k, v in r.hot_fields().items()}
        f["near"], f["far"] = f["near"].reshape(-1), f["far"].reshape(-1)
        f.pop("lossmult", None)
        return f

    # N_BATCHES different ray batches per rank, all resident in HBM: step i renders batch i % N_BATCHES, so no step
    # finds the lines its gathers touch warm from an identical step before it
    batches = [batch(20200823 + 1000 * rank + b) for b in range(N_BATCHES)]
    rays = batches[0]

    nstr = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstr)] if nstr > 1 else [torch.cuda.current_stream(dev)]
    outs = []
    for s_ in streams:
        with torch.cuda.stream(s_):
            outs.append(rc.render_rays(rays, None, outputs=_CACHE_DEVICE_KEYS))
    torch.cuda.synchronize()
    out = outs[0]
    first = {k: v.clone() for k, v in out.items()}          # batch 0, for the parity block

    def step(i):
        with torch.cuda.stream(streams[i % nstr]):
            rc.render_rays(batches[i % N_BATCHES], None, out=outs[i % nstr])

    # Timed region: two HIP events around the dominant kernel, on every 8th step by default (an event record costs
    # ~1.3 us of stream time; the last 16 sampled launches, spread over the last 128 steps, are averaged).  The full
    # per-stage profile is taken in a separate pass below.
    rc.set_profiling(args.profile_mode)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    rc.set_profiling(args.profile_mode)          # reset the event ring

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region -- EXACTLY --steps steps between two (barrier + synchronize), MAX over the ranks -- is run
    # --runs times back to back (9); the median run is the one reported and all of them are listed in
    # config.ms_per_step_runs (a 20-step region is 2.3 ms, a 200-step one 22 ms: the first regions after start-up still
    # warm the MALL with the tables the batches touch -- 0.120, 0.118, 0.117, 0.115, 0.115 ... 0.109 ms per step).
    run_s = []
    for _ in range(max(1, args.runs)):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        run_s.append(el)
    elapsed = sorted(run_s)[len(run_s) // 2]
    if dist is not None:
        # rendered pixels of all ranks, gathered once (image granularity), outside the step loop
        pix = torch.cat([out["rgb"], out["acc"][:, None]], dim=1)
        gathered = torch.empty((world * pix.shape[0], 4), device=dev)
        dist.all_gather(list(gathered.chunk(world)), pix)
    sh_ms = rc.stage_times_ms()["shader"] if args.profile_mode else float("nan")
    rc.set_profiling(0)
    # configs[3]: every rank takes part (collective inside); before the single-rank extras
    image = None
    if not args.no_image:
        rc.set_fused(True)
        try:
            image = image_line(model, cfg, dev, world, rank, dist)
        except Exception as e:      # noqa: BLE001
            print(f"[bench] image line failed: {e!r}", file=sys.stderr, flush=True)
            image = {"error": repr(e)}
    # separate pass: every stage bracketed by events (not part of `value`)
    rc.set_fused(False)
    rc.render_rays(rays, None, out=out)
    rc.set_profiling(1)
    for i in range(16):
        rc.render_rays(batches[i % N_BATCHES], None, out=out)      # single stream: undisturbed per-stage times
    torch.cuda.synchronize()
    stage = rc.stage_times_ms()
    rc.set_profiling(0)
    rc.set_fused(fused_mode)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    src_hash = rc_ext.source_hash()
    rays_total = RAYS_PER_BATCH * world * args.steps
    value = rays_total / elapsed
    flops = (FUSED_FLOP_PER_RAY if fused else SHADER_FLOP_PER_SAMPLE * SHADED_SAMPLES) * RAYS_PER_BATCH
    achieved_tf = flops / (sh_ms * 1e-3) / 1e12
    grid_ms = sum(stage[k] for k in GRID_BYTES)
    grid_bytes = sum(GRID_BYTES.values()) * RAYS_PER_BATCH
    grid_gbs = grid_bytes / (grid_ms * 1e-3) / 1e9
    pmc, pmc_src = measured_file("pmc_k_cache_fused.json" if fused else "pmc_k_cache_shader.json", src_hash)
    stamps, stamps_src = measured_file("fused_phase_stamps.json", src_hash)
    in_fused = None
    if stamps is not None:
        g_us = sum(stamps["phases_us"][k] for k in ("gather0", "gather1", "gather2"))
        in_fused = {"what": "gather phases inside k_cache_fused (in-kernel s_memtime stamps of the diagnostic build, "
                            "median over the 1024 rays; the level-2 phase looks the density and the appearance grid up together)",
                    "achieved": grid_bytes / (g_us * 1e-6) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": grid_bytes / (g_us * 1e-6) / 1e9 / PEAK_HBM_GBS, "sum_phase_us": g_us}
    # the same phases on coherent image rays (a 32x32-pixel tile / a 1024-pixel row strip of the 800x800 camera): the
    # regime render_image's chunks are in (profiles/r03_coherent_stamps.txt)
    coherent = {}
    for mode in ("tile", "strip"):
        st_c, src_c = measured_file(f"fused_phase_stamps_{mode}.json", src_hash)
        if st_c is not None:
            g_c = sum(st_c["phases_us"][k] for k in ("gather0", "gather1", "gather2"))
            coherent[mode] = {"achieved": grid_bytes / (g_c * 1e-6) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": grid_bytes / (g_c * 1e-6) / 1e9 / PEAK_HBM_GBS, "sum_phase_us": g_c,
                              "launch_us": st_c["total_us"], "source": src_c}
    res = {
        "metric": "rays/sec (1024-ray batch, 64+64 proposal + 32 shaded samples/ray), hotdog cache forward",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "hotdog cache render 1024 rays x (64,64,32) samples, cache-only passes, "
                               "synthetic rays + synthetic weights (configs[1])",
                   "rays_per_batch_per_gpu": RAYS_PER_BATCH, "distinct_batches_cycled": N_BATCHES,
                   "parallelism": f"ray-sharded x{world}",
                   "launch": "eager" if (args.profile_mode or args.graph_mode == 0 or fused) else "hipGraph",
                   "batches_in_flight": nstr,
                   "ms_per_step_runs": [e / args.steps * 1e3 for e in run_s],
                   "kernel_plan": {"fused": ("fused: 1 launch per batch, 1 wavefront per ray (k_cache_fused)" if arith == "bf16x3-split" else
                                             "fused: 1 launch per batch, 2 wavefronts per ray, 2 workgroups per CU (k_cache_fused_team)"),
                                   "fused1": "fused: 1 launch per batch, 1 wavefront per ray (k_cache_fused)",
                                   "staged": "staged: 13 launches per batch"}[args.plan],
                   "mlp_arithmetic": ("shader MLPs: fp32 operands split exactly into 3 bf16 pieces, 6 products per fp32 product on "
                                      "v_mfma_f32_32x32x16_bf16, fp32 accumulation (fp32-grade: the noise-floor parity test); "
                                      "density MLPs: v_mfma_f32_32x32x2_f32" if arith == "bf16x3-split" else
                                      "every MLP layer on v_mfma_f32_32x32x2_f32 (exact fp32 chain)"),
                   "kernel_source_hash": src_hash},
        "roofline": {"kernel": dominant, "bound": "mfma", "achieved": achieved_tf,
                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / PEAK_F32_MFMA_TFLOPS,
                     # `achieved` counts the ALGORITHMIC fp32 flops; `peak` stays the fp32 MFMA peak -- the precision the
                     # path delivers and the unit earlier rounds were priced in.  A split build issues the shader's share
                     # (8.32 of the 9.94 GFLOP) six times over on the bf16 pipe (2500 TF dense): `issued_bf16` prices that
                     "issued_bf16": (None if arith != "bf16x3-split" else
                                     {"flop_per_launch": 6 * 253824 * 32 * RAYS_PER_BATCH,
                                      "achieved": 6 * 253824 * 32 * RAYS_PER_BATCH / (sh_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                      "frac": 6 * 253824 * 32 * RAYS_PER_BATCH / (sh_ms * 1e-3) / 1e12 / 2500.0}),
                     # HBM-side bytes per launch: FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes (KiB as
                     # reported: random 4/16-byte gathers, the x2 correction for wide streaming reads does not apply);
                     # null when the file under profiles/ was not measured on these kernel sources
                     "traffic": None if pmc is None else (pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024,
                     "traffic_source": pmc_src,
                     "avg_launch_ms": sh_ms,
                     "algorithmic_flop_per_launch": flops},
        "hashgrid": {"kernels": "k_hashgrid_fwd x4 (grid0, grid1, grid2, grid_app), stand-alone: launch-per-stage plan, "
                                "separate pass (NOT the path `value` times)", "bound": "hbm",
                     "achieved": grid_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": grid_gbs / PEAK_HBM_GBS,
                     "sum_launch_ms": grid_ms, "algorithmic_bytes_per_batch": grid_bytes,
                     "in_fused_kernel": in_fused, "in_fused_source": stamps_src,
                     "in_fused_kernel_coherent": coherent or None},
        "stage_ms_separate_pass_staged_plan": stage,
        "image": image,
    }
    # the secondary lines must never cost the headline line: a failure is reported in place of the numbers
    def secondary(name, fn, *fargs):
        print(f"[bench] {name} line ...", file=sys.stderr, flush=True)
        try:
            return fn(*fargs)
        except Exception as e:      # noqa: BLE001
            print(f"[bench] {name} line failed: {e!r}", file=sys.stderr, flush=True)
            return {"error": repr(e)}

    if not args.no_transient and world == 1:
        res["transient"] = secondary("transient", transient_line, local_rank, dev)
    if not args.no_material and world == 1:
        res["material"] = secondary("material", material_line, local_rank, dev)
    if not args.no_train and world == 1:
        res["train_backward"] = secondary("train-backward", train_backward_line, local_rank, dev)
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = secondary("cpu_baseline", cpu_baseline, cfg, weights, RAYS_PER_BATCH)
    else:
        res["cpu_baseline"] = None
    # parity of what was timed: batch 0 of the timed loop against the fp32 oracle (256 of its rays: seconds on the host)
    res["parity"] = secondary("parity", parity_block, cfg, weights, rank, first)
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def parity_block(cfg, weights_np, rank, first, n_check=256):
    """max |rgb - oracle| and PSNR = -10 log10(mse) (internal/utils.py:50) of the first `n_check` rays of batch 0."""
    import numpy as np
    import torch

    import nrc_amd
    from oracle import cache_ref

    torch.set_num_threads(host_cpu_share())
    rays = nrc_amd.synthetic_rays(RAYS_PER_BATCH, seed=20200823 + 1000 * rank)
    rt = {k: torch.from_numpy(np.asarray(v)[:n_check]) for k, v in rays.hot_fields().items()}
    wt = {k: torch.from_numpy(v) for k, v in weights_np.items()}
    ref = cache_ref.cache_forward(wt, cfg, rt, None, want_grad_normals=False)["render"]
    d = (first["rgb"][:n_check].cpu() - ref["rgb"]).abs()
    mse = float((d ** 2).mean())
    return {"against": f"torch fp32 oracle on the first {n_check} rays of batch 0 (parity unpinned: the oracle is a "
                       "restatement, see oracle/__init__.py)",
            "max_abs_rgb": float(d.max()), "psnr_db": float(-10.0 * np.log10(max(mse, 1e-30))),
            "max_abs_acc": float((first["acc"][:n_check].cpu() - ref["acc"]).abs().max()), "tolerance": 1e-4}


if __name__ == "__main__":
    main()
