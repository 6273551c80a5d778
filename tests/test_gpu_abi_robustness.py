"""Host-side robustness of the C ABI on the device: graph replay after workspace growth, more caller streams than
workspace sets, entry points sharing a workspace set from different streams, error codes instead of exceptions."""
import numpy as np
import pytest
import torch

import common
import nrc_amd

pytestmark = pytest.mark.gpu

STAGED = dict(outputs=["rgb", "acc", "distance_median", "normals_pred"])


def _staged_rc(weights):
    from nrc_amd import rc_ext
    rc = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
    rc.load_weights(weights)
    rc.set_fused(False)            # the launch-per-stage plan: workspace + hipGraph replay
    return rc


def test_graph_replay_survives_workspace_growth_by_another_entry_point():
    """A captured small-n render, then rc_render_material with a larger n (reallocates workspace set 0), then the small
    render again: must equal an eager launch (the stale graph would read freed buffers)."""
    from oracle import material_ref
    cfg = nrc_amd.hotdog_config()
    rc = _staged_rc(common.weights_material_np(True))
    rc.set_graph_mode(2)           # capture on first sight
    small = nrc_amd.synthetic_rays(64, seed=5).hot_fields()
    bufs = rc.render_rays(small, None, **STAGED)
    first = {k: v.clone() for k, v in rc.render_rays(small, None, out=bufs).items()}     # replay of the capture
    torch.cuda.synchronize()
    n = 256
    rays = nrc_amd.synthetic_rays(n, seed=6)
    rc.render_material(rays.hot_fields(), material_ref.draw_randoms(cfg, n, seed=1))      # grows every buffer
    again = rc.render_rays(small, None, out=bufs)
    torch.cuda.synchronize()
    again = {k: v.clone() for k, v in again.items()}
    rc.set_graph_mode(0)
    eager = rc.render_rays(small, None, **STAGED)
    torch.cuda.synchronize()
    for k in eager:
        assert torch.equal(first[k], eager[k]), k
        assert torch.equal(again[k], eager[k]), k


def test_more_streams_than_workspace_sets():
    """Six caller streams on the launch-per-stage plan (four workspace sets): the fifth and sixth take over the least
    recently used sets and are ordered behind their previous users; every result equals the single-stream one."""
    rc = _staged_rc(common.weights_np())
    rc.set_graph_mode(0)
    n = 512
    batches = [nrc_amd.synthetic_rays(n, seed=100 + i).hot_fields() for i in range(6)]
    dev = [{k: rc._dev(v) for k, v in b.items() if v is not None} for b in batches]
    want = []
    for b in dev:
        want.append({k: v.clone() for k, v in rc.render_rays(b, None, **STAGED).items()})
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(6)]
    for rep in range(3):
        got = []
        for s, b in zip(streams, dev):
            with torch.cuda.stream(s):
                got.append(rc.render_rays(b, None, **STAGED))
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            for k in w:
                assert torch.equal(g[k], w[k]), (rep, k)


def test_material_and_cache_calls_from_two_streams_share_set_zero():
    """rc_render_material always works in workspace set 0; a staged rc_render_rays on another stream that owns set 0 is
    ordered against it instead of racing."""
    from oracle import material_ref
    cfg = nrc_amd.hotdog_config()
    rc = _staged_rc(common.weights_material_np(True))
    rc.set_graph_mode(0)
    n = 256
    rays = nrc_amd.synthetic_rays(n, seed=8)
    rnd = material_ref.draw_randoms(cfg, n, seed=2)
    fields = {k: rc._dev(v) for k, v in rays.hot_fields().items() if v is not None}
    want_c = {k: v.clone() for k, v in rc.render_rays(fields, None, **STAGED).items()}
    _, want_m = rc.render_material(fields, rnd)
    want_m = {k: v.clone() for k, v in want_m.items()}
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            got_c = rc.render_rays(fields, None, **STAGED)
        with torch.cuda.stream(s2):
            _, got_m = rc.render_material(fields, rnd)
        torch.cuda.synchronize()
        for k in want_c:
            assert torch.equal(got_c[k], want_c[k]), k
        for k in want_m:
            assert torch.equal(got_m[k], want_m[k]), k


def test_errors_come_back_as_codes():
    from nrc_amd import rc_ext
    rc = rc_ext.RadianceCache(nrc_amd.hotdog_config(), 0)
    with pytest.raises(rc_ext.RcError) as e:
        rc.render_rays(nrc_amd.synthetic_rays(4).hot_fields(), None)        # nothing loaded
    assert e.value.code == -3 and "missing weight" in str(e.value)
    with pytest.raises(rc_ext.RcError) as e:
        rc.load_weights({"params/Cache/Nope/kernel": np.zeros((2, 2), np.float32)})
    assert e.value.code == -1


def _loaded_rccl():
    """Path of the RCCL instance this process has mapped (torch's own copy), so that the communicator made below and
    rc_allgather_outputs talk to the same library."""
    import torch.distributed  # noqa: F401  (makes sure libtorch_hip and its RCCL are in)
    for line in open("/proc/self/maps"):
        if "librccl" in line:
            return line.split()[-1]
    return None


def test_allgather_outputs_over_rccl_world_of_one():
    """rc_allgather_outputs with a real RCCL communicator (one rank: this box has one GPU): the grouped collective runs
    on the caller's stream and `full` equals `local` for every gathered slot; slots missing on one side are skipped."""
    import ctypes as C
    import os
    path = _loaded_rccl() or "/opt/rocm/lib/librccl.so"
    os.environ["RC_RCCL_LIBRARY"] = path
    rccl = C.CDLL(path)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        rc = common.make_rc()
        n = 300
        local = rc.render_rays(nrc_amd.synthetic_rays(n, seed=3).hot_fields(), None, outputs=["rgb", "acc", "normals_pred"])
        full = rc.allgather_outputs(comm.value, local, world=1)
        torch.cuda.synchronize()
        for k in local:
            assert torch.equal(full[k], local[k]), k
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("n", [1, 5, 257, 2048])
def test_level_kernels_equal_the_separate_gather_and_mlp_kernels(n):
    """rc_set_fused(2): a proposal level that only hands its density on runs as ONE launch (rc_level.hip: grid lookup +
    density MLP, weights resident in LDS); rc_set_fused(0): k_hashgrid_fwd + k_density_mlp.  Same arithmetic in the same
    order: primary rays (levels 0 / 1), secondary and resampled rays (all three levels on the lean pass) bitwise equal."""
    from nrc_amd import rc_ext
    rc = common.make_rc()
    rc.set_graph_mode(0)
    rays = nrc_amd.synthetic_rays(n, seed=31).hot_fields()
    jit = common.jitters(n, seed=9)
    srays, srnd = common.secondary_case(n, seed=12)
    g = np.random.default_rng(4).gumbel(size=(n, 32)).astype(np.float32)
    cases = [(rays, {"jitter": jit}, rc_ext.RC_PASS_CACHE, None),
             (rays, {"jitter": jit, "gumbel": g}, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_RESAMPLE, ["rgb", "acc", "distance_median", "means"]),
             (srays, srnd, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_SECONDARY | rc_ext.RC_PASS_NO_ENVMAP, ["rgb", "acc", "distance_mean"])]
    for fields, rnd, mask, outs in cases:
        res = {}
        for mode in (0, 2):
            rc.set_fused(mode)
            o = rc.render_rays(fields, rnd, mask, outputs=outs)
            torch.cuda.synchronize()
            res[mode] = {k: v.clone() for k, v in o.items()}
            res[mode]["density0"] = torch.from_numpy(rc.workspace("density0")[: n * 64].copy())
            res[mode]["density2"] = torch.from_numpy(rc.workspace("density2")[: n * 32].copy())
        for k in res[0]:
            assert torch.equal(res[0][k].cpu(), res[2][k].cpu()), (mask, k)
    rc.set_fused(1)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 3, 130, 257, 768])
def test_material_stage_is_bitwise_the_same_on_every_launch_plan(n):
    """rc_render_material on the launch plans of rc_set_fused: 1 (default) runs the primary cache pass as the ONE fused
    launch (two wavefronts per ray) with its per-sample results exported for the shading-point pick, 3 the same on the
    one-wavefront-per-ray kernel, 2 the launch-per-stage pass with the level kernels, 0 one kernel per stage.  Every cache and material output, the picks and the secondary radiance are
    bitwise equal.  (768 primary rays = 24 576 secondary rays: from there on plan 1 runs a proposal level of the trace
    with its sampling in front as ONE launch, one ray per wave -- k_level_ray.)"""
    from oracle import material_ref
    from nrc_amd import rc_ext
    cfg = nrc_amd.hotdog_config()
    rc = rc_ext.RadianceCache(cfg, 0)
    rc.load_weights(common.weights_material_np(False))
    rays = nrc_amd.synthetic_rays(n, seed=5)
    rnd = material_ref.draw_randoms(cfg, n, seed=8)
    res = {}
    for mode in (0, 2, 3, 1):
        rc.set_fused(mode)
        cres, mres = rc.render_material(rays.hot_fields(), rnd)
        torch.cuda.synchronize()
        res[mode] = {"c:" + k: v.clone() for k, v in cres.items()}
        res[mode].update({"m:" + k: v.clone() for k, v in mres.items()})
        res[mode]["inds"] = torch.from_numpy(rc.workspace("inds", np.int32)[:n].copy())
        res[mode]["s:inds"] = torch.from_numpy(rc.workspace("s:inds", np.int32)[: n * 32].copy())
        res[mode]["sec_rgb"] = torch.from_numpy(rc.workspace("sec_rgb")[: n * 96].copy())
        res[mode]["weights2"] = torch.from_numpy(rc.workspace("weights2")[: n * 32].copy())
    for mode in (2, 3, 1):
        for k in res[0]:
            assert torch.equal(res[0][k].cpu(), res[mode][k].cpu()), (mode, k)


def test_level_kernel_with_its_sampling_in_front_equals_the_separate_kernels():
    """rc_set_fused(1) on a batch of >= 24 576 rays: k_level_ray (sample_level_ray + the level's lookup and MLP, one ray per
    wave) against k_sample_level + k_level (plan 2) and the one-kernel-per-stage plan (0), on a ragged ray count, for the
    resampling pass of primary rays and for secondary rays: fence posts, densities and outputs bitwise equal."""
    from nrc_amd import rc_ext
    rc = common.make_rc()
    rc.set_graph_mode(0)
    n = 25001
    rays = nrc_amd.synthetic_rays(n, seed=41).hot_fields()
    jit = common.jitters(n, seed=6)
    srays, srnd = common.secondary_case(n, seed=13)
    g = np.random.default_rng(5).gumbel(size=(n, 32)).astype(np.float32)
    cases = [(rays, {"jitter": jit, "gumbel": g}, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_RESAMPLE, ["rgb", "acc", "means"]),
             (srays, srnd, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_SECONDARY | rc_ext.RC_PASS_NO_ENVMAP, ["rgb", "acc"])]
    for fields, rnd, mask, outs in cases:
        res = {}
        for mode in (0, 2, 1):
            rc.set_fused(mode)
            o = rc.render_rays(fields, rnd, mask, outputs=outs)
            torch.cuda.synchronize()
            res[mode] = {k: v.clone() for k, v in o.items()}
            for nm, cnt in (("density0", n * 64), ("density1", n * 64), ("density2", n * 32), ("tdist0", n * 65),
                            ("sdist1", n * 65), ("tdist2", n * 33), ("means1", 3 * n * 64)):
                res[mode][nm] = torch.from_numpy(rc.workspace(nm)[:cnt].copy())
        for mode in (2, 1):
            for k in res[0]:
                assert torch.equal(res[0][k].cpu(), res[mode][k].cpu()), (mask, mode, k)
    rc.set_fused(1)


@pytest.mark.gpu
def test_level_kernels_run_time_layout_form_equals_the_compile_time_form(monkeypatch):
    """rc_level.hip has two forms of the F = 1 level lookup: kinds of the grid levels compile-time for the reference's
    layout (three dense levels with cell tables, then hashed ones; loads of a level pair split by corner between the
    half-waves) and read from the level records for any other layout.  RC_LEVEL_ANY_LAYOUT=1 makes the launcher pick the
    second on the reference's layout too: same densities, same outputs, bit for bit (k_level behind k_sample_level on
    plan 2, k_level_ray on plan 1 from 24 576 rays on)."""
    from nrc_amd import rc_ext
    rc = common.make_rc()
    rc.set_graph_mode(0)
    n = 24577
    rays = nrc_amd.synthetic_rays(n, seed=77).hot_fields()
    rnd = {"jitter": common.jitters(n, seed=9), "gumbel": np.random.default_rng(3).gumbel(size=(n, 32)).astype(np.float32)}
    mask = rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_RESAMPLE
    res = {}
    for form in ("0", "1"):
        monkeypatch.setenv("RC_LEVEL_ANY_LAYOUT", form)
        for mode in (2, 1):
            rc.set_fused(mode)
            o = rc.render_rays(rays, rnd, mask, outputs=["rgb", "acc"])
            torch.cuda.synchronize()
            r = {k: v.clone() for k, v in o.items()}
            for nm, cnt in (("density0", n * 64), ("density1", n * 64), ("density2", n * 32)):
                r[nm] = torch.from_numpy(rc.workspace(nm)[:cnt].copy())
            res[(form, mode)] = r
    monkeypatch.delenv("RC_LEVEL_ANY_LAYOUT")
    rc.set_fused(1)
    for mode in (2, 1):
        for k in res[("0", mode)]:
            assert torch.equal(res[("0", mode)][k].cpu(), res[("1", mode)][k].cpu()), (mode, k)
    assert float(res[("0", 1)]["density0"].abs().sum()) > 0.0


@pytest.mark.gpu
def test_render_chunks_equals_one_call_per_chunk():
    """rc_render_chunks (the chunk loop of render_image inside the library, chunk i on stream i % 2, outputs into row i of
    one arena) against one rc_render_rays per chunk: bitwise the same rows; passes that need random inputs are refused."""
    from nrc_amd import rc_ext
    rc = common.make_rc()
    chunk, n_chunks = 257, 5
    rays = nrc_amd.synthetic_rays(chunk * n_chunks, seed=19).hot_fields()
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in rays.items() if k != "lossmult"}
    keys = ["rgb", "acc", "distance_median", "normals_pred"]
    plan = rc.output_plan(keys, chunk)
    total = plan[0]
    ref = torch.zeros((n_chunks, total), dtype=torch.float32, device="cuda")
    for i in range(n_chunks):
        f = {k: v[i * chunk:(i + 1) * chunk].contiguous() for k, v in dev.items()}
        rc.render_chunk(f, None, rc_ext.RC_PASS_CACHE, plan, ref[i])
    torch.cuda.synchronize()
    arena = torch.zeros_like(ref)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ev = torch.cuda.Event(); ev.record()
    for s_ in streams:
        s_.wait_event(ev)
    rc.render_chunks(dev, chunk, n_chunks, rc_ext.RC_PASS_CACHE, plan, arena, [s_.cuda_stream for s_ in streams])
    for s_ in streams:
        s_.synchronize()
    assert torch.equal(arena, ref)
    assert float(arena.abs().sum()) > 0.0
    with pytest.raises(rc_ext.RcError):
        rc.render_chunks(dev, chunk, n_chunks, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_RESAMPLE, plan, arena,
                         [s_.cuda_stream for s_ in streams])


@pytest.mark.gpu
def test_pick_composite_equals_the_general_composite():
    """With one resampled sample per ray and only rgb / acc requested the compositing is finished by one thread per ray
    from k_resample's weight sum (rc_launch_composite_pick); any further output selects the wave-per-ray k_composite.
    rgb and acc are the same bits either way: primary rays with forced resampling, and secondary rays."""
    from nrc_amd import rc_ext
    rc = common.make_rc()
    n = 3001
    rays = nrc_amd.synthetic_rays(n, seed=23).hot_fields()
    rnd = {"jitter": common.jitters(n, seed=4), "gumbel": np.random.default_rng(8).gumbel(size=(n, 32)).astype(np.float32)}
    srays, srnd = common.secondary_case(n, seed=29)
    for fields, r, mask, extra in ((rays, rnd, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_RESAMPLE, "means"),
                                   (srays, srnd, rc_ext.RC_PASS_CACHE | rc_ext.RC_PASS_SECONDARY | rc_ext.RC_PASS_NO_ENVMAP, "distance_mean")):
        a = {k: v.clone() for k, v in rc.render_rays(fields, r, mask, outputs=["rgb", "acc"]).items()}
        b = {k: v.clone() for k, v in rc.render_rays(fields, r, mask, outputs=["rgb", "acc", extra]).items()}
        torch.cuda.synchronize()
        assert torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["acc"], b["acc"]), mask
        assert float(a["rgb"].abs().sum()) > 0.0 and float(a["acc"].sum()) > 0.0


@pytest.mark.gpu
def test_material_stage_with_per_stage_profiling_on_takes_the_envmap_along():
    """The EnvMap of the secondary trace is released where the LAST level kernel of the trace is launched.  With per-stage
    profiling on the trace runs one kernel per stage -- a plan without that spot -- and the stage must release the EnvMap
    itself: same outputs as without profiling, bit for bit (768 primary rays = 24 576 secondary rays)."""
    from oracle import material_ref
    from nrc_amd import rc_ext
    cfg = nrc_amd.hotdog_config()
    rc = rc_ext.RadianceCache(cfg, 0)
    rc.load_weights(common.weights_material_np(False))
    n = 768
    rays = nrc_amd.synthetic_rays(n, seed=5)
    rnd = material_ref.draw_randoms(cfg, n, seed=8)
    res = []
    for prof in (0, 1, 0):
        rc.set_profiling(prof)
        cres, mres = rc.render_material(rays.hot_fields(), rnd)
        torch.cuda.synchronize()
        res.append({**{"c:" + k: v.clone() for k, v in cres.items()}, **{"m:" + k: v.clone() for k, v in mres.items()}})
    rc.set_profiling(0)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]) and torch.equal(res[0][k], res[2][k]), k
    assert any(float(v.abs().sum()) > 0 for k, v in res[0].items() if k.startswith("m:"))


def test_f4_density_records_can_be_left_out(tmp_path):
    """RC_REC4_TABLES=0 (read once per process, hence the child process): the handle is built without the 2.5 GB of cell
    records of the F = 4 density grid and the level kernels of the secondary trace read the hash tables instead -- every
    output of the material stage is bitwise what the handle with the records renders."""
    import os
    import subprocess
    import sys
    from oracle import material_ref
    from nrc_amd import rc_ext
    n = 130
    cfg = nrc_amd.hotdog_config()
    rc = rc_ext.RadianceCache(cfg, 0)
    rc.load_weights(common.weights_material_np(False))
    rays = nrc_amd.synthetic_rays(n, seed=5)
    rnd = material_ref.draw_randoms(cfg, n, seed=8)
    cres, mres = rc.render_material(rays.hot_fields(), rnd)
    torch.cuda.synchronize()
    want = {"c:" + k: v.cpu().numpy() for k, v in cres.items()}
    want.update({"m:" + k: v.cpu().numpy() for k, v in mres.items()})
    out = tmp_path / "norec.npz"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = f"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
import numpy as np, torch
import common, nrc_amd
from nrc_amd import rc_ext
from oracle import material_ref
cfg = nrc_amd.hotdog_config()
rc = rc_ext.RadianceCache(cfg, 0)
rc.load_weights(common.weights_material_np(False))
rays = nrc_amd.synthetic_rays({n}, seed=5)
cres, mres = rc.render_material(rays.hot_fields(), material_ref.draw_randoms(cfg, {n}, seed=8))
torch.cuda.synchronize()
got = {{"c:" + k: v.cpu().numpy() for k, v in cres.items()}}
got.update({{"m:" + k: v.cpu().numpy() for k, v in mres.items()}})
np.savez({str(out)!r}, **got)
"""
    subprocess.run([sys.executable, "-c", child], check=True, env=dict(os.environ, RC_REC4_TABLES="0"), timeout=600)
    got = np.load(out)
    assert set(got.files) == set(want)
    for k, v in want.items():
        assert np.array_equal(got[k], v, equal_nan=True), k
