"""Training backward of the density field: oracle self-checks on CPU, the HIP path against
the oracle on the GPU, and the data-parallel gradient reduction on 2 gloo ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import common
import nrc_amd
from oracle import train_ref

CFG = nrc_amd.hotdog_config()


def _points(n, seed=3, spread=1.2):
    rng = np.random.Generator(np.random.PCG64(seed))
    p = rng.normal(size=(n, 3)).astype(np.float32) * spread      # some beyond the contraction radius
    p[:4] = 0.0                                                  # coincident samples: colliding table updates
    return p


def _points_off_kinks(level, n, seed=3, margin=2e-4):
    """Points whose hidden pre-activations all stay `margin` away from 0 (float64 oracle): at a ReLU kink the
    gradient jumps, and a float32 forward pass (errors ~1e-5 here: contracted coordinates times grid sizes up to 2048
    into white-noise tables) may sit on the other side -- in the torch float32 oracle just as in the HIP kernel."""
    pts = _points(n + n // 4, seed)
    w = common.weights_torch(dtype=torch.float64)
    m = train_ref.relu_margin(w, CFG, level, torch.from_numpy(pts).double()).numpy()
    keep = np.nonzero(m > margin)[0][:n]
    assert len(keep) == n
    return np.ascontiguousarray(pts[keep])


def _upstream(n, seed=4):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.normal(size=(n,)).astype(np.float32), (rng.normal(size=(n, 64)) * 0.1).astype(np.float32)


def test_oracle_backward_matches_finite_differences():
    """Reverse-mode restatement vs central differences of the oracle's own forward (float64)."""
    w = common.weights_torch(dtype=torch.float64)
    pts = torch.from_numpy(_points(24)).double()
    dd, df = (torch.from_numpy(a).double() for a in _upstream(24))
    level = 2
    grads, _, _ = train_ref.density_backward(w, CFG, level, pts, dd, df)

    def loss(ww):
        g, dens, feat = train_ref.density_backward(ww, CFG, level, pts, dd, df)
        return float((dd * dens).sum() + (df * feat).sum())

    rng = np.random.Generator(np.random.PCG64(0))
    checked = 0
    for name in ("params/Cache/Sampler/MLP_2/density_layers_0/kernel", "params/Cache/Sampler/MLP_2/density_layers_1/bias",
                 "params/Cache/Sampler/MLP_2/output_density_layer/kernel"):
        g = grads[name]
        for _ in range(3):
            idx = tuple(int(rng.integers(0, s)) for s in g.shape)
            eps = 1e-5
            wp = dict(w); wp[name] = w[name].clone(); wp[name][idx] += eps
            wm = dict(w); wm[name] = w[name].clone(); wm[name][idx] -= eps
            fd = (loss(wp) - loss(wm)) / (2 * eps)
            assert abs(fd - float(g[idx])) <= 1e-5 * max(1.0, abs(fd)), (name, idx, fd, float(g[idx]))
            checked += 1
    # a table entry that is actually touched
    tname = next(k for k in grads if "density_grid/grid_" in k)
    g = grads[tname]
    nz = torch.nonzero(g)
    assert len(nz) > 0
    idx = tuple(int(v) for v in nz[len(nz) // 2])
    eps = 1e-5
    wp = dict(w); wp[tname] = w[tname].clone(); wp[tname][idx] += eps
    wm = dict(w); wm[tname] = w[tname].clone(); wm[tname][idx] -= eps
    fd = (loss(wp) - loss(wm)) / (2 * eps)
    assert abs(fd - float(g[idx])) <= 1e-5 * max(1.0, abs(fd))
    assert checked == 9


def test_safe_exp_gradient_ignores_the_clip():
    """math.safe_exp's custom_jvp: y_dot = y x_dot with the clipped y (internal/math.py:153-171)."""
    x = torch.tensor([0.5, 80.0], dtype=torch.float64, requires_grad=True)
    y = train_ref._SafeExp.apply(x)
    (g,) = torch.autograd.grad(y.sum(), x)
    assert torch.allclose(g, torch.exp(torch.tensor([0.5, 70.0], dtype=torch.float64)))


def _oracle_flat(level, pts, dd, df, layout, dtype=torch.float64):
    w = common.weights_torch(dtype=dtype)
    g, dens, _ = train_ref.density_backward(w, CFG, level, torch.from_numpy(pts).to(dtype), torch.from_numpy(dd).to(dtype),
                                            None if df is None else torch.from_numpy(df).to(dtype))
    flat = np.zeros(sum(int(np.prod(s)) for _, _, s in layout), np.float64)
    for name, off, shape in layout:
        flat[off: off + int(np.prod(shape))] = g[name].double().numpy().reshape(-1)
    return flat, dens.double().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 1, 2])
@pytest.mark.parametrize("with_feature", [False, True])
def test_density_backward_matches_oracle(level, with_feature):
    rc = common.make_rc()
    n = 1000                                   # ragged: not a multiple of 32 / 64
    pts = _points_off_kinks(level, n)
    dd, df = _upstream(n)
    df = df if with_feature else None
    layout, total = rc.density_grad_layout(level)
    assert layout[-1][0].endswith("output_density_layer/bias") and total == layout[-1][1] + 1
    flat, dens = rc.density_backward(level, pts, dd, df)
    ref, dens_ref = _oracle_flat(level, pts, dd, df, layout)
    np.testing.assert_allclose(dens.cpu().numpy(), dens_ref, rtol=2e-4, atol=1e-6)
    got = flat.cpu().numpy().astype(np.float64)
    for name, off, shape in layout:
        sz = int(np.prod(shape))
        a, b = got[off: off + sz], ref[off: off + sz]
        scale = max(1e-12, float(np.abs(b).max()))
        # float32 against the float64 oracle; the torch float32 oracle sits at 2e-4 of the scale itself (tools/dbg_train.py)
        assert float(np.abs(a - b).max()) <= 5e-4 * scale + 1e-7, (name, float(np.abs(a - b).max()), scale)
        if "density_grid" in name:
            assert np.count_nonzero(a) == np.count_nonzero(b), name                     # the same entries are touched


@pytest.mark.gpu
def test_density_backward_accumulates_and_is_linear():
    rc = common.make_rc()
    pts = _points(512, seed=8)
    dd, df = _upstream(512, seed=9)
    g1, _ = rc.density_backward(2, pts, dd, df)
    g2, _ = rc.density_backward(2, pts, 2.0 * dd, 2.0 * df)
    np.testing.assert_allclose(g2.cpu().numpy(), 2.0 * g1.cpu().numpy(), rtol=1e-5, atol=1e-7)
    acc = g1.clone()
    rc.density_backward(2, pts, dd, df, grads=acc)
    np.testing.assert_allclose(acc.cpu().numpy(), 2.0 * g1.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # MLP gradients are reduced in a fixed order: bit-stable from call to call
    layout, _ = rc.density_grad_layout(2)
    off = next(o for name, o, _ in layout if name.endswith("density_layers_0/kernel"))
    g3, _ = rc.density_backward(2, pts, dd, df)
    assert torch.equal(g1[off:], g3[off:])
    with pytest.raises(nrc_amd.rc_ext.RcError):
        rc.density_backward(7, pts, dd)
    # empty batch
    g0, d0 = rc.density_backward(2, np.zeros((0, 3), np.float32), np.zeros((0,), np.float32))
    assert d0.numel() == 0 and float(g0.abs().max()) == 0.0


@pytest.mark.gpu
def test_density_backward_full_level_batch():
    """BASELINE batch: 1024 rays x 32 samples of level 2 in one call; column sums tie the pieces together
    (d b_out = sum of g_raw, independent of the MFMA path)."""
    rc = common.make_rc()
    n = 1024 * 32
    pts = _points(n, seed=12, spread=0.6)
    dd, _ = _upstream(n, seed=13)
    layout, _ = rc.density_grad_layout(2)
    flat, dens = rc.density_backward(2, pts, dd)
    g = nrc_amd.train.grads_as_dict(flat, layout)
    bout = float(g["params/Cache/Sampler/MLP_2/output_density_layer/bias"][0])
    expect = float((torch.from_numpy(dd).cuda().double() * dens.double()).sum())
    assert abs(bout - expect) <= 1e-4 * max(1.0, abs(expect))
    assert all(bool(torch.isfinite(v).all()) for v in g.values())


def test_allreduce_grads_two_ranks_gloo(tmp_path):
    """pmean of the flat gradient buffers over 2 ranks (gloo on CPU)."""
    script = tmp_path / "ar.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "import nrc_amd\n"
        "dist.init_process_group('gloo')\n"
        "r = dist.get_rank()\n"
        "bufs = [torch.full((1000,), float(r + 1)), torch.arange(10, dtype=torch.float32) * (r + 1)]\n"
        "nrc_amd.train.allreduce_grads(bufs)\n"
        "assert torch.allclose(bufs[0], torch.full((1000,), 1.5)), bufs[0][:3]\n"
        "assert torch.allclose(bufs[1], torch.arange(10, dtype=torch.float32) * 1.5)\n"
        "dist.destroy_process_group()\n"
        "print('ok', r)\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


@pytest.mark.gpu
def test_gradient_descent_on_the_density_field_reduces_the_loss():
    """A few plain SGD steps driven by rc_density_backward: fit level 2's density at fixed points to a target field.
    Exercises the whole loop a trainer would run -- forward value, upstream gradient, gradient buffer by tensor name,
    parameter update on the device, rc_load_weights of the changed tensors (device pointers), repacked weight stream."""
    rc = common.make_rc()
    level = 2
    pts = torch.from_numpy(_points(4096, seed=21, spread=0.5)).cuda()
    layout, total = rc.density_grad_layout(level)
    params = {k: torch.from_numpy(v).cuda() for k, v in common.weights_np().items() if any(k == name for name, _, _ in layout)}
    assert set(params) == {name for name, _, _ in layout}
    zeros = torch.zeros(pts.shape[0], device="cuda")
    _, dens0 = rc.density_backward(level, pts, zeros)
    target = 0.5 * dens0 + 0.05                         # a reachable field
    flat = torch.zeros(total, dtype=torch.float32, device="cuda")
    losses = []
    lr = {"grid": 5.0, "mlp": 5e-2}
    for step in range(16):
        _, dens = rc.density_backward(level, pts, zeros)           # forward value only (zero upstream: no gradient)
        diff = dens - target
        losses.append(float((diff * diff).mean()))
        flat.zero_()
        g, _, _ = nrc_amd.train.density_grads(rc, level, pts, 2.0 * diff / diff.numel(), None, flat)
        for name, grad in g.items():
            params[name] -= (lr["grid"] if "density_grid" in name else lr["mlp"]) * grad
        rc.load_weights(params)
    assert losses[-1] < 0.6 * losses[0], losses
    assert all(b <= a * 1.02 for a, b in zip(losses, losses[1:])), losses


@pytest.mark.gpu
@pytest.mark.parametrize("grid_id", [1, 3])
def test_hashgrid_backward_is_the_transpose_of_the_lookup(grid_id):
    """rc_hashgrid_backward against autograd through the oracle's hash encoding, and <lookup(T), d> == <T, backward(d)>."""
    from oracle import hashgrid_ref, mathx
    from oracle.cache_ref import P
    rc = common.make_rc()
    n = 777
    pts = _points(n, seed=31)
    g = rc.cfg_grid(grid_id)
    rng = np.random.Generator(np.random.PCG64(32))
    d = rng.normal(size=(n, g.out_dim)).astype(np.float32)
    layout, total = rc.hashgrid_grad_layout(grid_id)
    flat = rc.hashgrid_backward(grid_id, pts, d)
    # oracle: autograd of sum(d * encoding) w.r.t. the tables
    w = common.weights_torch(dtype=torch.float64)
    prefix = layout[0][0].rsplit("/", 1)[0]
    names = [name for name, _, _ in layout]
    ww = dict(w)
    for k in names:
        ww[k] = w[k].detach().clone().requires_grad_(True)
    x = hashgrid_ref.hash_encoding(ww, prefix, g, mathx.contract_radius(torch.from_numpy(pts).double(), CFG.contract_radius))
    grads = torch.autograd.grad((torch.from_numpy(d).double() * x).sum(), [ww[k] for k in names])
    got = flat.cpu().numpy().astype(np.float64)
    for (name, off, shape), gr in zip(layout, grads):
        ref = gr.numpy().reshape(-1)
        a = got[off: off + ref.size]
        scale = max(1e-12, float(np.abs(ref).max()))
        # float32 trilinear weights at grid sizes up to 2048: a coordinate ulp is 1e-4 of a cell (2.2e-4 measured at 2048)
        assert float(np.abs(a - ref).max()) <= 5e-4 * scale + 1e-7, (name, float(np.abs(a - ref).max()), scale)
    # adjoint identity with the device's own forward lookup
    tables = torch.cat([torch.from_numpy(common.weights_np()[name]).reshape(-1) for name in names]).cuda()
    fwd = rc.hashgrid_lookup(grid_id, pts)
    lhs = float((fwd.double() * torch.from_numpy(d).cuda().double()).sum())
    rhs = float((tables.double() * flat.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    assert P in names[0]


@pytest.mark.gpu
@pytest.mark.parametrize("grid_id", [0, 1])
def test_sliced_scatter_at_batch_size_adjoint_and_level_sums(grid_id):
    """k_grid_scatter_sliced (F = 1 grids: every level through LDS slices, whole-row flush) at the size it is built for --
    65 536 + 37 points, every (level, slice, point range) workgroup of the plan busy -- through two size-independent
    properties: the adjoint identity <lookup(T), d> == <T, backward(d)> with the device's own forward lookup, and, per
    level, sum of the level's table gradient == precondition * sum of d[:, level] (the eight trilinear weights of a point
    sum to one; the points are kept 0.05 of the box away from its faces, so no corner of a dense level is zero padding)."""
    rc = common.make_rc()
    n = 65536 + 37
    pts = np.clip(_points(n, seed=41, spread=0.5), -1.8, 1.8)
    g = rc.cfg_grid(grid_id)
    rng = np.random.Generator(np.random.PCG64(42))
    d = rng.normal(size=(n, g.out_dim)).astype(np.float32)
    layout, total = rc.hashgrid_grad_layout(grid_id)
    flat = rc.hashgrid_backward(grid_id, pts, d)
    assert bool(torch.isfinite(flat).all())
    names = [name for name, _, _ in layout]
    tables = torch.cat([torch.from_numpy(common.weights_np()[name]).reshape(-1) for name in names]).cuda()
    fwd = rc.hashgrid_lookup(grid_id, pts)
    lhs = float((fwd.double() * torch.from_numpy(d).cuda().double()).sum())
    rhs = float((tables.double() * flat.double()).sum())
    assert abs(lhs - rhs) <= 2e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    dsum = torch.from_numpy(d).double().sum(0).numpy()
    for l, (name, off, shape) in enumerate(layout):
        got = float(flat[off: off + int(np.prod(shape))].double().sum())
        want = float(g.precondition_scaling * dsum[l])
        assert abs(got - want) <= 2e-3 * max(1.0, float(np.abs(d[:, l]).sum()) * g.precondition_scaling * 1e-3), (name, got, want)
