"""N>1 path on CPU: two gloo ranks shard an image's rays, render their share with the CPU
oracle standing in for the device model, and all-gather once.  The result must equal the
single-process render."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_apply(cfg, weights):
    from oracle import cache_ref

    def apply(rng, rays):
        r = {k: torch.from_numpy(np.asarray(v)) for k, v in rays.hot_fields().items()}
        out = cache_ref.cache_forward(weights, cfg, r, None, want_grad_normals=False)["render"]
        return {"render": out}
    return apply


class _Cfg:
    render_chunk_size = 16


def _worker(rank, world, port, H, W, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nrc_amd
    import common
    from nrc_amd import model as M
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = nrc_amd.hotdog_config()
    rays = nrc_amd.synthetic_camera_rays(H, W)
    out = M.render_image_distributed(_oracle_apply(cfg, common.weights_torch()), None, rays, _Cfg(),
                                     keys=("rgb", "acc", "distance_median"))
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_render_equals_single_process():
    import nrc_amd
    import common
    from nrc_amd import model as M
    H, W = 5, 9                                   # 45 rays: uneven split (23 + 22), ragged chunks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, H, W, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=500)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = nrc_amd.hotdog_config()
    rays = nrc_amd.synthetic_camera_rays(H, W)
    single = M.render_image_distributed(_oracle_apply(cfg, common.weights_torch()), None, rays, _Cfg(),
                                        keys=("rgb", "acc", "distance_median"))
    assert got["rgb"].shape == (H, W, 3) and got["acc"].shape == (H, W)
    for k in got:
        assert np.allclose(got[k], single[k].numpy(), atol=1e-6), k   # same arithmetic per ray (batch shapes differ)
    # to_host=True: what render_image returns (numpy on the host), same values, caller-owned arrays
    host = M.render_image_distributed(_oracle_apply(cfg, common.weights_torch()), None, rays, _Cfg(),
                                      keys=("rgb", "acc", "distance_median"), to_host=True)
    for k in single:
        assert isinstance(host[k], np.ndarray) and host[k].dtype == np.float32
        assert np.array_equal(host[k], single[k].numpy()), k


# ---- num_repeats > 1: the running mean covers the reference's stat_keys only (internal/models.py:2398-2401, 2473-2490)
class _RepeatApply:
    """A render whose result depends on how often it has been called for the current chunk: repeat i adds i to every
    key and turns the predicted normal.  Stat keys (rgb, acc) must come back as the mean over the repeats, every other
    key (distance_median, normals_pred) as the FIRST repeat's value -- on every path."""

    def __init__(self, num_repeats):
        self.calls, self.num_repeats = 0, num_repeats

    def __call__(self, rng, rays):
        i = self.calls % self.num_repeats
        self.calls += 1
        d = torch.from_numpy(np.asarray(rays.directions)).reshape(-1, 3).float()
        ang = torch.tensor(0.7 * i)
        nrm = torch.stack([torch.cos(ang + d[:, 0]), torch.sin(ang + d[:, 0]), torch.zeros_like(d[:, 0])], -1)
        return {"render": {"rgb": d * 0.5 + i, "acc": d[:, 0] * 0.25 + 2.0 * i, "distance_median": d[:, 1] + 10.0 * i,
                           "normals_pred": nrm}}


def _repeat_expected(rays, num_repeats):
    d = torch.from_numpy(np.asarray(rays.directions)).float()
    mean_i = (num_repeats - 1) / 2.0
    nrm = torch.stack([torch.cos(d[..., 0]), torch.sin(d[..., 0]), torch.zeros_like(d[..., 0])], -1)
    return {"rgb": d * 0.5 + mean_i, "acc": d[..., 0] * 0.25 + 2.0 * mean_i, "distance_median": d[..., 1], "normals_pred": nrm}


def _worker_repeats(rank, world, port, H, W, q):
    sys.path.insert(0, ROOT)
    import nrc_amd
    from nrc_amd import model as M
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rays = nrc_amd.synthetic_camera_rays(H, W)
    out = M.render_image_distributed(_RepeatApply(3), 7, rays, _Cfg(), keys=("rgb", "acc", "distance_median", "normals_pred"),
                                     num_repeats=3)
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_repeats_average_stat_keys_only_like_render_image():
    import nrc_amd
    from nrc_amd import model as M
    H, W = 5, 9
    rays = nrc_amd.synthetic_camera_rays(H, W)
    want = _repeat_expected(rays, 3)
    keys = ("rgb", "acc", "distance_median", "normals_pred")
    # one process, no process group
    single = M.render_image_distributed(_RepeatApply(3), 7, rays, _Cfg(), keys=keys, num_repeats=3)
    for k in keys:
        assert np.allclose(single[k].numpy(), want[k].numpy(), atol=1e-6), k
    assert np.allclose(np.linalg.norm(single["normals_pred"].numpy(), axis=-1), 1.0, atol=1e-6)
    # the host loop of render_image on the same callable (the contract both follow)
    ra = _RepeatApply(3)

    def render_fn(rng, chunk_rays, passes, resample):
        flat = chunk_rays.tree_map(lambda r: np.asarray(r).reshape((-1,) + np.asarray(r).shape[2:]))
        out = ra(rng, flat)["render"]
        return {k: v[None, None] for k, v in out.items()}, rng

    img, _ = M.render_image(render_fn, None, rays, _Cfg(), ("cache",), verbose=False, num_repeats=3)
    for k in keys:
        assert np.allclose(img[k], single[k].numpy(), atol=1e-6), k
    # two gloo ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_repeats, args=(r, 2, port, H, W, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=500)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for k in keys:
        assert np.allclose(got[k], want[k].numpy(), atol=1e-6), k
