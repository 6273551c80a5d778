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
