"""Assemble a `checkpoint_<step>` file the way flax.training.checkpoints writes one, with the `msgpack` package ONLY
(nothing of nrc_amd.checkpoint is imported): the independent fixture of SURVEY 8(f) rank 2 / VERDICT r2 item 6.

Layout written (flax.serialization.msgpack_serialize of the Trainer's TrainState, engine/trainer.py:2054-2066):

    {"step": int, "params": {"params": {"Cache": {"Sampler": {"MLP_0": {...}}, ...}}, "opt_state": {...}}

    * every ndarray leaf: msgpack ExtType(1, packb((shape, dtype.name, raw C-order bytes)))
    * numpy scalars: ExtType(3, same triple); Python complex: ExtType(2, packb((re, im)))
    * an array above `max_chunk_bytes` (flax: 2^30): {"__msgpack_chunked_array__": True,
      "shape": {"0": d0, "1": d1, ...}, "chunks": {"0": ext(flat[0:c]), "1": ext(flat[c:2c]), ...}}

No flax in the image: the layout is written from flax's documented format; parameter NAMES are the ones SURVEY 8(a')
infers from the reference's module attributes (internal/geometry.py:123-153, grid_utils.py:851-852, nerf.py:232-346,
surface_light_field.py:343-403) and stay unverified against a real checkpoint.

    python tests/golden/make_flax_checkpoint.py <out dir> [step]     # the full hotdog inventory, synthetic weights (~330 MB)
"""
import os
import sys

import msgpack
import numpy as np


def _ext(arr, code=1):
    arr = np.ascontiguousarray(arr)
    return msgpack.ExtType(code, msgpack.packb((arr.shape, arr.dtype.name, arr.tobytes("C")), use_bin_type=True))


def _leaf(arr, max_chunk_bytes):
    arr = np.asarray(arr)
    if arr.nbytes <= max_chunk_bytes:
        return _ext(arr)
    per = max(1, max_chunk_bytes // arr.dtype.itemsize)
    flat = arr.reshape(-1)
    return {"__msgpack_chunked_array__": True,
            "shape": {str(i): int(d) for i, d in enumerate(arr.shape)},
            "chunks": {str(i): _ext(flat[o: o + per]) for i, o in enumerate(range(0, flat.size, per))}}


def nest(flat, max_chunk_bytes):
    """{"params/Cache/.../kernel": ndarray} -> nested dict with ext-typed leaves."""
    tree = {}
    for name, arr in flat.items():
        node = tree
        parts = name.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _leaf(arr, max_chunk_bytes)
    return tree


def assemble(flat_params, checkpoint_dir, step, max_chunk_bytes=2 ** 30):
    """Write <checkpoint_dir>/checkpoint_<step>; `flat_params` uses the names rc_load_weights takes ("params/...").
    Returns the path."""
    inner = {k[len("params/"):]: v for k, v in flat_params.items()}
    assert len(inner) == len(flat_params) and all(k.startswith("params/") for k in flat_params)
    state = {"step": int(step),
             "params": {"params": nest(inner, max_chunk_bytes)},
             # what an optax state adds next to the parameters: must be skipped by a parameter restore
             "opt_state": {"0": {"count": _ext(np.asarray(np.int32(step)), 3)}, "lr": msgpack.ExtType(2, msgpack.packb((1e-3, 0.0)))}}
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = os.path.join(checkpoint_dir, f"checkpoint_{int(step)}")
    with open(path, "wb") as f:
        f.write(msgpack.packb(state, strict_types=True))
    return path


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, root)
    import nrc_amd
    cfg = nrc_amd.hotdog_config()
    w = nrc_amd.synthetic_weights(cfg, passes=("cache", "material"))
    p = assemble(w, sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25000, max_chunk_bytes=4 << 20)
    print(p, os.path.getsize(p), "bytes,", len(w), "tensors")
