"""Reader (and writer) for the checkpoints the reference saves.

The Trainer writes `flax.training.checkpoints.save_checkpoint_multiprocess(checkpoint_dir, state, step)`
(engine/trainer.py:2054-2066): one file `checkpoint_<step>` holding `flax.serialization.msgpack_serialize`
of the TrainState, i.e. msgpack with three extension types (1: ndarray as (shape, dtype name, raw bytes);
2: Python complex; 3: numpy scalar) and arrays above 2^30 bytes split into
{"__msgpack_chunked_array__": True, "shape": {...}, "chunks": {"0": ..., "1": ...}}.  `state.params` is the
variable dict {"params": {"Cache": {...}, ...}}, which flattened with "/" gives exactly the names
`rc_load_weights` takes ("params/Cache/Sampler/MLP_0/density_grid/hash_0128", ...).

Neither flax nor jax is needed: msgpack + numpy.  Restoring by prefix mirrors
train_utils.restore_partial_checkpoint (internal/train_utils.py:4035-4088).
"""
from __future__ import annotations

import os
import re
from typing import Any, Dict, Iterable, Optional

import msgpack
import numpy as np

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
_MAX_CHUNK_BYTES = 2 ** 30


def _ndarray_from_bytes(data: bytes) -> np.ndarray:
    shape, dtype_name, buffer = msgpack.unpackb(data, raw=True)
    name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
    if name == "bfloat16":
        # stored as 2-byte words: widen to float32 (upper half of the word)
        raw = np.frombuffer(buffer, dtype=np.uint16).astype(np.uint32) << 16
        return raw.view(np.float32).reshape(tuple(shape))
    return np.frombuffer(buffer, dtype=np.dtype(name), count=-1).reshape(tuple(shape))


def _ext_hook(code: int, data: bytes):
    if code == _EXT_NDARRAY:
        return _ndarray_from_bytes(data)
    if code == _EXT_NPSCALAR:
        return _ndarray_from_bytes(data)[()]
    if code == _EXT_COMPLEX:
        re_, im_ = msgpack.unpackb(data)
        return complex(re_, im_)
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get("__msgpack_chunked_array__"):
            shape = tuple(tree["shape"][str(i)] for i in range(len(tree["shape"])))
            chunks = [tree["chunks"][str(i)] for i in range(len(tree["chunks"]))]
            return np.concatenate([np.asarray(c).reshape(-1) for c in chunks]).reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def read_flax_msgpack(src) -> Dict[str, Any]:
    """Path, bytes or file object -> the nested state dict (numpy leaves)."""
    if isinstance(src, (bytes, bytearray, memoryview)):
        data = bytes(src)
    elif hasattr(src, "read"):
        data = src.read()
    else:
        with open(src, "rb") as f:
            data = f.read()
    tree = msgpack.unpackb(data, ext_hook=_ext_hook, raw=False, strict_map_key=False)
    return _unchunk(tree)


def latest_checkpoint(checkpoint_dir: str, prefix: str = "checkpoint_") -> Optional[str]:
    """flax.training.checkpoints.latest_checkpoint: natural order of the step suffix."""
    if not os.path.isdir(checkpoint_dir):
        return None
    best, best_key = None, None
    for name in os.listdir(checkpoint_dir):
        if not name.startswith(prefix) or name.endswith(".tmp") or "tmp" in name[len(prefix):]:
            continue
        key = [float(t) if re.fullmatch(r"\d+(\.\d+)?", t) else t for t in re.split(r"(\d+(?:\.\d+)?)", name[len(prefix):]) if t]
        try:
            if best_key is None or key > best_key:
                best, best_key = name, key
        except TypeError:
            continue
    return None if best is None else os.path.join(checkpoint_dir, best)


def flatten(tree: Dict[str, Any], prefix: str = "") -> Dict[str, np.ndarray]:
    """flax.traverse_util.flatten_dict(tree, sep='/')."""
    out: Dict[str, np.ndarray] = {}
    for k, v in tree.items():
        name = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten(v, name))
        else:
            out[name] = v
    return out


def load_params(path: str, prefixes: Optional[Iterable[str]] = None, exclude_prefixes: Optional[Iterable[str]] = None,
                dtype=np.float32) -> Dict[str, np.ndarray]:
    """`checkpoint_<step>` file (or a directory: its latest checkpoint) -> {"params/...": array}, optionally restricted
    the way restore_partial_checkpoint restricts it (prefixes on the flattened names, e.g. "params/Cache")."""
    if os.path.isdir(path):
        found = latest_checkpoint(path)
        if found is None:
            raise FileNotFoundError(f"no checkpoint_* file in {path}")
        path = found
    state = read_flax_msgpack(path)
    params = state["params"] if "params" in state and isinstance(state["params"], dict) else state
    flat = flatten(params)
    if not any(k.startswith("params/") for k in flat):       # a bare {"Cache": ...} tree
        flat = {"params/" + k: v for k, v in flat.items()}
    pre = None if prefixes is None else tuple(prefixes)
    exc = None if exclude_prefixes is None else tuple(exclude_prefixes)
    keep = {}
    for k, v in flat.items():
        if pre is not None and not k.startswith(pre):
            continue
        if exc is not None and k.startswith(exc):
            continue
        a = np.asarray(v)
        keep[k] = np.ascontiguousarray(a, dtype=dtype) if a.dtype.kind == "f" else a
    return keep


# ------------------------------------------------------------------------------------------------
# writer (same container; used to export synthetic weights and by the tests)
# ------------------------------------------------------------------------------------------------
def _ndarray_to_bytes(arr: np.ndarray) -> bytes:
    return msgpack.packb((arr.shape, arr.dtype.name, arr.tobytes("C")), use_bin_type=True)


def _ext_pack(x):
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _ndarray_to_bytes(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _ndarray_to_bytes(np.asarray(x)))
    if isinstance(x, complex):
        return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((x.real, x.imag)))
    return x


def _chunk_leaves(tree, max_chunk_bytes):
    if isinstance(tree, dict):
        return {k: _chunk_leaves(v, max_chunk_bytes) for k, v in tree.items()}
    if isinstance(tree, np.ndarray) and tree.size * tree.dtype.itemsize > max_chunk_bytes:
        n = max(1, int(max_chunk_bytes / tree.dtype.itemsize))
        flat = tree.reshape(-1)
        chunks = [flat[i:i + n] for i in range(0, flat.size, n)]
        return {"__msgpack_chunked_array__": True, "shape": {str(i): int(s) for i, s in enumerate(tree.shape)},
                "chunks": {str(i): c for i, c in enumerate(chunks)}}
    return tree


def unflatten(flat: Dict[str, np.ndarray]) -> Dict[str, Any]:
    tree: Dict[str, Any] = {}
    for name, v in flat.items():
        node = tree
        parts = name.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return tree


def write_flax_msgpack(state: Dict[str, Any], path: str, max_chunk_bytes: int = _MAX_CHUNK_BYTES) -> None:
    data = msgpack.packb(_chunk_leaves(state, max_chunk_bytes), default=_ext_pack, strict_types=True)
    with open(path, "wb") as f:
        f.write(data)


def save_params(flat_params: Dict[str, np.ndarray], checkpoint_dir: str, step: int = 0) -> str:
    """{"params/...": array} -> <dir>/checkpoint_<step> shaped like a TrainState ({"step", "params": {"params": ...}})."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    tree = unflatten({k: np.asarray(v) for k, v in flat_params.items()})
    path = os.path.join(checkpoint_dir, f"checkpoint_{step}")
    write_flax_msgpack({"step": int(step), "params": tree}, path)
    return path
