"""Flax msgpack checkpoint container.  No flax / jax in the image: the byte layout is
checked against a hand-assembled file that spells out flax.serialization's encoding."""
import os

import msgpack
import numpy as np
import pytest

import nrc_amd
from nrc_amd import checkpoint as ck


def test_reads_hand_assembled_flax_bytes(tmp_path):
    """ExtType 1 = msgpack((shape, dtype.name, bytes)); numpy scalar = ExtType 3; chunked arrays as a dict."""
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    ext = lambda arr, code=1: msgpack.ExtType(code, msgpack.packb((arr.shape, arr.dtype.name, arr.tobytes("C")), use_bin_type=True))
    big = np.arange(10, dtype=np.float32)
    tree = {"step": 1234,
            "params": {"params": {"Cache": {"Shader": {"tint_layer": {"kernel": ext(a), "bias": ext(np.ones(3, np.float32))}}},
                                  "big": {"__msgpack_chunked_array__": True, "shape": {"0": 2, "1": 5},
                                          "chunks": {"0": ext(big[:4]), "1": ext(big[4:8]), "2": ext(big[8:])}}}},
            "opt_state": {"count": ext(np.asarray(np.int32(7)), 3), "z": msgpack.ExtType(2, msgpack.packb((1.0, -2.0)))}}
    p = tmp_path / "checkpoint_1234"
    p.write_bytes(msgpack.packb(tree, strict_types=True))
    st = ck.read_flax_msgpack(str(p))
    assert st["step"] == 1234 and st["opt_state"]["count"] == 7 and st["opt_state"]["count"].dtype == np.int32
    assert st["opt_state"]["z"] == complex(1, -2)
    k = st["params"]["params"]["Cache"]["Shader"]["tint_layer"]["kernel"]
    assert k.dtype == np.float32 and np.array_equal(k, a)
    assert np.array_equal(st["params"]["params"]["big"], big.reshape(2, 5))
    flat = ck.load_params(str(p), prefixes=["params/Cache"])
    assert sorted(flat) == ["params/Cache/Shader/tint_layer/bias", "params/Cache/Shader/tint_layer/kernel"]


def test_round_trip_of_the_cache_inventory_and_prefix_restore(tmp_path):
    cfg = nrc_amd.hotdog_config()
    shapes = nrc_amd.param_shapes(cfg, passes=("cache", "material"))
    rng = np.random.default_rng(0)
    # small stand-ins with the real names (the real tables are 100+ MB): one value per tensor, broadcast on load
    w = {k: rng.normal(size=(min(s[0], 4),) + tuple(s[1:][-1:])).astype(np.float32) for k, s in shapes.items()}
    path = ck.save_params(w, str(tmp_path), step=25000)
    ck.save_params({"params/x": np.zeros(1, np.float32)}, str(tmp_path), step=5000)
    assert os.path.basename(ck.latest_checkpoint(str(tmp_path))) == "checkpoint_25000" and path.endswith("checkpoint_25000")
    back = ck.load_params(str(tmp_path))
    assert sorted(back) == sorted(w) and all(np.array_equal(back[k], w[k]) for k in w)
    cache_only = ck.load_params(str(tmp_path), prefixes=["params/Cache"], exclude_prefixes=["params/Cache/EnvMap"])
    assert cache_only and all(k.startswith("params/Cache") and not k.startswith("params/Cache/EnvMap") for k in cache_only)
    assert any(k.startswith("params/MaterialShader") for k in back) and not any(k.startswith("params/MaterialShader") for k in cache_only)


def test_chunked_write_and_bfloat16_read(tmp_path):
    a = np.arange(1000, dtype=np.float32).reshape(10, 100)
    p = str(tmp_path / "checkpoint_1")
    ck.write_flax_msgpack({"params": {"params": {"t": a}}}, p, max_chunk_bytes=1024)
    raw = msgpack.unpackb(open(p, "rb").read(), ext_hook=lambda c, d: ("ext", c), raw=False, strict_map_key=False)
    assert raw["params"]["params"]["t"]["__msgpack_chunked_array__"] is True and len(raw["params"]["params"]["t"]["chunks"]) == 4
    assert np.array_equal(ck.load_params(p)["params/t"], a)
    # bfloat16 leaves (dtype name "bfloat16", 2-byte words) are widened to float32
    words = (np.array([1.0, -2.5, 3.0], np.float32).view(np.uint32) >> 16).astype(np.uint16)
    ext = msgpack.ExtType(1, msgpack.packb(((3,), "bfloat16", words.tobytes()), use_bin_type=True))
    q = tmp_path / "checkpoint_2"
    q.write_bytes(msgpack.packb({"params": {"params": {"b": ext}}}, strict_types=True))
    assert np.array_equal(ck.load_params(str(q))["params/b"], np.array([1.0, -2.5, 3.0], np.float32))


def test_reads_the_independently_assembled_inventory(tmp_path):
    """tests/golden/make_flax_checkpoint.py writes the Flax container with msgpack alone (no nrc_amd.checkpoint code):
    the full hotdog inventory by NAME, small stand-in tensors, chunking forced by a small threshold."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_flax_checkpoint", os.path.join(os.path.dirname(__file__), "golden", "make_flax_checkpoint.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    cfg = nrc_amd.hotdog_config()
    shapes = nrc_amd.param_shapes(cfg, passes=("cache", "material"))
    rng = np.random.default_rng(5)
    w = {k: rng.normal(size=(min(s[0], 37),) + tuple(s[1:])).astype(np.float32) for k, s in shapes.items()}
    path = mk.assemble(w, str(tmp_path), 31000, max_chunk_bytes=256)
    raw = msgpack.unpackb(open(path, "rb").read(), ext_hook=lambda c, d: ("ext", c), raw=False, strict_map_key=False)
    assert raw["step"] == 31000 and "opt_state" in raw
    some = raw["params"]["params"]["Cache"]["Sampler"]["MLP_2"]["density_layers_0"]["kernel"]
    assert some["__msgpack_chunked_array__"] is True                     # 37 x 64 floats > 256 bytes
    back = ck.load_params(str(tmp_path))
    assert sorted(back) == sorted(w) and all(back[k].dtype == np.float32 and np.array_equal(back[k], w[k]) for k in w)
    part = ck.load_params(str(tmp_path), prefixes=["params/Cache"])
    assert part and set(part) == {k for k in w if k.startswith("params/Cache")}
