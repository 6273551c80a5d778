"""`make hostcheck`: the host-only parts of the weight packing (csrc/rc_pack_host.h -- MFMA A-fragment packing, the
folded shader bottleneck, the cell-table index rule, the directional-encoding table) compiled for the CPU under
AddressSanitizer + UBSan and compared with numpy restatements of the layouts (sanitizers on the CPU build;
GPU sanitizers are not available on the pool).  A sanitizer report makes the program exit non-zero."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neural-radiance-caching_amd", "csrc")


def acc_feat(t, r, h):
    """Feature that accumulator register r of tile t holds on half-wave h (v_mfma_f32_32x32x2_f32 D layout)."""
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h


def np_pack(steps, tiles, layers):
    """steps: [(row_h0, row_h1)], >= 0 input row, -1 zero, -2 bias.  tiles: [[(layer, col, bias_ok) or None] * 32]."""
    out = np.zeros((len(steps), len(tiles), 64), np.float32)
    for s, rows in enumerate(steps):
        for t, tile in enumerate(tiles):
            for lane in range(64):
                h, i = lane >> 5, lane & 31
                ent = tile[i]
                if ent is None:
                    continue
                name, col, bias_ok = ent
                k, b = layers[name]
                row = rows[h]
                if row == -2:
                    out[s, t, lane] = b[col] if bias_ok else 0.0
                elif row >= 0 and row < k.shape[0]:
                    out[s, t, lane] = k[row, col]
    return out.reshape(-1)


def unsplit(words, n_steps, n_tiles):
    """The split form of a layer (rc_pack_host.h pack_split: [block of 8 k-steps][tile][piece][lane][4 dwords], a dword = the
    bf16 of an even step in its low half and of the following step in its high half) back to [step][tile][lane] float32:
    the three pieces must add up to the packed weight EXACTLY, each must be the truncated top 16 bits of what the pieces
    in front of it leave, and the steps that pad the last block must be zero."""
    nb = (n_steps + 7) // 8
    w = np.asarray(words).view(np.uint32).reshape(nb, n_tiles, 3, 64, 4)
    halves = np.stack([w & 0xFFFF, w >> 16], axis=-1).reshape(nb, n_tiles, 3, 64, 8)          # [..., j] = step 8 q + j
    pieces = (halves.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    total = pieces.sum(axis=2)                                                                 # exact in float64
    val = total.astype(np.float32)
    assert np.array_equal(val.astype(np.float64), total)
    rest = val.astype(np.float32)
    for p in range(3):
        top = (rest.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        assert np.array_equal(top.astype(np.float64), pieces[:, :, p])
        rest = (rest - top).astype(np.float32)
    assert not rest.any()
    out = val.transpose(0, 3, 1, 2).reshape(nb * 8, n_tiles, 64)                                # [step][tile][lane]
    assert not out[n_steps:].any()
    return out[:n_steps].reshape(-1)


def natural(K):
    return [(2 * i, 2 * i + 1 if 2 * i + 1 < K else -1) for i in range((K + 1) // 2)]


def full_tile(name, t, out_dim, bias_ok=True):
    return [(name, 32 * t + i, bias_ok) if 32 * t + i < out_dim else None for i in range(32)]


@pytest.fixture(scope="module")
def hostcheck(tmp_path_factory):
    r = subprocess.run(["make", "-C", CSRC, "hostcheck"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    d = tmp_path_factory.mktemp("hostcheck")
    rng = np.random.default_rng(11)
    layers = {}
    for name, (i, o) in dict(a=(37, 70), b=(64, 40), c=(9, 5), d=(64, 3), bott=(12, 16), cons=(21, 9)).items():
        k = rng.normal(size=(i, o)).astype(np.float32)
        b = rng.normal(size=(o,)).astype(np.float32)
        k.tofile(d / f"in_{name}_kernel.bin")
        b.tofile(d / f"in_{name}_bias.bin")
        layers[name] = (k, b)
    grid = rng.normal(size=(5, 5, 5, 2)).astype(np.float32)          # [z, y, x, F] = entry ((k2-1) N + (k1-1)) N + (k0-1)
    grid.tofile(d / "in_grid.bin")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(CSRC, "hostcheck"), str(d)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "hostcheck ok" in r.stdout, r.stdout + r.stderr       # non-zero: a sanitizer report
    return d, layers, grid


def _out(d, name):
    return np.fromfile(d / f"out_{name}.bin", dtype=np.float32)


def test_fragment_packing_matches_numpy(hostcheck):
    d, layers, _ = hostcheck
    want = np_pack(natural(37) + [(-2, -1)], [full_tile("a", t, 70) for t in range(3)], layers)
    assert np.array_equal(unsplit(_out(d, "pack"), 20, 3), want)
    acc_steps = [(acc_feat(t, r, 0), acc_feat(t, r, 1)) for t in range(2) for r in range(16)]
    want = np_pack(acc_steps, [full_tile("b", t, 40, False) for t in range(2)], layers)
    assert np.array_equal(unsplit(_out(d, "pack_acc"), 32, 2), want)
    # "by register": output r sits in accumulator register r of BOTH half-waves = tile rows (r & 3) + 8 (r >> 2) + 4 h
    tile = [None] * 32
    for i in range(32):
        r = (i & 3) + 4 * (i >> 3)
        if r < 5:
            tile[i] = ("c", r, True)
    want = np_pack(natural(9) + [(-2, -1)], [tile], layers)
    assert np.array_equal(unsplit(_out(d, "by_reg"), 6, 1), want)


def test_dot_fragments_match_numpy(hostcheck):
    d, layers, _ = hostcheck
    k, b = layers["d"]
    want = []
    for o in range(3):
        for t in range(2):
            for r in range(16):
                want.append([k[acc_feat(t, r, lane >> 5), o] for lane in range(64)])
    for o in range(3):
        want.append([b[o]] * 64)
    want = np.asarray(want, np.float32).reshape(-1)
    got = _out(d, "dot")                     # padded with zero fragments to whole 1-KiB pieces (rc_dfr)
    assert got.size == (3 * 33 + 3) // 4 * 4 * 64 and np.array_equal(got[:want.size], want) and not got[want.size:].any()


def test_folded_bottleneck_matches_fp64_product(hostcheck):
    d, layers, _ = hostcheck
    (kb, bb), (kc, _) = layers["bott"], layers["cons"]
    wk = np.concatenate([(kb.astype(np.float64) @ kc[:16].astype(np.float64)).astype(np.float32), kc[16:21]])
    wb = (bb.astype(np.float64) @ kc[:16].astype(np.float64)).astype(np.float32)
    got_k, got_b = _out(d, "fold_kernel").reshape(17, 9), _out(d, "fold_bias")
    # the C++ sums its 16 products in index order in fp64; numpy's dot may pair them differently: one float32 ulp
    assert np.allclose(got_k, wk, rtol=0, atol=1e-6) and np.array_equal(got_k[12:], kc[16:21])
    assert np.allclose(got_b, wb, rtol=0, atol=1e-6)


def test_cell_table_matches_padded_volume(hostcheck):
    d, _, grid = hostcheck
    N, F, M = 5, 2, 8
    pad = np.zeros((N + 2, N + 2, N + 2, F), np.float32)             # [k2, k1, k0]: zero padding at 0 and N + 1
    pad[1:-1, 1:-1, 1:-1] = grid
    want = np.zeros((M, M, M, 8, F), np.float32)                     # [q2, q1, q0, corner]
    for q2 in range(M):
        for q1 in range(M):
            for q0 in range(M):
                for c in range(8):
                    b0, b1, b2 = (c >> 2) & 1, (c >> 1) & 1, c & 1
                    k0, k1, k2 = (min(max(q - 1 + b, 0), N + 1) for q, b in ((q0, b0), (q1, b1), (q2, b2)))
                    want[q2, q1, q0, c] = pad[k2, k1, k0]
    assert np.array_equal(_out(d, "cells"), want.reshape(-1))


def test_ide_table_matches_the_oracle(hostcheck):
    import sys
    sys.path.insert(0, ROOT)
    from oracle import mathx
    d, _, _ = hostcheck
    v = _out(d, "ide")
    coef = v[: 36 * 17].reshape(36, 17)
    m = v[36 * 17: 36 * 17 + 36]
    sigma = v[36 * 17 + 36:]
    ml, mat = mathx.ide_tables(5)                                    # ml [2, 36] (m, l), mat [k, term]
    ml, mat = np.asarray(ml), np.asarray(mat, np.float64)
    assert np.array_equal(m, ml[0].astype(np.float32))
    l = ml[1].astype(np.float64)
    assert np.allclose(sigma, 0.5 * l * (l + 1))
    assert np.allclose(coef[:, : mat.shape[0]], mat.T.astype(np.float32), rtol=1e-6, atol=0)
    assert not coef[:, mat.shape[0]:].any()
