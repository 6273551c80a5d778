"""The power of the ray-distance ladder (csrc/rc_dev_sample.h pow_pos): x^c = 2^(c log2 x) with the exponent of x split off
and the product c * e carried with its rounding residual.  The GPU test (test_gpu_parity.py::test_power_ladder_mapping_...)
checks the kernel; this is the same arithmetic step by step in numpy float32 -- with correctly rounded log2 / exp2 where the
hardware's are within 1 ulp -- against float64: the ALGORITHM's own error stays below 1 ulp over the whole domain the
ladder uses (x = 1 - 0.6 y in (6e-8, 1], c = 1 / p = -2/3), so the kernel's is the hardware's log2 / exp2 on top."""
import numpy as np


def pow_pos_f32(x, c):
    f = np.float32
    m, e = np.frexp(x.astype(np.float32))            # x = m 2^e, m in [0.5, 1)
    low = m < f(0.70710678)
    m = np.where(low, m + m, m).astype(np.float32)
    e = np.where(low, e - 1, e)
    l = np.log2(m.astype(np.float64)).astype(np.float32)
    fe = e.astype(np.float32)
    ce = (f(c) * fe).astype(np.float32)
    r = (np.float64(f(c)) * fe.astype(np.float64) - ce.astype(np.float64)).astype(np.float32)     # fma(c, fe, -ce): exact
    rest = (np.float64(f(c)) * l.astype(np.float64) + r.astype(np.float64)).astype(np.float32)    # fma(c, l, r)
    n = np.rint(ce)
    fr = ((ce - n).astype(np.float32) + rest).astype(np.float32)
    return np.ldexp(np.exp2(fr.astype(np.float64)).astype(np.float32), n.astype(np.int32))


def test_pow_pos_algorithm_error_is_below_one_ulp():
    rng = np.random.default_rng(3)
    c = np.float32(1.0) / np.float32(-1.5)
    y = rng.uniform(0.0, 1.6666666, 200000).astype(np.float32)
    x = np.concatenate([(np.float32(-0.6) * y + np.float32(1.0)).astype(np.float32),
                        np.exp2(rng.uniform(-24.0, 0.0, 100000)).astype(np.float32), np.float32([1.0, 0.5, 0.25, 6e-8])])
    x = x[x > 0]
    got = pow_pos_f32(x, c).astype(np.float64)
    want = np.power(x.astype(np.float64), np.float64(c))
    ulp = np.spacing(want.astype(np.float32)).astype(np.float64)
    err = np.abs(got - want) / ulp
    assert err.max() < 1.0, float(err.max())
    assert np.array_equal(pow_pos_f32(np.float32([1.0]), c), np.float32([1.0]))
