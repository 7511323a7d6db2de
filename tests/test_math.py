"""include/mort_math.h against glibc/numpy: every fp32 function must be within 1 ULP of the
correctly rounded result (computed in float64 and rounded once), specials must follow IEEE."""
import math

import numpy as np
import pytest


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def vec(fn, xs):
    return np.array([fn(float(x)) for x in xs], dtype=np.float32)


def test_sinf_cosf(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(0, 2 * math.pi, 20000), rng.uniform(-50, 50, 5000), [0.0, math.pi, 2 * math.pi]]).astype(np.float32)
    for name, ref in (("sinf", np.sin), ("cosf", np.cos)):
        got = vec(getattr(L, "mort_oracle_" + name), xs)
        want = ref(xs.astype(np.float64)).astype(np.float32)
        # near zeros of the function the relative error of the fp64 reference itself matters; use abs there
        ok = (ulp_diff(got, want) <= 1) | (np.abs(got.astype(np.float64) - ref(xs.astype(np.float64))) < 1e-9)
        assert ok.all(), (name, xs[~ok][:5], got[~ok][:5], want[~ok][:5])


def test_sincosf_is_sinf_and_cosf(oracle):
    """The kernels evaluate sin and cos of random_cosine_direction's angle together (mort_sincosf): bit-identical
    to the separate functions, including out-of-domain, inf and NaN arguments."""
    import ctypes as C
    L = oracle.lib()
    L.mort_oracle_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mort_oracle_sincosf.restype = None
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(0, 2 * math.pi, 20000), rng.uniform(-1000, 1000, 5000),
                         [0.0, -0.0, math.pi, 2 * math.pi, 1e5, -1e5, 99999.99, 3e38, np.inf, -np.inf, np.nan]]).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    for x in xs:
        L.mort_oracle_sincosf(float(x), C.byref(s), C.byref(c))
        want_s = np.float32(L.mort_oracle_sinf(float(x))); want_c = np.float32(L.mort_oracle_cosf(float(x)))
        assert np.float32(s.value).view(np.uint32) == want_s.view(np.uint32) or (np.isnan(s.value) and np.isnan(want_s)), x
        assert np.float32(c.value).view(np.uint32) == want_c.view(np.uint32) or (np.isnan(c.value) and np.isnan(want_c)), x


def test_sin_f64(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(2)
    xs = rng.uniform(-500, 500, 20000)
    got = np.array([L.mort_oracle_sin(float(x)) for x in xs])
    assert np.max(np.abs(got - np.sin(xs))) < 5e-14


def test_acosf(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-1, 1, 20000), [-1.0, 1.0, 0.0, -0.99999994, 0.99999994]]).astype(np.float32)
    got = vec(L.mort_oracle_acosf, xs)
    want = np.arccos(xs.astype(np.float64)).astype(np.float32)
    assert (ulp_diff(got, want) <= 1).all()
    assert math.isnan(L.mort_oracle_acosf(1.0000001)) and math.isnan(L.mort_oracle_acosf(-1.5))


def test_atan2f(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(4)
    ys = rng.uniform(-2, 2, 20000).astype(np.float32)
    xs = rng.uniform(-2, 2, 20000).astype(np.float32)
    got = np.array([L.mort_oracle_atan2f(float(y), float(x)) for y, x in zip(ys, xs)], dtype=np.float32)
    want = np.arctan2(ys.astype(np.float64), xs.astype(np.float64)).astype(np.float32)
    assert (ulp_diff(got, want) <= 1).all()
    for y, x in ((0.0, 1.0), (0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (-0.0, -1.0), (1.0, 1.0), (-1.0, -1.0)):
        assert L.mort_oracle_atan2f(y, x) == pytest.approx(math.atan2(y, x), abs=1e-7)


def test_logf(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(0, 1, 20000), rng.uniform(0, 1e-30, 100), [1.0, 0.5, 5.9604645e-08]]).astype(np.float32)
    xs = xs[xs > 0]
    got = vec(L.mort_oracle_logf, xs)
    want = np.log(xs.astype(np.float64)).astype(np.float32)
    assert (ulp_diff(got, want) <= 1).all()
    assert L.mort_oracle_logf(0.0) == -math.inf
    assert math.isnan(L.mort_oracle_logf(-1.0))
