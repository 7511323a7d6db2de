"""XORWOW (cuRAND's generator, restated in oracle/mort_oracle.c): recurrence and 2^67 sequence
skip cross-checked against rocRAND's independent engine; uniform mapping bounds; seeding layout.

Not pinned here (nothing in the container can pin it): cuRAND's seed-scramble constants."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from mort_amd import structs as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def py_xorwow(d, v, n):
    v = list(v)
    out = []
    for _ in range(n):
        t = (v[0] ^ (v[0] >> 2)) & 0xFFFFFFFF
        v[0], v[1], v[2], v[3] = v[1], v[2], v[3], v[4]
        v[4] = ((v[4] ^ ((v[4] << 4) & 0xFFFFFFFF)) ^ (t ^ ((t << 1) & 0xFFFFFFFF))) & 0xFFFFFFFF
        d = (d + 362437) & 0xFFFFFFFF
        out.append((v[4] + d) & 0xFFFFFFFF)
    return d, v, out


def test_recurrence_matches_pure_python(oracle):
    L = oracle.lib()
    st = S.RngState()
    L.mort_oracle_rng_init(st, 69420, 0)
    d0, v0 = st.d, list(st.v)
    got = [L.mort_oracle_rng_next(st) for _ in range(64)]
    d1, v1, want = py_xorwow(d0, v0, 64)
    assert got == want and st.d == d1 and list(st.v) == v1


def test_seed_scramble_formula(oracle):
    """curand_init's published scramble, recomputed here in Python integers."""
    L = oracle.lib()
    for seed in (0, 1, 69420, 0x123456789ABCDEF):
        st = S.RngState()
        L.mort_oracle_rng_init(st, seed, 0)
        s0 = (seed & 0xFFFFFFFF) ^ 0xAAD26B49
        s1 = (seed >> 32) ^ 0xF7DCEFDD
        t0 = (1099087573 * s0) & 0xFFFFFFFF
        t1 = (2591861531 * s1) & 0xFFFFFFFF
        assert st.d == (6615241 + t1 + t0) & 0xFFFFFFFF
        assert list(st.v) == [(123456789 + t0) & 0xFFFFFFFF, 362436069 ^ t0, (521288629 + t1) & 0xFFFFFFFF,
                              88675123 ^ t1, (5783321 + t0) & 0xFFFFFFFF]


@pytest.fixture(scope="module")
def rocrand_tool(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rocrand/rocrand_xorwow.h"):
        pytest.skip("hipcc / rocRAND headers not available")
    exe = str(tmp_path_factory.mktemp("rr") / "rocrand_check")
    subprocess.check_call([hipcc, "-O1", "--offload-arch=gfx950", "-o", exe,
                           os.path.join(ROOT, "tests", "tools", "rocrand_xorwow_check.cpp")])
    return exe


def test_sequence_skip_matches_rocrand(oracle, rocrand_tool):
    """state(seed, subsequence k) == rocRAND's discard_subsequence(k) applied to state(seed, 0),
    and the following outputs agree (host code of the rocRAND header; no GPU involved)."""
    L = oracle.lib()
    subs = [0, 1, 2, 3, 4, 5, 63, 64, 1199, 1200, 809999, 16777215, 2**24 + 12345, 2**40 + 7]
    base = S.RngState()
    L.mort_oracle_rng_init(base, 69420, 0)
    lines = "".join(f"{base.d} {base.v[0]} {base.v[1]} {base.v[2]} {base.v[3]} {base.v[4]} {k} 5\n" for k in subs)
    out = subprocess.run([rocrand_tool], input=lines, capture_output=True, text=True, check=True).stdout.split("\n")
    for k, line in zip(subs, out):
        st = S.RngState()
        L.mort_oracle_rng_init(st, 69420, k)
        last = 0
        for _ in range(5):
            last = L.mort_oracle_rng_next(st)
        want = [int(t) for t in line.split()]
        assert [st.d] + list(st.v) + [last] == want, k


def test_seed_array_layout(oracle):
    """states[x + y*W] = init(seed, x + y*W): the sequential fill equals per-index initialisation."""
    L = oracle.lib()
    W, H = 37, 11
    st = oracle.seed_states(69420, W, H)
    for idx in (0, 1, 36, 37, 200, W * H - 1):
        one = S.RngState()
        L.mort_oracle_rng_init(one, 69420, idx)
        assert st["d"][idx] == one.d and list(st["v"][idx]) == list(one.v)
    assert (st["bf"] == 0).all() and (st["bed"] == 0).all()


def test_uniform_and_random_float_ranges(oracle):
    L = oracle.lib()
    st = S.RngState()
    L.mort_oracle_rng_init(st, 1, 0)
    us = np.array([L.mort_oracle_rng_uniform(st) for _ in range(20000)], dtype=np.float32)
    assert us.min() > 0.0 and us.max() <= 1.0
    fs = np.array([L.mort_oracle_random_float(st) for _ in range(20000)], dtype=np.float32)
    assert fs.min() >= 0.0 and fs.max() <= 1.0
    ints = [L.mort_oracle_random_int(st, 0, 1) for _ in range(2000)]
    assert set(ints) == {0, 1}
    # extreme outputs: x = 0xffffffff -> uniform 1.0 -> random_float 0.0; x = 0 -> 2^-33 -> random_float rounds to 1.0
    assert np.float32(np.float32(4294967295) * np.float32(2.3283064e-10) + np.float32(2.3283064e-10) / np.float32(2)) == np.float32(1.0)
    assert np.float32(1.0 - float(np.float32(2.3283064e-10) / np.float32(2))) == np.float32(1.0)


def test_random_float_single_rounding():
    """dev_math.h computes rng.cuh:17-23's (float)(1.0 - (double)u) as the fp32 subtraction 1.0f - u.

    Identical for every generator output (a C loop over all 2^32 agrees); here: both ends of the range,
    every output that lands near a binade boundary of u, and a stride over the rest.
    """
    edges = np.concatenate([np.arange(0, 1 << 21, dtype=np.uint64), np.arange((1 << 32) - (1 << 21), 1 << 32, dtype=np.uint64)])
    pows = np.concatenate([np.arange(max(0, (1 << k) - 4096), (1 << k) + 4096, dtype=np.uint64) for k in range(8, 32)])
    stride = np.arange(0, 1 << 32, 257, dtype=np.uint64)
    for xs in (edges, pows, stride):
        x = xs.astype(np.uint32)
        u = x.astype(np.float32) * np.float32(2.3283064e-10) + np.float32(2.3283064e-10) / np.float32(2.0)
        assert u.dtype == np.float32
        via_f64 = (1.0 - u.astype(np.float64)).astype(np.float32)
        direct = np.float32(1.0) - u
        assert np.array_equal(via_f64.view(np.uint32), direct.view(np.uint32))
