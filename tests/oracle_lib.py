"""Loader for the CPU oracle (oracle/libmort_oracle.so).  Test infrastructure:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it."""
import ctypes as C
import os
import subprocess

import numpy as np

from mort_amd import structs as S

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(_ROOT, "oracle", "libmort_oracle.so")
_lib = None


class Hit(C.Structure):
    _fields_ = [("p", S.Vec3), ("normal", S.Vec3), ("mat_idx", C.c_int), ("mat_type", C.c_int),
                ("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("front_face", C.c_bool)]


class Stats(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("rng_draws", C.c_uint64)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")])
        L = C.CDLL(_PATH)
        W, Cam, St = C.POINTER(S.World), C.POINTER(S.Camera), C.POINTER(S.RngState)
        fp = C.POINTER(C.c_float)
        L.mort_oracle_rng_init.argtypes = [St, C.c_uint64, C.c_uint64]
        L.mort_oracle_rng_seed.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        L.mort_oracle_rng_next.argtypes = [St]; L.mort_oracle_rng_next.restype = C.c_uint
        L.mort_oracle_rng_uniform.argtypes = [St]; L.mort_oracle_rng_uniform.restype = C.c_float
        L.mort_oracle_random_float.argtypes = [St]; L.mort_oracle_random_float.restype = C.c_float
        L.mort_oracle_random_int.argtypes = [St, C.c_int, C.c_int]; L.mort_oracle_random_int.restype = C.c_int
        L.mort_oracle_rng_step_matrix_2p67.argtypes = [C.c_void_p]
        L.mort_oracle_render.argtypes = [W, Cam, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.POINTER(Stats)]
        L.mort_oracle_render.restype = C.c_int
        L.mort_oracle_world_hit.argtypes = [W, fp, C.c_float, C.c_float, St, C.POINTER(Hit)]
        L.mort_oracle_world_hit.restype = C.c_bool
        L.mort_oracle_aabb_hit.argtypes = [C.POINTER(S.Aabb), fp, C.c_float, C.c_float]
        L.mort_oracle_aabb_hit.restype = C.c_bool
        L.mort_oracle_get_ray.argtypes = [Cam, C.c_int, C.c_int, C.c_int, C.c_int, St, fp]
        L.mort_oracle_ray_color.argtypes = [W, Cam, fp, St, fp]
        L.mort_oracle_texture_value.argtypes = [W, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp]
        L.mort_oracle_pdf_value.argtypes = [W, C.c_int, C.c_int, fp, fp]; L.mort_oracle_pdf_value.restype = C.c_float
        L.mort_oracle_light_random.argtypes = [W, C.c_int, C.c_int, fp, St, fp]
        for n in ("sinf", "cosf", "acosf", "logf"):
            f = getattr(L, "mort_oracle_" + n); f.argtypes = [C.c_float]; f.restype = C.c_float
        L.mort_oracle_atan2f.argtypes = [C.c_float, C.c_float]; L.mort_oracle_atan2f.restype = C.c_float
        L.mort_oracle_sin.argtypes = [C.c_double]; L.mort_oracle_sin.restype = C.c_double
        _lib = L
    return _lib


STATE_DTYPE = np.dtype([("d", "<u4"), ("v", "<u4", (5,)), ("bf", "<i4"), ("bfd", "<i4"), ("be", "<f4"),
                        ("pad", "<u4"), ("bed", "<f8")])
assert STATE_DTYPE.itemsize == 48


def seed_states(seed, width, height):
    st = np.zeros(width * height, dtype=STATE_DTYPE)
    lib().mort_oracle_rng_seed(st.ctypes.data, seed, width, height)
    return st


def render(world, cam, states=None, seed=S.DEFAULT_SEED, rows=None, nthreads=8, want_accum=True, want_segments=True):
    """Oracle render. Returns dict(rgba, accum, segments_px, segments, rng_draws, states)."""
    W, H = cam.image_width, cam.image_height
    if states is None:
        states = seed_states(seed, W, H)
    rgba = np.zeros((H, W, 4), dtype=np.uint8)
    accum = np.zeros((H, W, 3), dtype=np.float32) if want_accum else None
    seg = np.zeros((H, W), dtype=np.uint32) if want_segments else None
    r0, r1 = rows if rows is not None else (0, H)
    st = Stats()
    rc = lib().mort_oracle_render(world.ptr, C.byref(cam), states.ctypes.data, r0, r1, rgba.ctypes.data,
                                  accum.ctypes.data if accum is not None else None,
                                  seg.ctypes.data if seg is not None else None, nthreads, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"mort_oracle_render failed: {rc}")
    return dict(rgba=rgba, accum=accum, segments_px=seg, segments=int(st.segments), rng_draws=int(st.rng_draws),
                states=states)
