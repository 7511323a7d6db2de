"""ctypes binding of libmort_hip.so (include/mort_hip.h): the gfx950 render path.

There is no CPU fallback: if the library is missing or no MI355X is present
every call raises.  The four call sites of the reference this replaces are
world::toDevice() (world.cuh:98-102), setup_rng<<<>>> (mort.cu:709), the
per-bounce scratch allocations (mort.cu:712-725) and renderKernel<<<>>>
(mort.cu:106).
"""
import ctypes as C
import os

import numpy as np

from . import structs as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MORT_HIP_LIB") or os.path.join(_HERE, "lib", "libmort_hip.so")  # override: debug builds only
_lib = None

MODE_MEGA = 0
MODE_WAVE = 1
MODE_THROUGHPUT = 2  # non-parity: one stream per (pixel, stratum row); see include/mort_hip.h

EXPORTS = [
    "mort_hip_strerror", "mort_hip_last_error", "mort_hip_init", "mort_hip_shutdown", "mort_hip_upload_world",
    "mort_hip_set_partition", "mort_hip_rng_seed", "mort_hip_rng_load", "mort_hip_rng_store", "mort_hip_render",
    "mort_hip_render_device", "mort_hip_local_rows", "mort_hip_global_row", "mort_hip_rng_seed_host", "mort_hip_render_host",
    "mort_hip_comm_id", "mort_hip_comm_init", "mort_hip_comm_destroy", "mort_hip_render_gather", "mort_hip_comm_selftest",
    "mort_hip_calib_valu", "mort_hip_calib_hbm_copy",
]
HOST_TREE = 1


class Partition(C.Structure):
    _fields_ = [("rank", C.c_int), ("nranks", C.c_int), ("rows_per_block", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("seconds", C.c_double), ("segments", C.c_uint64), ("pixels", C.c_uint64),
                ("eff_samples", C.c_uint64), ("rng_draws", C.c_uint64), ("algorithmic_hbm_bytes", C.c_uint64),
                ("scene_in_lds", C.c_int), ("local_rows", C.c_int), ("kernel_vgprs", C.c_int),
                ("kernel_lds_bytes", C.c_int), ("reference_walks", C.c_uint64), ("kernel_name", C.c_char * 64), ("gather_seconds", C.c_double)]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["kernel_name"] = d["kernel_name"].decode()
        return d


class CalibValu(C.Structure):
    _fields_ = [("waves_per_simd", C.c_int), ("kind", C.c_int), ("seconds", C.c_double), ("cycles_per_wave", C.c_double),
                ("clock_ghz", C.c_double), ("valu_per_wave", C.c_double), ("simds_seen", C.c_int), ("resident_waves_per_simd", C.c_double),
                ("cycles_per_valu_per_wave", C.c_double), ("cycles_per_valu_per_simd", C.c_double)]


class MortHipError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__(f"{what}: {status} ({lib().mort_hip_strerror(status).decode()}) {detail}".strip())


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `make hip` (or __graft_entry__.build()); "
                               "there is no CPU fallback for the render path")
        L = C.CDLL(LIB_PATH)
        ctx = C.c_void_p
        L.mort_hip_strerror.argtypes = [C.c_int]; L.mort_hip_strerror.restype = C.c_char_p
        L.mort_hip_last_error.argtypes = [ctx]; L.mort_hip_last_error.restype = C.c_char_p
        L.mort_hip_init.argtypes = [C.c_int, C.POINTER(ctx)]; L.mort_hip_init.restype = C.c_int
        L.mort_hip_shutdown.argtypes = [ctx]; L.mort_hip_shutdown.restype = None
        L.mort_hip_upload_world.argtypes = [ctx, C.POINTER(S.World)]; L.mort_hip_upload_world.restype = C.c_int
        L.mort_hip_set_partition.argtypes = [ctx, C.POINTER(Partition)]; L.mort_hip_set_partition.restype = C.c_int
        L.mort_hip_rng_seed.argtypes = [ctx, C.c_uint64, C.c_int, C.c_int]; L.mort_hip_rng_seed.restype = C.c_int
        L.mort_hip_rng_load.argtypes = [ctx, C.c_void_p, C.c_int, C.c_int]; L.mort_hip_rng_load.restype = C.c_int
        L.mort_hip_rng_store.argtypes = [ctx, C.c_void_p, C.c_int, C.c_int]; L.mort_hip_rng_store.restype = C.c_int
        L.mort_hip_render.argtypes = [ctx, C.POINTER(S.Camera), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(Stats)]
        L.mort_hip_render.restype = C.c_int
        L.mort_hip_render_device.argtypes = [ctx, C.POINTER(S.Camera), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(Stats)]
        L.mort_hip_render_device.restype = C.c_int
        L.mort_hip_local_rows.argtypes = [ctx, C.c_int]; L.mort_hip_local_rows.restype = C.c_int
        L.mort_hip_global_row.argtypes = [ctx, C.c_int]; L.mort_hip_global_row.restype = C.c_int
        L.mort_hip_rng_seed_host.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p]; L.mort_hip_rng_seed_host.restype = C.c_int
        L.mort_hip_render_host.argtypes = [C.POINTER(S.World), C.POINTER(S.Camera), C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        L.mort_hip_render_host.restype = C.c_int
        L.mort_hip_calib_valu.argtypes = [ctx, C.c_int, C.c_int, C.POINTER(CalibValu)]; L.mort_hip_calib_valu.restype = C.c_int
        L.mort_hip_calib_hbm_copy.argtypes = [ctx, C.c_size_t, C.c_int, C.POINTER(C.c_double)]; L.mort_hip_calib_hbm_copy.restype = C.c_int
        _lib = L
    return _lib


class Context:
    """One mort_ctx: one GPU, one partition, one uploaded world."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        st = lib().mort_hip_init(device, C.byref(self._h))
        if st != 0:
            self._h = None
            raise MortHipError(st, "mort_hip_init")

    def _chk(self, st, what):
        if st != 0:
            raise MortHipError(st, what, lib().mort_hip_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().mort_hip_shutdown(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload_world(self, world):
        self._chk(lib().mort_hip_upload_world(self._h, world.ptr), "mort_hip_upload_world")

    def set_partition(self, rank, nranks, rows_per_block=8):
        p = Partition(rank, nranks, rows_per_block)
        self._chk(lib().mort_hip_set_partition(self._h, C.byref(p)), "mort_hip_set_partition")

    def rng_seed(self, seed, width, height):
        self._chk(lib().mort_hip_rng_seed(self._h, seed, width, height), "mort_hip_rng_seed")

    def rng_load(self, states, width, height):
        assert states.nbytes == width * height * 48
        self._chk(lib().mort_hip_rng_load(self._h, states.ctypes.data, width, height), "mort_hip_rng_load")

    def rng_store(self, width, height, dtype=None):
        out = np.zeros(width * height * 48, dtype=np.uint8)
        self._chk(lib().mort_hip_rng_store(self._h, out.ctypes.data, width, height), "mort_hip_rng_store")
        return out.view(dtype) if dtype is not None else out

    def local_rows(self, height):
        return lib().mort_hip_local_rows(self._h, height)

    def global_row(self, local_row):
        return lib().mort_hip_global_row(self._h, local_row)

    def render(self, cam, mode=MODE_MEGA, want_accum=True, want_segments=False):
        """Renders the owned rows into full-size host arrays (rows not owned stay zero)."""
        W, H = cam.image_width, cam.image_height
        rgba = np.zeros((H, W, 4), dtype=np.uint8)
        accum = np.zeros((H, W, 3), dtype=np.float32) if want_accum else None
        seg = np.zeros((H, W), dtype=np.uint32) if want_segments else None
        st = Stats()
        self._chk(lib().mort_hip_render(self._h, C.byref(cam), mode, rgba.ctypes.data,
                                        accum.ctypes.data if accum is not None else None,
                                        seg.ctypes.data if seg is not None else None, C.byref(st)), "mort_hip_render")
        return dict(rgba=rgba, accum=accum, segments_px=seg, stats=st.asdict())

    def render_device(self, cam, d_rgba, d_accum=0, stream=0, mode=MODE_MEGA, sync=True):
        """Render into caller-owned device buffers (packed owned rows); d_* are raw device addresses."""
        st = Stats()
        self._chk(lib().mort_hip_render_device(self._h, C.byref(cam), mode, d_rgba, d_accum or None, stream or None,
                                               C.byref(st) if sync else None), "mort_hip_render_device")
        return st.asdict() if sync else None

    def calib_valu(self, waves_per_simd, kind=0):
        """Shader cycles one SIMD needs per wave64 VALU instruction at `waves_per_simd` resident waves (include/mort_hip.h)."""
        r = CalibValu()
        self._chk(lib().mort_hip_calib_valu(self._h, waves_per_simd, kind, C.byref(r)), "mort_hip_calib_valu")
        return {k: getattr(r, k) for k, _ in r._fields_}

    def calib_hbm_copy(self, nbytes=1 << 30, reps=3):
        """GB/s (read + write) of a float4 copy of `nbytes` per buffer: the HBM rate this box reaches."""
        g = C.c_double(0)
        self._chk(lib().mort_hip_calib_hbm_copy(self._h, nbytes, reps, C.byref(g)), "mort_hip_calib_hbm_copy")
        return g.value


def seed_states_host(seed, width, height, dtype=None):
    """curand_init(seed, x + y*W, 0) for every pixel, on the host (mort_hip_rng_seed_host): W*H 48-byte records."""
    out = np.zeros(width * height * 48, dtype=np.uint8)
    st = lib().mort_hip_rng_seed_host(seed, width, height, out.ctypes.data)
    if st != 0:
        raise MortHipError(st, "mort_hip_rng_seed_host")
    return out.view(dtype) if dtype is not None else out


def render_host(world, cam, states=None, seed=S.DEFAULT_SEED, nthreads=1, tree=False, want_accum=True, want_segments=True):
    """`mort --mode host`: the kernel body as a host loop (mort_hip_render_host).  No GPU involved.  Returns the same
    dict as Context.render plus the final states (raw uint8 view of the 48-byte records)."""
    W, H = cam.image_width, cam.image_height
    st8 = seed_states_host(seed, W, H) if states is None else np.ascontiguousarray(states).view(np.uint8).reshape(-1).copy()
    assert st8.nbytes == W * H * 48
    rgba = np.zeros((H, W, 4), dtype=np.uint8)
    accum = np.zeros((H, W, 3), dtype=np.float32) if want_accum else None
    seg = np.zeros((H, W), dtype=np.uint32) if want_segments else None
    stt = Stats()
    rc = lib().mort_hip_render_host(world.ptr, C.byref(cam), st8.ctypes.data, nthreads, HOST_TREE if tree else 0, rgba.ctypes.data,
                                    accum.ctypes.data if accum is not None else None, seg.ctypes.data if seg is not None else None, C.byref(stt))
    if rc != 0:
        raise MortHipError(rc, "mort_hip_render_host")
    return dict(rgba=rgba, accum=accum, segments_px=seg, stats=stt.asdict(), states=st8)
