"""ctypes binding of libmort_host.so -- the C host scene layer
(include/mort_host.h): world containers, constructors, BVH builder, camera
set-up and the reference's ten built-in scenes (mort.cu:129-631,649-689).
No GPU dependency.
"""
import ctypes as C
import os

import numpy as np

from . import structs as S

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libmort_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `make host` (or __graft_entry__.build())")
        L = C.CDLL(_LIB_PATH)
        W, Cam = C.POINTER(S.World), C.POINTER(S.Camera)
        L.mort_world_init.argtypes = [W]; L.mort_world_init.restype = C.c_int
        L.mort_world_free.argtypes = [W]; L.mort_world_free.restype = None
        L.mort_scene_build.argtypes = [C.c_int, W, Cam, C.POINTER(S.SceneOpts)]; L.mort_scene_build.restype = C.c_int
        L.mort_camera_defaults.argtypes = [Cam]; L.mort_camera_defaults.restype = None
        L.mort_camera_initialize.argtypes = [Cam]; L.mort_camera_initialize.restype = None
        L.mort_camera_effective_spp.argtypes = [Cam]; L.mort_camera_effective_spp.restype = C.c_int
        L.mort_host_rng_init.argtypes = [C.POINTER(S.HostRng), C.c_uint32, C.c_int]
        L.mort_host_rand.argtypes = [C.POINTER(S.HostRng)]; L.mort_host_rand.restype = C.c_int
        L.mort_host_random_float.argtypes = [C.POINTER(S.HostRng)]; L.mort_host_random_float.restype = C.c_float
        i, f, b, V = C.c_int, C.c_float, C.c_bool, S.Vec3
        for name, args in {
            "mort_add_solid_color": [W, V],
            "mort_add_checker_texture": [W, f, i, i, i, i],
            "mort_add_image_texture": [W, C.c_void_p, i, i],
            "mort_add_noise_texture": [W, f, C.POINTER(S.HostRng)],
            "mort_add_lambertian": [W, i, i],
            "mort_add_metal": [W, V, f],
            "mort_add_dielectric": [W, f],
            "mort_add_diffuse_light": [W, i, i],
            "mort_add_isotropic": [W, i, i],
            "mort_add_sphere": [W, V, f, i, i, b],
            "mort_add_moving_sphere": [W, V, V, f, i, i, b],
            "mort_add_quad": [W, V, V, V, i, i, b],
            "mort_add_translate": [W, i, i, V, b],
            "mort_add_rotate_y": [W, i, i, f, b],
            "mort_add_constant_medium": [W, i, i, f, i, i, b],
            "mort_add_hittable_list": [W, b],
            "mort_list_add": [W, i, i, i],
            "mort_add_bvh": [W, i, b],
        }.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.mort_box.argtypes = [W, V, V, i, i]; L.mort_box.restype = None
        L.mort_rotated_box.argtypes = [W, V, V, f, i, i]; L.mort_rotated_box.restype = None
        L.mort_rotated_smoke_box.argtypes = [W, V, V, f, f, i, i]; L.mort_rotated_smoke_box.restype = None
        L.mort_write_ppm.argtypes = [C.c_char_p, C.c_void_p, i, i]; L.mort_write_ppm.restype = C.c_int
        _lib = L
    return _lib


def vec3(x, y, z):
    return S.Vec3((C.c_float * 3)(x, y, z))


class World:
    """Owns a mort_world whose host arrays are allocated at the reference's capacities."""

    def __init__(self):
        self.c = S.World()
        if lib().mort_world_init(C.byref(self.c)) != 0:
            raise MemoryError("mort_world_init failed")
        self._keepalive = []

    def close(self):
        if self.c is not None:
            lib().mort_world_free(C.byref(self.c))
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ptr(self):
        return C.byref(self.c)

    def counts(self):
        o, m, t = self.c.objs, self.c.mats, self.c.texs
        return dict(spheres=o.num_spheres, quads=o.num_quads, translates=o.num_translates, rotate_y=o.num_rotate_y,
                    constant_medium=o.num_constant_medium, hittable_list=o.num_hittable_list, bvh=o.num_bvh,
                    lambertians=m.num_lambertians, metals=m.num_metals, dielectrics=m.num_dielectrics,
                    diffuse_lights=m.num_diffuse_lights, isotropics=m.num_isotropics,
                    solid_colors=t.num_solid_colors, checker_textures=t.num_checker_textures,
                    image_textures=t.num_image_textures, noise_textures=t.num_noise_textures)


def synthetic_earth(width=1024, height=512):
    """Deterministic stand-in for imgs/earthmap.jpg (same 1024x512 RGB shape) used when the
    decoded texels are not supplied: smooth bands + a checker so u/v errors show up."""
    y, x = np.mgrid[0:height, 0:width]
    r = (x * 255 // (width - 1)).astype(np.uint8)
    g = (y * 255 // (height - 1)).astype(np.uint8)
    b = ((((x // 32) + (y // 32)) % 2) * 200 + 20).astype(np.uint8)
    return np.ascontiguousarray(np.stack([r, g, b], axis=-1))


_EARTH_FIXTURE = os.path.join(os.path.dirname(_HERE), "tests", "golden", "earthmap_rgb.npz")


_EARTH_JPEG = os.path.join(os.path.dirname(_HERE), "tests", "golden", "earthmap.jpg")


def read_image(path):
    """JPEG (the product's own baseline decoder, mort_jpeg.c) or PPM -> HxWx3 uint8, or None."""
    L = lib()
    L.mort_read_image.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mort_read_image.restype = C.c_void_p
    w, h = C.c_int(0), C.c_int(0)
    p = L.mort_read_image(path.encode(), C.byref(w), C.byref(h))
    if not p:
        return None
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(h.value, w.value, 3)).copy()
    C.CDLL(None).free(C.c_void_p(p))
    return out


def load_earth():
    """Texels of the reference's imgs/earthmap.jpg (tests/golden/earthmap.jpg is that data file), decoded by the
    product's own JPEG decoder -- byte-identical to the reference's stb_image decode (tests/test_jpeg.py pins it against
    tests/golden/earthmap_rgb.npz); the decoded fixture or a synthetic stand-in if the file is absent."""
    if os.path.exists(_EARTH_JPEG):
        img = read_image(_EARTH_JPEG)
        if img is not None:
            return img
    if os.path.exists(_EARTH_FIXTURE):
        return np.ascontiguousarray(np.load(_EARTH_FIXTURE)["rgb"])
    return synthetic_earth()


def build_scene(scene_id, width=None, spp=None, depth=None, aspect=None, args_rtl=0, earth=None):
    """mort <scene_id> plus the CLI overrides. Returns (World, Camera) with the camera initialised.

    earth: optional HxWx3 uint8 array of decoded earthmap texels for scenes 3/8/9."""
    w = World()
    cam = S.Camera()
    opts = S.SceneOpts()
    opts.args_rtl = args_rtl
    if scene_id in (3, 8, 9):
        if earth is None:
            earth = load_earth()
        earth = np.ascontiguousarray(earth, dtype=np.uint8)
        w._keepalive.append(earth)
        opts.earth_texels = earth.ctypes.data
        opts.earth_height, opts.earth_width = earth.shape[0], earth.shape[1]
    lib().mort_scene_build(scene_id, w.ptr, C.byref(cam), C.byref(opts))
    if width is not None:
        cam.image_width = int(width)
    if spp is not None:
        cam.samples_per_pixel = int(spp)
    if depth is not None:
        cam.bounce_limit = int(depth)
    if aspect is not None:
        cam.aspect_ratio = float(aspect)
    lib().mort_camera_initialize(C.byref(cam))
    return w, cam


def effective_spp(cam):
    return lib().mort_camera_effective_spp(C.byref(cam))


def write_ppm(path, rgba, width, height):
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    return lib().mort_write_ppm(path.encode(), rgba.ctypes.data, width, height)
