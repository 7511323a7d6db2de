"""More closed-form known answers for the CPU oracle (nothing from the reference can pin the oracle, so
every primitive, instance, medium, material and texture gets an answer derived by hand from the reference's formulas):
translate / rotate_y instances (objects.cuh:268-278,334-366), constant_medium (objects.cuh:396-434), diffuse_light, metal,
dielectric, lambertian incl. the two different pi literals (materials.cuh:54 vs pdf.cuh:48), bounce limit, Perlin noise."""
import ctypes as C
import math

import numpy as np
import pytest

from mort_amd import host, structs as S


def ray7(o, d, t=0.0):
    return (C.c_float * 7)(*o, *d, t)


def f3(v):
    return (C.c_float * 3)(*v)


def mat(w, kind, *a):
    L = host.lib()
    if kind in ("lamb", "light", "iso"):
        col = L.mort_add_solid_color(w.ptr, host.vec3(*a[0]))
        add = {"lamb": L.mort_add_lambertian, "light": L.mort_add_diffuse_light, "iso": L.mort_add_isotropic}[kind]
        return {"lamb": S.MAT_LAMBERTIAN, "light": S.MAT_DIFFUSE_LIGHT, "iso": S.MAT_ISOTROPIC}[kind], add(w.ptr, S.TEXTURE_SOLID, col)
    if kind == "metal":
        return S.MAT_METAL, L.mort_add_metal(w.ptr, host.vec3(*a[0]), a[1])
    return S.MAT_DIELECTRIC, L.mort_add_dielectric(w.ptr, a[0])


def camera(background=(0.0, 0.0, 0.0), depth=10):
    _, cam = host.build_scene(2, width=16, spp=1, depth=depth)
    for i in range(3):
        cam.background.e[i] = background[i]
    cam.light_obj_type = -1
    host.lib().mort_camera_initialize(C.byref(cam))
    return cam


def color(oracle, w, cam, o, d, seed=1):
    st = S.RngState()
    oracle.lib().mort_oracle_rng_init(st, seed, 0)
    out = (C.c_float * 3)()
    oracle.lib().mort_oracle_ray_color(w.ptr, cam, ray7(o, d), st, out)
    return list(out)


def test_translate_and_rotate_y_instances(oracle):
    L = host.lib()
    w = host.World()
    mt, mi = mat(w, "lamb", (0.5, 0.5, 0.5))
    s = L.mort_add_sphere(w.ptr, host.vec3(0, 0, 0), 1.0, mt, mi, True)
    L.mort_add_translate(w.ptr, S.OBJ_SPHERE, s, host.vec3(3, 0, -5), False)
    st, hit = S.RngState(), oracle.Hit()
    assert oracle.lib().mort_oracle_world_hit(w.ptr, ray7((3, 0, 0), (0, 0, -1)), 0.001, math.inf, st, hit)
    assert hit.t == 4.0 and hit.p.tolist() == [3.0, 0.0, -4.0] and hit.normal.tolist() == [0.0, 0.0, 1.0]
    assert not oracle.lib().mort_oracle_world_hit(w.ptr, ray7((0, 0, 0), (0, 0, -1)), 0.001, math.inf, st, hit)
    # a quad facing +z at z = -2, rotated by +90 degrees about y: it now faces +x at x = -2 (objects.cuh:334-366)
    w2 = host.World()
    mt, mi = mat(w2, "lamb", (0.5, 0.5, 0.5))
    q = L.mort_add_quad(w2.ptr, host.vec3(-1, -1, -2), host.vec3(2, 0, 0), host.vec3(0, 2, 0), mt, mi, True)
    L.mort_add_rotate_y(w2.ptr, S.OBJ_QUAD, q, 90.0, False)
    assert oracle.lib().mort_oracle_world_hit(w2.ptr, ray7((5, 0, 0), (-1, 0, 0)), 0.001, math.inf, st, hit)
    assert hit.t == pytest.approx(7.0, abs=1e-5) and hit.p.tolist() == pytest.approx([-2.0, 0.0, 0.0], abs=1e-5)
    assert hit.normal.tolist() == pytest.approx([1.0, 0.0, 0.0], abs=1e-6) and hit.front_face
    assert not oracle.lib().mort_oracle_world_hit(w2.ptr, ray7((0, 0, 5), (0, 0, -1)), 0.001, 6.9, st, hit)


def test_constant_medium_distance_from_the_stream(oracle):
    """t = t1 + (-1/density) * log(u) / |d| with u the next random_float of the stream, accepted iff it falls inside
    the boundary (objects.cuh:396-434); the draw happens only when the ray spans the boundary."""
    L = host.lib()
    for density, seed in ((0.9, 3), (0.4, 5), (2.5, 11)):
        w = host.World()
        bt, bi = mat(w, "glass", 1.5)
        b = L.mort_add_sphere(w.ptr, host.vec3(0, 0, -5), 1.0, bt, bi, True)
        pt, pi_ = mat(w, "iso", (0.8, 0.8, 0.8))
        L.mort_add_constant_medium(w.ptr, S.OBJ_SPHERE, b, density, pt, pi_, False)
        st, probe, hit = S.RngState(), S.RngState(), oracle.Hit()
        oracle.lib().mort_oracle_rng_init(st, seed, 0)
        oracle.lib().mort_oracle_rng_init(probe, seed, 0)
        u = oracle.lib().mort_oracle_random_float(probe)
        want = 4.0 + (-1.0 / density) * math.log(u) / 2.0  # |d| = 2: boundary spans t in [2, 3] -> distance inside = 2
        got = oracle.lib().mort_oracle_world_hit(w.ptr, ray7((0, 0, 0), (0, 0, -2)), 0.001, math.inf, st, hit)
        inside = (-1.0 / density) * math.log(u) <= 2.0
        assert got == inside
        if got:
            assert hit.t == pytest.approx(2.0 + (want - 4.0), rel=1e-5) and hit.normal.tolist() == [1.0, 0.0, 0.0] and hit.front_face
        assert st.d == probe.d and list(st.v) == list(probe.v)  # exactly one draw
        # a ray that misses the boundary draws nothing
        st2 = S.RngState(); oracle.lib().mort_oracle_rng_init(st2, seed, 0)
        assert not oracle.lib().mort_oracle_world_hit(w.ptr, ray7((0, 3, 0), (0, 0, -1)), 0.001, math.inf, st2, hit)
        ref = S.RngState(); oracle.lib().mort_oracle_rng_init(ref, seed, 0)
        assert st2.d == ref.d and list(st2.v) == list(ref.v)


def test_materials_closed_form(oracle):
    L = host.lib()
    bg = (0.25, 0.5, 0.75)
    # miss -> background; depth 0 -> black (camera.cuh:154-163)
    empty = host.World()
    assert color(oracle, empty, camera(bg), (0, 0, 0), (0, 0, -1)) == pytest.approx(list(bg))
    # diffuse light: emission on the front face, nothing from behind, no scatter (materials.cuh:151-163)
    w = host.World()
    mt, mi = mat(w, "light", (4, 3, 2))
    L.mort_add_quad(w.ptr, host.vec3(-1, -1, -3), host.vec3(2, 0, 0), host.vec3(0, 2, 0), mt, mi, False)  # normal +z
    assert color(oracle, w, camera(bg), (0, 0, 0), (0, 0, -1)) == [4.0, 3.0, 2.0]
    assert color(oracle, w, camera(bg), (0, 0, -6), (0, 0, 1)) == [0.0, 0.0, 0.0]
    assert color(oracle, w, camera(bg, depth=0), (0, 0, 0), (0, 0, -1)) == [0.0, 0.0, 0.0]
    # mirror (fuzz 0): albedo x what the reflected ray sees; the 45-degree quad (normal (0,-1,1)/sqrt 2) sends -z to -y
    w = host.World()
    mt, mi = mat(w, "metal", (0.8, 0.6, 0.4), 0.0)
    L.mort_add_quad(w.ptr, host.vec3(-1, -1, -4), host.vec3(2, 0, 0), host.vec3(0, 2, 2), mt, mi, False)
    lt, li = mat(w, "light", (5, 5, 5))
    L.mort_add_quad(w.ptr, host.vec3(-50, -20, -50), host.vec3(0, 0, 100), host.vec3(100, 0, 0), lt, li, False)  # floor light, normal u x v = +y
    got = color(oracle, w, camera((0, 0, 0)), (0, 0, 0), (0, 0, -1))
    assert got == pytest.approx([0.8 * 5, 0.6 * 5, 0.4 * 5], rel=1e-6)
    # glass sphere at normal incidence: reflect or refract, attenuation 1 either way, every path ends on the background
    w = host.World()
    mt, mi = mat(w, "glass", 1.5)
    L.mort_add_sphere(w.ptr, host.vec3(0, 0, -5), 1.0, mt, mi, False)
    for seed in range(1, 6):
        assert color(oracle, w, camera(bg, depth=50), (0, 0, 0), (0, 0, -1), seed) == pytest.approx(list(bg), rel=1e-6)
    # lambertian sphere under a uniform sky: one bounce, then the sky; weight = albedo * scatter_pdf / pdf with the
    # reference's two pi literals: (cos / 3.141592565) / (cos / 3.1415926)  (materials.cuh:54, pdf.cuh:48)
    w = host.World()
    mt, mi = mat(w, "lamb", (0.6, 0.4, 0.2))
    L.mort_add_sphere(w.ptr, host.vec3(0, 0, -5), 1.0, mt, mi, False)
    ratio = 3.1415926 / 3.141592565
    for seed in range(1, 6):
        got = color(oracle, w, camera((1, 1, 1)), (0, 0, 0), (0, 0, -1), seed)
        assert got == pytest.approx([0.6 * ratio, 0.4 * ratio, 0.2 * ratio], rel=2e-6)
    assert color(oracle, w, camera((0, 0, 0)), (0, 0, 0), (0, 0, -1)) == [0.0, 0.0, 0.0]
    # isotropic phase function: scatter_pdf = pdf = 1 / (4 pi) -> weight = albedo exactly
    w = host.World()
    mt, mi = mat(w, "iso", (0.3, 0.6, 0.9))
    L.mort_add_sphere(w.ptr, host.vec3(0, 0, -5), 1.0, mt, mi, False)
    got = color(oracle, w, camera((1, 1, 1), depth=1), (0, 0, 0), (0, 0, -1))
    assert got == [0.0, 0.0, 0.0]  # depth 1: the scattered ray is never traced
    vals = [color(oracle, w, camera((1, 1, 1), depth=50), (0, 0, 0), (0, 0, -1), s) for s in range(1, 9)]
    for v in vals:  # each path: albedo^k for the k bounces it took inside / off the sphere
        k = round(math.log(v[0]) / math.log(0.3))
        assert k >= 1 and v == pytest.approx([0.3**k, 0.6**k, 0.9**k], rel=1e-5)


def test_perlin_noise_texture_properties(oracle):
    """noise_texture::value = 0.5 (1 + sin(scale z + 10 turb)) * (1,1,1) (textures.cuh:198-202): grey, in [0, 1],
    deterministic, and continuous (a 1e-4 step moves it by < 5e-2)."""
    w, _ = host.build_scene(4)
    out, out2 = (C.c_float * 3)(), (C.c_float * 3)()
    rng = np.random.default_rng(5)
    for p in rng.uniform(-20, 20, (200, 3)):
        oracle.lib().mort_oracle_texture_value(w.ptr, S.TEXTURE_NOISE, 0, 0.0, 0.0, f3(p), out)
        assert out[0] == out[1] == out[2] and 0.0 <= out[0] <= 1.0
        oracle.lib().mort_oracle_texture_value(w.ptr, S.TEXTURE_NOISE, 0, 0.3, 0.7, f3(p), out2)
        assert list(out) == list(out2)  # uv are not read
        oracle.lib().mort_oracle_texture_value(w.ptr, S.TEXTURE_NOISE, 0, 0.0, 0.0, f3(p + 1e-4), out2)
        assert abs(out[0] - out2[0]) < 5e-2
