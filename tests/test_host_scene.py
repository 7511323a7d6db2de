"""Host scene layer (libmort_host.so): struct layout, the ten built-in scenes, BVH builder
invariants (objects.cuh:529-611 semantics), camera set-up (camera.cuh:47-84), host LCG."""
import ctypes as C
import math

import numpy as np
import pytest

from mort_amd import host, structs as S


def test_struct_sizes_match_reference_layout():
    # sizes of the reference structs on x86-64 (also static-asserted in mort_scene.h)
    assert C.sizeof(S.Sphere) == 72 and C.sizeof(S.Quad) == 108 and C.sizeof(S.Bvh) == 41992
    assert C.sizeof(S.HittableList) == 8036 and C.sizeof(S.NoiseTexture) == 6152
    assert C.sizeof(S.Camera) == 240 and C.sizeof(S.World) == 264 and C.sizeof(S.RngState) == 48


def test_msvc_rand_model():
    g = S.HostRng()
    host.lib().mort_host_rng_init(C.byref(g), 1, 0)
    # first outputs of MSVC rand() with the default seed 1
    assert [host.lib().mort_host_rand(C.byref(g)) for _ in range(5)] == [41, 18467, 6334, 26500, 19169]


SCENE_COUNTS = {
    1: dict(spheres=486, bvh=1, hittable_list=1, lambertians=399, metals=65, dielectrics=22, solid_colors=400, checker_textures=1),
    2: dict(spheres=2, lambertians=1, checker_textures=1, solid_colors=2),
    3: dict(spheres=1, image_textures=1, lambertians=1),
    4: dict(spheres=2, noise_textures=1, lambertians=1),
    5: dict(quads=5, lambertians=5, solid_colors=5),
    6: dict(spheres=1, quads=12, translates=1, rotate_y=1, hittable_list=2, lambertians=3, diffuse_lights=1, dielectrics=1),
    7: dict(quads=18, translates=2, rotate_y=2, constant_medium=2, hittable_list=2, lambertians=5, diffuse_lights=1),
    8: dict(spheres=1007, quads=2401, translates=1, rotate_y=1, constant_medium=2, hittable_list=1, lambertians=7, metals=1,
            dielectrics=1, diffuse_lights=1, image_textures=1, noise_textures=1),
    10: dict(spheres=35, bvh=1, hittable_list=1, lambertians=35),
}


@pytest.mark.parametrize("sid", sorted(SCENE_COUNTS))
def test_scene_catalogue_counts(sid):
    w, cam = host.build_scene(sid)
    got = w.counts()
    for k, v in SCENE_COUNTS[sid].items():
        assert got[k] == v, (sid, k, got[k], v)


def test_scene_camera_parameters():
    # (width, height, spp, depth, light type) per mort.cu scene functions
    want = {1: (1200, 675, 100, 20, -1), 2: (1200, 675, 20, 50, -1), 5: (400, 400, 100, 50, -1),
            6: (600, 600, 1000, 50, S.OBJ_HITTABLE_LIST), 7: (800, 800, 2000, 50, 4), 8: (800, 800, 1000, 40, S.OBJ_QUAD),
            9: (400, 400, 250, 4, S.OBJ_QUAD), 10: (1200, 675, 1, 5, -1)}
    for sid, (W, H, spp, depth, lt) in want.items():
        _, cam = host.build_scene(sid)
        assert (cam.image_width, cam.image_height, cam.samples_per_pixel, cam.bounce_limit, cam.light_obj_type) == (W, H, spp, depth, lt)
    assert host.effective_spp(host.build_scene(1, spp=500)[1]) == 484
    assert host.effective_spp(host.build_scene(6)[1]) == 961


def test_unknown_scene_is_empty_world():
    w, cam = host.build_scene(11)
    assert all(v == 0 for v in w.counts().values())


def test_camera_initialize_against_float64_math():
    _, cam = host.build_scene(1)
    assert cam.sqrt_spp == 10 and cam.pixel_samples_scale == np.float32(0.01)
    lookfrom = np.array([13.0, 2.0, 3.0])
    wv = lookfrom / np.linalg.norm(lookfrom)
    u = np.cross([0, 1, 0], wv); u /= np.linalg.norm(u)
    v = np.cross(wv, u)
    h = math.tan(math.radians(20) / 2)
    vh = 2 * h * 10.0
    vw = vh * (1200 / 675)
    du = vw * u / 1200
    dv = vh * v / 675          # pixel_delta_v = -viewport_v / H with viewport_v = vh * (-v): points up
    ul = lookfrom - 10 * wv - vw * u / 2 - vh * v / 2
    p00 = ul + 0.5 * (du + dv)
    np.testing.assert_allclose(cam.pixel_delta_u.tolist(), du, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(cam.pixel_delta_v.tolist(), dv, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(cam.pixel00_loc.tolist(), p00, rtol=2e-6)
    assert cam.pixel_delta_v.e[1] > 0  # row 0 is the bottom row


def test_bvh_builder_invariants():
    w, _ = host.build_scene(1)
    b = w.c.objs.host_bvh[0]
    spheres = w.c.objs.host_sphere
    n_nodes = 0
    seen = []
    stack = [0]
    while stack:
        n = stack.pop()
        n_nodes += 1
        bb = b.bounding_boxes[n]
        if b.is_internal_node[n]:
            assert b.left_children_types[n] == S.OBJ_BVH and b.right_children_types[n] == S.OBJ_BVH
            for ch in (b.left_children_idxs[n], b.right_children_idxs[n]):
                cb = b.bounding_boxes[ch]
                for ax in "xyz":
                    assert getattr(cb, ax).imin >= getattr(bb, ax).imin and getattr(cb, ax).imax <= getattr(bb, ax).imax
                stack.append(ch)
        else:
            kids = {(b.left_children_types[n], b.left_children_idxs[n]), (b.right_children_types[n], b.right_children_idxs[n])}
            for t, i in kids:
                assert t == S.OBJ_SPHERE
                seen.append(i)
                sb = spheres[i].bbox
                for ax in "xyz":
                    assert getattr(sb, ax).imin >= getattr(bb, ax).imin and getattr(sb, ax).imax <= getattr(bb, ax).imax
    assert n_nodes == 511 and sorted(seen) == list(range(486))


def test_bvh_sort_is_stable_median_split():
    """Scene 10 inserts 35 spheres in descending order; after the build the first leaf holds the smallest."""
    w, _ = host.build_scene(10)
    b = w.c.objs.host_bvh[0]
    n = 0
    while b.is_internal_node[n]:
        n = b.left_children_idxs[n]
    first = w.c.objs.host_sphere[b.left_children_idxs[n]]
    assert first.center1.e[0] == 1.0  # centres are (35-i, 35-i, 35-i): smallest is 1


def test_arg_order_profile_changes_scene1_layout_only():
    a, _ = host.build_scene(1, args_rtl=0)
    b, _ = host.build_scene(1, args_rtl=1)
    assert a.counts() == b.counts()  # both orders give 486 spheres
    ca = [a.c.objs.host_sphere[i].center1.tolist() for i in range(486)]
    cb = [b.c.objs.host_sphere[i].center1.tolist() for i in range(486)]
    assert ca != cb
