"""`mort --mode host` (mort_hip_render_host): the product's kernel body as a host loop -- the same compiled functions
the GPU kernels run (dev_pixel.h / dev_trace.h / dev_shade.h / dev_gen.h), no GPU, no oracle in the path.  Checked here
against the committed golden vectors and the CPU oracle, bit for bit, in both forms: the reference's scan over the
flattened items, and the single-lane walk of this build's unified tree (what one GPU lane of mega_gen_kernel does)."""
import os

import numpy as np
import pytest

from mort_amd import host, hip, structs as S
from tests.golden.make_golden import CASES

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "oracle_golden.npz"))


def same(out, ref, oracle):
    assert (out["rgba"] == ref["rgba"]).all(), "uchar4 image differs"
    assert (out["accum"].view(np.uint32) == ref["accum"].view(np.uint32)).all(), "fp32 accumulators differ (bitwise)"
    assert (out["segments_px"] == ref["segments_px"]).all(), "per-pixel segment counts differ"
    assert out["stats"]["segments"] == ref["segments"] and out["stats"]["rng_draws"] == ref["rng_draws"]
    st = out["states"].view(oracle.STATE_DTYPE)
    assert (st["d"] == ref["states"]["d"]).all() and (st["v"] == ref["states"]["v"]).all(), "final RNG states differ"


def test_host_seeding_matches_oracle(oracle):
    for W, H in ((200, 112), (37, 11), (5, 3)):
        got = hip.seed_states_host(69420, W, H, oracle.STATE_DTYPE)
        want = oracle.seed_states(69420, W, H)
        assert (got["d"] == want["d"]).all() and (got["v"] == want["v"]).all()


@pytest.mark.parametrize("tree", [False, True])
@pytest.mark.parametrize("name", sorted(CASES))
def test_host_mode_matches_oracle_and_golden(oracle, name, tree):
    sid, width, spp, depth = CASES[name]
    world, cam = host.build_scene(sid, width=width, spp=spp, depth=depth)
    ref = oracle.render(world, cam, nthreads=8)
    out = hip.render_host(world, cam, nthreads=8, tree=tree)
    same(out, ref, oracle)
    assert (out["rgba"] == GOLD[name + "_rgba"]).all()
    assert (out["accum"].view(np.uint32) == GOLD[name + "_accum"].view(np.uint32)).all()
    walked = "unified tree" in out["stats"]["kernel_name"]
    assert walked == (tree and sid not in (1, 10))  # scenes 1 / 10 are reference BVHs: no unified tree


def test_config1_host_serial(oracle):
    """BASELINE config 1 as stated: Scene 1 200x112, 4 spp, host-side SERIAL loop (one thread)."""
    world, cam = host.build_scene(1, width=200, spp=4)
    out = hip.render_host(world, cam, nthreads=1)
    assert (out["rgba"] == GOLD["s1_c1_rgba"]).all() and out["stats"]["eff_samples"] == 200 * 112 * 4
    assert "1 thread" in out["stats"]["kernel_name"]


def test_host_tree_other_views_and_thread_counts(oracle):
    """Unified-tree walk against the oracle from inside the final scene and the Cornell box; 1 vs 5 threads agree."""
    import ctypes as C
    for sid, frm, at in ((9, (300.0, 250.0, 100.0), (200.0, 200.0, 400.0)), (6, (278.0, 278.0, 100.0), (200.0, 100.0, 555.0)),
                         (7, (100.0, 300.0, 100.0), (100.0, 0.0, 100.001)), (9, (478.0, 278.0, -600.0), (0.0, 278.0, -600.0))):
        world, cam = host.build_scene(sid, width=40, spp=4, depth=6)
        for i in range(3):
            cam.lookfrom.e[i] = frm[i]; cam.lookat.e[i] = at[i]
        host.lib().mort_camera_initialize(C.byref(cam))
        ref = oracle.render(world, cam, nthreads=8)
        a = hip.render_host(world, cam, nthreads=1, tree=True)
        b = hip.render_host(world, cam, nthreads=5, tree=True)
        same(a, ref, oracle)
        same(b, ref, oracle)


@pytest.mark.parametrize("name", sorted(__import__("tests.worlds", fromlist=["x"]).FLAT_WORLDS))
def test_host_tree_on_awkward_flat_worlds(oracle, name):
    """Empty world, single primitives, coincident spheres and quads (equal t: the scan decides), concentric glass,
    instance chains, every material lit by a sphere light (mixture + sphere_pdf), media with a quad light, and a world
    whose list is scanned after a medium (no unified tree: order matters) -- tree walk and item scan against the oracle."""
    from tests.worlds import FLAT_WORLDS, flat_world, flat_camera
    spec = FLAT_WORLDS[name]
    w, ids = flat_world(spec["prims"], media=spec.get("media", ()), late_list=spec.get("late_list", False))
    light = ids[spec["light"][1]] if spec.get("light") else None
    cam = flat_camera(light=light, spp=4, width=72)
    ref = oracle.render(w, cam, nthreads=8)
    for tree in (False, True):
        out = hip.render_host(w, cam, nthreads=8, tree=tree)
        same(out, ref, oracle)
        assert ("unified tree" in out["stats"]["kernel_name"]) == (tree and name != "media_then_list")
        if tree and name == "coincident":
            assert out["stats"]["reference_walks"] == 0  # equal t is resolved in place by scan rank, not by the scan


def test_host_tree_random_views(oracle):
    """60 random cameras in and around the Cornell box, the smoke box and the final scene: tree walk vs oracle."""
    import ctypes as C
    rng = np.random.default_rng(2024)
    for sid in (6, 7, 9):
        world, cam = host.build_scene(sid, width=20, spp=1, depth=10)
        for k in range(20):
            frm, at = rng.uniform(-100, 655, 3), rng.uniform(0, 555, 3)
            for i in range(3):
                cam.lookfrom.e[i] = frm[i]; cam.lookat.e[i] = at[i]
            cam.vfov = int(rng.uniform(20, 100))
            host.lib().mort_camera_initialize(C.byref(cam))
            ref = oracle.render(world, cam, nthreads=8)
            same(hip.render_host(world, cam, nthreads=8, tree=True), ref, oracle)


def test_grazing_hit_from_far_away_regression(monkeypatch):
    """Final scene at 800x800, 100 spp, pixel (659, 358): one ray in 3.6e8 grazes a radius-10 cluster sphere from ~1500
    units away, where sphere::hit's fp32 discriminant accepts a hit the exact line misses by more than round 1's fixed
    box pads -- its per-run trees lost that hit (305 vs 429 segments for the pixel).  The unified tree sizes its pads
    from the sphere test's error bound (scene_compile.h build_unified) and must agree with the reference's scan."""
    monkeypatch.setenv("MORT_HOST_ROWS", "358,359")
    world, cam = host.build_scene(8, width=800, spp=100)
    states = hip.seed_states_host(69420, 800, 800)
    tree = hip.render_host(world, cam, states=states, nthreads=8, tree=True)
    scan = hip.render_host(world, cam, states=states, nthreads=8, tree=False)
    assert "unified tree" in tree["stats"]["kernel_name"] and "item scan" in scan["stats"]["kernel_name"]
    assert tree["segments_px"][358, 659] == scan["segments_px"][358, 659] == 305
    assert (tree["segments_px"][358] == scan["segments_px"][358]).all()
    assert (tree["accum"][358].view(np.uint32) == scan["accum"][358].view(np.uint32)).all()
    assert (tree["states"] == scan["states"]).all()


def _random_world(rng, n_spheres, n_quads, n_boxes, n_media, with_light):
    """A random brute-force world: spheres (some moving, some huge, some tiny), quads, rotated / translated boxes, media in
    sphere boundaries; every material kind.  Returns (world, light (type, idx) or None)."""
    import ctypes as C
    L = host.lib()
    w = host.World()

    palette = []

    def material():
        # the reference's tables are small (20 isotropics, 50 dielectrics, ...): draw from a palette of at most 16
        if len(palette) >= 16:
            return palette[int(rng.integers(0, len(palette)))]
        m = new_material()
        palette.append(m)
        return m

    def new_material():
        k = rng.integers(0, 6)
        col = tuple(rng.uniform(0.1, 0.95, 3))
        if k == 0:
            return S.MAT_METAL, L.mort_add_metal(w.ptr, host.vec3(*col), float(rng.uniform(0, 1)))
        if k == 1:
            return S.MAT_DIELECTRIC, L.mort_add_dielectric(w.ptr, float(rng.uniform(1.1, 2.0)))
        if k == 2:
            c = L.mort_add_solid_color(w.ptr, host.vec3(*col))
            return S.MAT_ISOTROPIC, L.mort_add_isotropic(w.ptr, S.TEXTURE_SOLID, c)
        if k == 3:
            c1 = L.mort_add_solid_color(w.ptr, host.vec3(*col)); c2 = L.mort_add_solid_color(w.ptr, host.vec3(.9, .9, .9))
            ck = L.mort_add_checker_texture(w.ptr, float(rng.uniform(0.2, 2.0)), S.TEXTURE_SOLID, c1, S.TEXTURE_SOLID, c2)
            return S.MAT_LAMBERTIAN, L.mort_add_lambertian(w.ptr, S.TEXTURE_CHECKER, ck)
        c = L.mort_add_solid_color(w.ptr, host.vec3(*col))
        return S.MAT_LAMBERTIAN, L.mort_add_lambertian(w.ptr, S.TEXTURE_SOLID, c)

    light = None
    if with_light:
        c = L.mort_add_solid_color(w.ptr, host.vec3(7, 7, 7))
        lm = L.mort_add_diffuse_light(w.ptr, S.TEXTURE_SOLID, c)
        q = L.mort_add_quad(w.ptr, host.vec3(-2, 6, -2), host.vec3(4, 0, 0), host.vec3(0, 0, 4), S.MAT_DIFFUSE_LIGHT, lm, False)
        light = (S.OBJ_QUAD, q)
    mt, mi = material()
    L.mort_add_sphere(w.ptr, host.vec3(0, -1000, 0), 1000.0, mt, mi, False)  # ground: a huge sphere among small ones
    for _ in range(n_spheres):
        mt, mi = material()
        c = rng.uniform(-6, 6, 3) * (1, 0.3, 1) + (0, 1.2, 0)
        r = float(rng.choice([0.05, 0.2, 0.5, 1.0, 1.5]))
        if rng.random() < 0.3:
            L.mort_add_moving_sphere(w.ptr, host.vec3(*c), host.vec3(*(c + rng.uniform(-0.5, 0.5, 3))), r, mt, mi, False)
        else:
            L.mort_add_sphere(w.ptr, host.vec3(*c), r, mt, mi, False)
    for _ in range(n_quads):
        mt, mi = material()
        L.mort_add_quad(w.ptr, host.vec3(*rng.uniform(-5, 5, 3)), host.vec3(*rng.uniform(-2, 2, 3)), host.vec3(*rng.uniform(-2, 2, 3)), mt, mi, False)
    for _ in range(n_boxes):
        mt, mi = material()
        L.mort_rotated_box(w.ptr, host.vec3(*rng.uniform(0.3, 2.0, 3)), host.vec3(*(rng.uniform(-5, 5, 3) * (1, 0, 1))), float(rng.uniform(-60, 60)), mt, mi)
    for _ in range(n_media):
        b = L.mort_add_sphere(w.ptr, host.vec3(*(rng.uniform(-4, 4, 3) * (1, 0.2, 1) + (0, 1, 0))), float(rng.uniform(0.5, 2.5)), S.MAT_DIELECTRIC, L.mort_add_dielectric(w.ptr, 1.5), True)
        c = L.mort_add_solid_color(w.ptr, host.vec3(*rng.uniform(0.2, 1.0, 3)))
        L.mort_add_constant_medium(w.ptr, S.OBJ_SPHERE, b, float(rng.uniform(0.05, 3.0)), S.MAT_ISOTROPIC, L.mort_add_isotropic(w.ptr, S.TEXTURE_SOLID, c), False)
    w.c.bvh_mode = False
    return w, light


@pytest.mark.parametrize("seed", range(40))
def test_unified_tree_equals_scan_on_random_worlds(oracle, seed):
    """Random worlds x random cameras (inside, outside, far away, grazing the ground): the unified-tree walk, the
    reference's scan (both in the product's host loop) and the oracle agree bit for bit."""
    import ctypes as C
    from tests.worlds import set_view
    rng = np.random.default_rng(1000 + seed)
    w, light = _random_world(rng, n_spheres=int(rng.integers(1, 60)), n_quads=int(rng.integers(0, 12)), n_boxes=int(rng.integers(0, 3)),
                             n_media=int(rng.integers(0, 3)), with_light=bool(seed % 2))
    _, cam = host.build_scene(2, width=56, spp=4, depth=int(rng.integers(2, 20)))
    for i in range(3):
        cam.background.e[i] = float(rng.uniform(0.0, 0.8)) * (0 if light and seed % 4 == 1 else 1)
    if light:
        cam.light_obj_type, cam.light_obj_idx = light
    views = [((0, 2, 9), (0, 1, 0)), ((0.3, 0.05, 0.2), (4, 0.3, 1)), ((40, 25, -60), (0, 0, 0)), (tuple(rng.uniform(-5, 5, 3) + (0, 6, 0)), tuple(rng.uniform(-2, 2, 3)))]
    for frm, at in views:
        set_view(cam, frm, at, vfov=int(rng.integers(20, 90)), defocus=float(rng.choice([0.0, 0.0, 0.8])))
        ref = oracle.render(w, cam, nthreads=8)
        tree = hip.render_host(w, cam, nthreads=8, tree=True)
        scan = hip.render_host(w, cam, nthreads=8, tree=False)
        assert "unified tree" in tree["stats"]["kernel_name"]
        same(tree, ref, oracle)
        same(scan, ref, oracle)
