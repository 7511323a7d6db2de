"""Parity tests of the unified-tree megakernel (mort_amd/csrc/hip/mega_gen.hip): worlds WITHOUT reference BVHs
-- reference scenes 2..9 and hand-built worlds -- through the C ABI, against the CPU oracle, BIT-EXACT (uchar4
image, fp32 accumulator bits, per-pixel segment counts, final XORWOW words), plus cross-checks against the
one-lane-per-pixel kernel (mega_kernel) at sizes the oracle cannot reach in a test.  Parity against the CUDA
reference itself is unpinned."""
import ctypes as C
import os

import numpy as np
import pytest

from mort_amd import host, hip, structs as S
from tests.test_gpu_parity import render_gpu, assert_same
from tests.worlds import FLAT_WORLDS, flat_world as _flat_world, flat_camera as _flat_camera, set_view as _set_view

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tree_kernel_for_every_world(monkeypatch):
    """By default worlds with fewer than 48 solid primitives stay on the one-lane-per-pixel kernel (it is faster there);
    these tests exercise the unified-tree kernel on all of them."""
    monkeypatch.setenv("MORT_GEN_MIN_PRIMS", "0")


def test_default_kernel_choice(gpu_ctx, oracle, monkeypatch):
    monkeypatch.delenv("MORT_GEN_MIN_PRIMS", raising=False)
    for sid, want in ((6, "mega_kernel"), (9, "mega_gen_kernel"), (1, "mega_bvh_kernel")):
        world, cam = host.build_scene(sid, width=48, spp=1)
        out = render_gpu(gpu_ctx, world, cam, oracle=oracle)
        assert out["stats"]["kernel_name"].startswith(want)
        assert_same(out, oracle.render(world, cam, nthreads=8))


@pytest.mark.parametrize("sid,width,spp,depth", [(2, 120, 4, None), (3, 120, 4, None), (4, 96, 4, None), (5, 96, 9, None),
                                                 (6, 96, 16, None), (7, 64, 9, None), (8, 72, 4, None), (9, 96, 9, None), (8, 64, 4, 6)])
def test_unified_tree_kernel_matches_oracle(gpu_ctx, oracle, sid, width, spp, depth):
    world, cam = host.build_scene(sid, width=width, spp=spp, depth=depth)
    ref = oracle.render(world, cam, nthreads=16)
    out = render_gpu(gpu_ctx, world, cam, oracle=oracle)
    assert out["stats"]["kernel_name"].startswith("mega_gen_kernel"), out["stats"]["kernel_name"]
    assert out["stats"]["scene_in_lds"] == 1
    assert_same(out, ref)


@pytest.mark.parametrize("sid,width,spp", [(6, 64, 9), (7, 48, 9), (9, 64, 4)])
def test_one_lane_per_pixel_kernel_still_matches(oracle, monkeypatch, sid, width, spp):
    """mega_kernel (no unified tree) stays the kernel for worlds the tree does not cover and is the body of the
    host loop (--mode host): it must keep giving the oracle's bits."""
    monkeypatch.setenv("MORT_NO_GEN", "1")
    world, cam = host.build_scene(sid, width=width, spp=spp)
    ref = oracle.render(world, cam, nthreads=16)
    with hip.Context(0) as ctx:
        out = render_gpu(ctx, world, cam, oracle=oracle)
    assert out["stats"]["kernel_name"] == "mega_kernel"
    assert_same(out, ref)


@pytest.mark.parametrize("env", [{"MORT_GEN_BLOCK_SIZE": "768"}, {"MORT_GEN_BLOCK_SIZE": "512"}, {"MORT_GEN_BLOCK_SIZE": "256"},
                                 {"MORT_GEN_THRESHOLDS": "2,2,2,2"}, {"MORT_GEN_THRESHOLDS": "64,64,64,64"}, {"MORT_NO_TILE_ORDER": "1"}])
@pytest.mark.parametrize("sid,width,spp", [(6, 200, 4), (9, 160, 4)])
def test_gen_scheduling_choices_do_not_reach_the_pixels(gpu_ctx, oracle, monkeypatch, env, sid, width, spp):
    """Workgroup shape, batch thresholds and tile order of the state machine must not change a bit; two frames, the
    second ordered by the first one's costs."""
    world, cam = host.build_scene(sid, width=width, spp=spp)
    ref1 = oracle.render(world, cam, nthreads=16)
    ref2 = oracle.render(world, cam, nthreads=16, states=ref1["states"].copy())
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    out1 = render_gpu(gpu_ctx, world, cam, oracle=oracle)
    assert out1["stats"]["kernel_name"].startswith("mega_gen_kernel")
    assert_same(out1, ref1)
    out2 = gpu_ctx.render(cam, want_accum=True, want_segments=True)
    out2["states"] = gpu_ctx.rng_store(cam.image_width, cam.image_height, oracle.STATE_DTYPE)
    assert_same(out2, ref2)


@pytest.mark.parametrize("sid", [6, 7, 9])
def test_gen_other_viewpoints(gpu_ctx, oracle, sid):
    """Cameras the built-in scenes never use: inside the box / the sphere cluster, at floor level, looking along an
    axis (zero direction components: the scan decides), from far outside (beyond the reach the tree's pads were sized
    for: the one-lane-per-pixel kernel takes over), with and without defocus."""
    world, cam = host.build_scene(sid, width=72, spp=4, depth=8)
    rng = np.random.default_rng(11 + sid)
    views = [((278, 278, 100), (278, 278, 555)), ((278, 1.0, 278), (300, 1.0, 0)), ((100, 300, 100), (100, 0, 100.001)),
             ((278, 278, -800), (278, 278, 0)), ((-3e5, 4e5, -9e5), (278, 278, 0)), ((130, 60, 200), (400, 200, 300))]
    views += [(tuple(rng.uniform(0, 555, 3)), tuple(rng.uniform(0, 555, 3))) for _ in range(4)]
    kernels = set()
    for k, (frm, at) in enumerate(views):
        _set_view(cam, frm, at, vfov=40 if k % 3 else 80, defocus=0.0 if k % 2 else 0.5)
        ref = oracle.render(world, cam, nthreads=16)
        out = render_gpu(gpu_ctx, world, cam, oracle=oracle)
        kernels.add(out["stats"]["kernel_name"].split("<")[0])
        assert_same(out, ref)
    assert kernels == {"mega_gen_kernel", "mega_kernel"}  # the far camera falls outside the tree's reach


@pytest.mark.parametrize("name", sorted(FLAT_WORLDS))
def test_small_and_awkward_flat_worlds(gpu_ctx, oracle, name):
    spec = FLAT_WORLDS[name]
    w, ids = _flat_world(spec["prims"], media=spec.get("media", ()), late_list=spec.get("late_list", False))
    light = None
    if spec.get("light"):
        light = ids[spec["light"][1]]
    cam = _flat_camera(light=light)
    ref = oracle.render(w, cam, nthreads=16)
    out = render_gpu(gpu_ctx, w, cam, oracle=oracle)
    assert_same(out, ref)
    want = "mega_kernel" if name == "media_then_list" else "mega_gen_kernel"
    assert out["stats"]["kernel_name"].startswith(want), out["stats"]["kernel_name"]
    if name == "coincident":
        assert out["stats"]["reference_walks"] == 0  # equal t is resolved in place by scan rank


@pytest.mark.parametrize("sid,width,aspect,spp,depth", [(6, 800, None, 4, None), (8, 800, None, 1, None), (8, 1920, 16.0 / 9.0, 1, 12)])
def test_full_size_frames_agree_between_kernels(oracle, monkeypatch, sid, width, aspect, spp, depth):
    """BASELINE configs 3 / 4 geometry (Cornell 800x800; final scene 800x800 and 1920x1080) at low spp: too big for the
    oracle in a test, so (a) the unified-tree kernel and the one-lane-per-pixel kernel -- different traversals of
    different trees -- must agree bit for bit on image, accumulators, per-pixel segment counts and final streams,
    (b) run-to-run determinism, (c) a two-way row partition composes to the same frame with the same segment total."""
    world, cam = host.build_scene(sid, width=width, spp=spp, depth=depth, aspect=aspect)
    W, H = cam.image_width, cam.image_height

    def run(nranks=1):
        rgba = np.zeros((H, W, 4), np.uint8); acc = np.zeros((H, W, 3), np.float32); seg = np.zeros((H, W), np.uint32)
        states = None
        total = 0
        name = None
        with hip.Context(0) as ctx:
            for r in range(nranks):
                ctx.set_partition(r, nranks, 8)
                ctx.upload_world(world)
                ctx.rng_seed(S.DEFAULT_SEED, W, H)
                o = ctx.render(cam, want_accum=True, want_segments=True)
                rows = [ctx.global_row(l) for l in range(ctx.local_rows(H))]
                rgba[rows] = o["rgba"][rows]; acc[rows] = o["accum"][rows]; seg[rows] = o["segments_px"][rows]
                total += o["stats"]["segments"]
                name = o["stats"]["kernel_name"]
                st = ctx.rng_store(W, H, oracle.STATE_DTYPE).reshape(H, W)
                states = st.copy() if states is None else states
                states[rows] = st[rows]
        return dict(rgba=rgba, acc=acc, seg=seg, total=total, name=name, states=states)

    a = run()
    b = run()
    c = run(nranks=2)
    monkeypatch.setenv("MORT_NO_GEN", "1")
    d = run()
    assert a["name"].startswith("mega_gen_kernel") and d["name"] == "mega_kernel"
    for other in (b, c, d):
        assert (a["rgba"] == other["rgba"]).all()
        assert (a["acc"].view(np.uint32) == other["acc"].view(np.uint32)).all()
        assert (a["seg"] == other["seg"]).all() and a["total"] == other["total"] == int(a["seg"].sum())
        assert (a["states"]["d"] == other["states"]["d"]).all() and (a["states"]["v"] == other["states"]["v"]).all()
    assert (a["rgba"][..., 3] == 255).all()


@pytest.mark.parametrize("sid,width,spp,depth", [(2, 96, 4, None), (3, 96, 4, None), (4, 64, 4, None), (5, 64, 9, None), (6, 96, 16, None),
                                                 (7, 64, 9, None), (8, 72, 4, None), (9, 96, 9, None), (8, 64, 4, 6), (6, 61, 5, 3)])
def test_wavefront_mode_on_unified_tree_worlds(gpu_ctx, oracle, sid, width, spp, depth):
    """MORT_MODE_WAVE (wave_gen.hip) on every non-BVH scene -- BASELINE config 5 is scene 8 through this mode -- gives the
    oracle's bits: lights / MIS, quads, instance chains, media (their stream draws happen in the shade kernel)."""
    world, cam = host.build_scene(sid, width=width, spp=spp, depth=depth)
    ref = oracle.render(world, cam, nthreads=16)
    gpu_ctx.set_partition(0, 1, 8)
    gpu_ctx.upload_world(world)
    gpu_ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    out = gpu_ctx.render(cam, mode=hip.MODE_WAVE, want_accum=True, want_segments=True)
    out["states"] = gpu_ctx.rng_store(cam.image_width, cam.image_height, oracle.STATE_DTYPE)
    assert out["stats"]["kernel_name"].startswith("wf_trav_gen")
    assert_same(out, ref)


@pytest.mark.parametrize("name", ["coincident", "boxes_and_instances", "every_material_lit_by_sphere", "lit_by_quad_with_media", "empty"])
def test_wavefront_mode_on_awkward_flat_worlds(gpu_ctx, oracle, name):
    spec = FLAT_WORLDS[name]
    w, ids = _flat_world(spec["prims"], media=spec.get("media", ()))
    cam = _flat_camera(light=ids[spec["light"][1]] if spec.get("light") else None, spp=4, width=96)
    ref = oracle.render(w, cam, nthreads=16)
    gpu_ctx.set_partition(0, 1, 8)
    gpu_ctx.upload_world(w)
    gpu_ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    out = gpu_ctx.render(cam, mode=hip.MODE_WAVE, want_accum=True, want_segments=True)
    out["states"] = gpu_ctx.rng_store(cam.image_width, cam.image_height, oracle.STATE_DTYPE)
    assert_same(out, ref)


def test_wavefront_and_megakernel_agree_at_config_size(oracle):
    """Final scene at 800x800 (config 5's scene; its 4096x4096 x 10000 spp is an 8-GPU job), 1 spp: the wavefront form and
    the unified-tree megakernel agree on every byte, accumulator bit, segment count and final stream word; a two-way row
    partition of the wavefront render composes to the same frame."""
    world, cam = host.build_scene(8, width=800, spp=1)
    W, H = cam.image_width, cam.image_height
    outs = []
    for mode, nranks in ((hip.MODE_MEGA, 1), (hip.MODE_WAVE, 1), (hip.MODE_WAVE, 2)):
        rgba = np.zeros((H, W, 4), np.uint8); acc = np.zeros((H, W, 3), np.float32); seg = np.zeros((H, W), np.uint32)
        total = 0
        states = None
        with hip.Context(0) as ctx:
            for r in range(nranks):
                ctx.set_partition(r, nranks, 8)
                ctx.upload_world(world)
                ctx.rng_seed(S.DEFAULT_SEED, W, H)
                o = ctx.render(cam, mode=mode, want_accum=True, want_segments=True)
                rows = [ctx.global_row(l) for l in range(ctx.local_rows(H))]
                rgba[rows] = o["rgba"][rows]; acc[rows] = o["accum"][rows]; seg[rows] = o["segments_px"][rows]
                total += o["stats"]["segments"]
                st = ctx.rng_store(W, H, oracle.STATE_DTYPE).reshape(H, W)
                states = st.copy() if states is None else states
                states[rows] = st[rows]
        outs.append((rgba, acc, seg, total, states))
    a = outs[0]
    for b in outs[1:]:
        assert (a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all() and (a[2] == b[2]).all() and a[3] == b[3]
        assert (a[4]["d"] == b[4]["d"]).all() and (a[4]["v"] == b[4]["v"]).all()


def test_config5_geometry_wavefront_equals_megakernel():
    """BASELINE config 5's exact frame geometry and mode -- book-2 final scene, 4096x4096, wavefront kernels -- at 1 spp on
    one GPU (the config's 10 000 spp on 8 GPUs is the same frame 10 000 times over an 8-way row partition): 16.8 M paths
    through wave_gen.hip agree with the unified-tree megakernel on every byte, per-pixel segment count and final stream."""
    world, cam = host.build_scene(8, width=4096, spp=1, aspect=1.0)
    W, H = cam.image_width, cam.image_height
    assert (W, H) == (4096, 4096)
    res = []
    for mode in (hip.MODE_MEGA, hip.MODE_WAVE):
        with hip.Context(0) as ctx:
            ctx.upload_world(world)
            ctx.rng_seed(S.DEFAULT_SEED, W, H)
            o = ctx.render(cam, mode=mode, want_accum=False, want_segments=True)
            st = ctx.rng_store(W, H)
        res.append((o["rgba"], o["segments_px"], o["stats"]["segments"], st, o["stats"]["kernel_name"]))
    a, b = res
    assert a[4].startswith("mega_gen_kernel") and b[4].startswith("wf_trav_gen")
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2] == b[2] == int(a[1].sum(dtype=np.uint64))
    assert (a[3] == b[3]).all()
    assert (a[0][..., 3] == 255).all()


@pytest.mark.parametrize("seed", range(8))
def test_random_worlds_on_the_gpu(gpu_ctx, oracle, seed):
    """The CPU fuzz's random worlds (tests/test_host_mode.py) through the unified-tree megakernel and the wavefront mode."""
    from tests.test_host_mode import _random_world
    rng = np.random.default_rng(5000 + seed)
    w, light = _random_world(rng, n_spheres=int(rng.integers(1, 60)), n_quads=int(rng.integers(0, 12)), n_boxes=int(rng.integers(0, 3)),
                             n_media=int(rng.integers(0, 3)), with_light=bool(seed % 2))
    _, cam = host.build_scene(2, width=96, spp=4, depth=int(rng.integers(2, 20)))
    for i in range(3):
        cam.background.e[i] = float(rng.uniform(0.0, 0.8))
    if light:
        cam.light_obj_type, cam.light_obj_idx = light
    for frm, at in (((0, 2, 9), (0, 1, 0)), ((0.3, 0.05, 0.2), (4, 0.3, 1)), (tuple(rng.uniform(-5, 5, 3) + (0, 6, 0)), tuple(rng.uniform(-2, 2, 3)))):
        _set_view(cam, frm, at, vfov=int(rng.integers(20, 90)), defocus=float(rng.choice([0.0, 0.8])))
        ref = oracle.render(w, cam, nthreads=16)
        out = render_gpu(gpu_ctx, w, cam, oracle=oracle)
        assert out["stats"]["kernel_name"].startswith("mega_gen_kernel")
        assert_same(out, ref)
        gpu_ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
        wv = gpu_ctx.render(cam, mode=hip.MODE_WAVE, want_accum=True, want_segments=True)
        wv["states"] = gpu_ctx.rng_store(cam.image_width, cam.image_height, oracle.STATE_DTYPE)
        assert_same(wv, ref)
