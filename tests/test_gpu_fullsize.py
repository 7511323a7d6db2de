"""BASELINE configs 3 and 4/5's scene at their STATED size, spp and depth (Cornell 800x800 x 1000 spp, depth 50; book-2 final scene
800x800 x 1000 spp, depth 40; final scene 1920x1080 at depth 40), where the CPU oracle cannot follow in a test: size-independent
properties instead -- different kernels (different traversals of different trees, different scheduling, different workgroup shapes)
must agree BIT FOR BIT on image, fp32 accumulators, per-pixel segment counts and final XORWOW words, and a render must repeat
itself.  The same kernels are compared with the oracle at small sizes in test_gpu_gen.py / test_gpu_parity.py; round 1's wrong ray
showed up at 800x800 x 100 spp and at no test size, which is what these cases are for.  About 40 s of GPU time in all."""
import numpy as np
import pytest

from mort_amd import host, hip, structs as S

pytestmark = pytest.mark.gpu


def _render(world, cam, oracle, monkeypatch, mode=hip.MODE_MEGA, env=None, nranks=1, want_accum=True):
    W, H = cam.image_width, cam.image_height
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    rgba = np.zeros((H, W, 4), np.uint8)
    acc = np.zeros((H, W, 3), np.float32)
    seg = np.zeros((H, W), np.uint32)
    states, total, name = None, 0, None
    with hip.Context(0) as ctx:
        for r in range(nranks):
            ctx.set_partition(r, nranks, 8)
            ctx.upload_world(world)
            ctx.rng_seed(S.DEFAULT_SEED, W, H)
            o = ctx.render(cam, mode=mode, want_accum=want_accum, want_segments=True)
            rows = [ctx.global_row(l) for l in range(ctx.local_rows(H))]
            rgba[rows] = o["rgba"][rows]
            seg[rows] = o["segments_px"][rows]
            if want_accum:
                acc[rows] = o["accum"][rows]
            total += o["stats"]["segments"]
            name = o["stats"]["kernel_name"]
            st = ctx.rng_store(W, H, oracle.STATE_DTYPE).reshape(H, W)
            states = st.copy() if states is None else states
            states[rows] = st[rows]
    for k in (env or {}):
        monkeypatch.delenv(k, raising=False)
    return dict(rgba=rgba, acc=acc, seg=seg, total=total, name=name, states=states)


def _same(a, b):
    assert (a["rgba"] == b["rgba"]).all(), f"{a['name']} vs {b['name']}: {int((a['rgba'] != b['rgba']).any(axis=-1).sum())} pixels differ"
    assert (a["acc"].view(np.uint32) == b["acc"].view(np.uint32)).all(), f"{a['name']} vs {b['name']}: accumulators"
    assert (a["seg"] == b["seg"]).all() and a["total"] == b["total"] == int(a["seg"].sum(dtype=np.uint64)), f"{a['name']} vs {b['name']}: segments"
    assert (a["states"]["d"] == b["states"]["d"]).all() and (a["states"]["v"] == b["states"]["v"]).all(), f"{a['name']} vs {b['name']}: streams"


def test_config3_cornell_800x800_1000spp(oracle, monkeypatch):
    """BASELINE config 3 as stated (`mort 6 --width 800`: 961 effective spp, depth 50): the default kernel (one lane per pixel, the
    reference's scan) == itself again == the unified-tree megakernel == the wavefront pipeline."""
    world, cam = host.build_scene(6, width=800, spp=1000)
    assert (cam.image_width, cam.image_height, cam.sqrt_spp, cam.bounce_limit) == (800, 800, 31, 50)
    a = _render(world, cam, oracle, monkeypatch)
    assert a["name"] == "mega_kernel"
    assert (a["rgba"][..., 3] == 255).all() and a["total"] > 961 * 640000
    _same(a, _render(world, cam, oracle, monkeypatch))
    g = _render(world, cam, oracle, monkeypatch, env={"MORT_GEN_MIN_PRIMS": "0"})
    assert g["name"].startswith("mega_gen_kernel")
    _same(a, g)
    w = _render(world, cam, oracle, monkeypatch, mode=hip.MODE_WAVE)
    assert w["name"].startswith("wf_trav_gen")
    _same(a, w)


def test_final_scene_800x800_1000spp(oracle, monkeypatch):
    """The scene of configs 4 / 5 at its catalogue size and spp (`mort 8`: 800x800, 961 effective spp, depth 40): the unified-tree
    megakernel == itself again == a two-way row partition of it == the wavefront pipeline, 3.4 G segments each."""
    world, cam = host.build_scene(8, width=800, spp=1000)
    assert (cam.image_width, cam.image_height, cam.sqrt_spp, cam.bounce_limit) == (800, 800, 31, 40)
    a = _render(world, cam, oracle, monkeypatch)
    assert a["name"].startswith("mega_gen_kernel")
    assert (a["rgba"][..., 3] == 255).all() and a["total"] > 3 * 10**9
    _same(a, _render(world, cam, oracle, monkeypatch))
    _same(a, _render(world, cam, oracle, monkeypatch, nranks=2))
    w = _render(world, cam, oracle, monkeypatch, mode=hip.MODE_WAVE)
    assert w["name"].startswith("wf_trav_gen")
    _same(a, w)


def test_headline_scene1_1200x675_500spp(oracle, monkeypatch):
    """BASELINE config 2 (the headline) at its stated size: the 1024-thread BVH megakernel the launch picks (128 registers, four waves per
    SIMD, some per-pixel values in private memory) == the 768-thread build without private memory == a four-way partition (512-thread
    drain kernels, one context per rank) == the wavefront pipeline, bit for bit."""
    world, cam = host.build_scene(1, width=1200, spp=500)
    assert (cam.image_width, cam.image_height, cam.bounce_limit) == (1200, 675, 20)
    a = _render(world, cam, oracle, monkeypatch)
    assert a["name"].startswith("mega_bvh_kernel<1024"), a["name"]
    b = _render(world, cam, oracle, monkeypatch, env={"MORT_FAST_BLOCK_SIZE": "768"})
    assert b["name"].startswith("mega_bvh_kernel<768"), b["name"]
    _same(a, b)
    _same(a, _render(world, cam, oracle, monkeypatch, nranks=4))
    _same(a, _render(world, cam, oracle, monkeypatch, mode=hip.MODE_WAVE))


@pytest.mark.parametrize("block", ["1024", "768", "256"])
def test_final_scene_workgroup_shapes_fill_the_chip(oracle, monkeypatch, block):
    """Every workgroup shape of the unified-tree megakernel on a frame that gives every CU several workgroups and reaches the deep
    (HBM) levels of the bounce stack: 800x800 x 16 spp of the final scene, >= 830 workgroups; the default shape is the reference."""
    world, cam = host.build_scene(8, width=800, spp=16)
    a = _render(world, cam, oracle, monkeypatch)
    b = _render(world, cam, oracle, monkeypatch, env={"MORT_GEN_BLOCK_SIZE": block})
    assert b["name"].startswith(f"mega_gen_kernel<{block}"), b["name"]
    _same(a, b)


@pytest.mark.parametrize("heavy", ["3,2,12,50", "1,1,4,100", "4,1,8,75", "2,1,1,30"])
def test_final_scene_heavy_waves_do_not_reach_the_pixels(oracle, monkeypatch, heavy):
    """Heavy waves (mega_bvh.h FastArgs.heavy_*: some waves take a few lanes' worth of pixels from the head of the cost order, the others start
    behind it; which tiles form the head is decided on the device) are scheduling only: the final scene 800x800 x 16 spp with them forced in
    four shapes -- and a second frame, whose head comes from the first frame's costs instead of the probe's -- equals the launch without them."""
    world, cam = host.build_scene(8, width=800, spp=16)
    a = _render(world, cam, oracle, monkeypatch, env={"MORT_GEN_NO_HEAVY": "1"})
    b = _render(world, cam, oracle, monkeypatch, env={"MORT_GEN_HEAVY": heavy, "MORT_GEN_BLOCK_SIZE": "1024"})
    assert b["name"].startswith("mega_gen_kernel<1024"), b["name"]
    _same(a, b)
    W, H = cam.image_width, cam.image_height
    for k, v in {"MORT_GEN_HEAVY": heavy, "MORT_GEN_BLOCK_SIZE": "1024"}.items():
        monkeypatch.setenv(k, v)
    with hip.Context(0) as ctx:  # two frames on one context: the streams continue, the second frame's order and head come from the first one's costs
        ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, W, H)
        f1 = ctx.render(cam, want_accum=False, want_segments=True)
        f2 = ctx.render(cam, want_accum=False, want_segments=True)
    monkeypatch.delenv("MORT_GEN_HEAVY"); monkeypatch.delenv("MORT_GEN_BLOCK_SIZE")
    monkeypatch.setenv("MORT_GEN_NO_HEAVY", "1")
    with hip.Context(0) as ctx:
        ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, W, H)
        g1 = ctx.render(cam, want_accum=False, want_segments=True)
        g2 = ctx.render(cam, want_accum=False, want_segments=True)
    monkeypatch.delenv("MORT_GEN_NO_HEAVY")
    assert (f1["rgba"] == g1["rgba"]).all() and (f2["rgba"] == g2["rgba"]).all()
    assert (f2["segments_px"] == g2["segments_px"]).all() and f2["stats"]["segments"] == g2["stats"]["segments"]


def test_config4_geometry_depth40(oracle, monkeypatch):
    """BASELINE config 4's frame (final scene 1920x1080, depth 40) at 4 spp: megakernel == two-way partition == wavefront pipeline."""
    world, cam = host.build_scene(8, width=1920, spp=4, aspect=16.0 / 9.0)
    assert (cam.image_width, cam.image_height, cam.bounce_limit) == (1920, 1080, 40)
    a = _render(world, cam, oracle, monkeypatch)
    assert a["name"].startswith("mega_gen_kernel")
    _same(a, _render(world, cam, oracle, monkeypatch, nranks=2))
    _same(a, _render(world, cam, oracle, monkeypatch, mode=hip.MODE_WAVE))
