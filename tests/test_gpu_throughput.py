"""MORT_MODE_THROUGHPUT -- the labelled NON-PARITY mode (include/mort_hip.h): one XORWOW stream per (pixel, stratum row) instead
of the reference's one per pixel.  It is not compared with the reference's images; what is checked here is that it computes exactly
what it says: the same per-sample arithmetic as the oracle (get_ray + ray_color, the oracle's known-answer entry points) on the
stream with subsequence x + (y*sqrt_spp + s_j)*W, rows summed in order -- bit for bit -- plus determinism, partition invariance,
frame-to-frame stream carry-over, and statistical agreement with the parity render."""
import ctypes as C

import numpy as np
import pytest

from mort_amd import host, hip, structs as S

pytestmark = pytest.mark.gpu


def expected_substream(oracle, world, cam, seed, frames=1):
    """The mode's definition, from the oracle's primitives (pure-Python loop: small images only)."""
    L = oracle.lib()
    W, H, n = cam.image_width, cam.image_height, cam.sqrt_spp
    st = oracle.seed_states(seed, W, H * n)  # subsequence x + vy*W with vy = y*n + s_j
    ray7 = (C.c_float * 7)()
    rgb = (C.c_float * 3)()
    for _ in range(frames):
        acc = np.zeros((H, W, 3), np.float32)
        for y in range(H):
            for x in range(W):
                px = np.zeros(3, np.float32)
                for j in range(n):
                    sp = C.cast(st[x + (y * n + j) * W:].ctypes.data, C.POINTER(S.RngState))
                    row = np.zeros(3, np.float32)
                    for i in range(n):
                        L.mort_oracle_get_ray(C.byref(cam), x, y, i, j, sp, ray7)
                        L.mort_oracle_ray_color(world.ptr, C.byref(cam), ray7, sp, rgb)
                        row = row + np.array(rgb[:], np.float32)
                    px = px + row
                px = np.float32(cam.pixel_samples_scale) * px
                px[px != px] = 0
                acc[y, x] = px
    g = np.sqrt(acc.astype(np.float64)).astype(np.float32)  # correctly rounded, like mort_sqrtf
    rgba = np.zeros((H, W, 4), np.uint8)
    rgba[..., :3] = (np.float32(256) * np.clip(g, np.float32(0), np.float32(0.999))).astype(np.int32)
    rgba[..., 3] = 255
    return acc, rgba


def render_tp(ctx, world, cam, seed=S.DEFAULT_SEED, part=(0, 1, 8)):
    ctx.set_partition(*part)
    ctx.upload_world(world)
    ctx.rng_seed(seed, cam.image_width, cam.image_height)
    return ctx.render(cam, mode=hip.MODE_THROUGHPUT, want_accum=True)


@pytest.mark.parametrize("sid,width,spp,depth", [(1, 24, 9, 8), (10, 20, 4, 6), (1, 17, 16, 5), (9, 20, 4, 6), (8, 16, 9, 5)])
def test_substream_render_is_the_oracle_arithmetic_on_the_substreams(gpu_ctx, oracle, sid, width, spp, depth):
    world, cam = host.build_scene(sid, width=width, spp=spp, depth=depth)
    out = render_tp(gpu_ctx, world, cam)
    assert out["stats"]["kernel_name"].startswith("mega_bvh_kernel" if sid in (1, 10) else "mega_gen_kernel") and out["stats"]["kernel_name"].endswith(", true>")
    acc, rgba = expected_substream(oracle, world, cam, S.DEFAULT_SEED)
    assert (out["accum"].view(np.uint32) == acc.view(np.uint32)).all()
    assert (out["rgba"] == rgba).all()


def test_streams_carry_over_between_frames(gpu_ctx, oracle):
    world, cam = host.build_scene(1, width=16, spp=4, depth=6)
    render_tp(gpu_ctx, world, cam)
    out2 = gpu_ctx.render(cam, mode=hip.MODE_THROUGHPUT, want_accum=True)  # second frame: the streams continue
    acc, rgba = expected_substream(oracle, world, cam, S.DEFAULT_SEED, frames=2)
    assert (out2["accum"].view(np.uint32) == acc.view(np.uint32)).all() and (out2["rgba"] == rgba).all()


def test_one_sample_per_pixel_is_the_parity_render(gpu_ctx, oracle):
    """sqrt_spp = 1: (pixel, stratum row) = pixel, subsequence x + y*W -- the reference's keying, so the oracle's image."""
    world, cam = host.build_scene(1, width=160, spp=1, depth=12)
    out = render_tp(gpu_ctx, world, cam)
    ref = oracle.render(world, cam, nthreads=8)
    assert (out["rgba"] == ref["rgba"]).all() and (out["accum"].view(np.uint32) == ref["accum"].view(np.uint32)).all()


def test_leaves_the_per_pixel_states_alone_and_is_deterministic(gpu_ctx, oracle):
    world, cam = host.build_scene(1, width=120, spp=16, depth=10)
    a = render_tp(gpu_ctx, world, cam)
    st = gpu_ctx.rng_store(cam.image_width, cam.image_height, oracle.STATE_DTYPE)
    want = oracle.seed_states(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    assert (st["d"] == want["d"]).all() and (st["v"] == want["v"]).all()
    b = render_tp(gpu_ctx, world, cam)
    assert (a["rgba"] == b["rgba"]).all() and (a["accum"].view(np.uint32) == b["accum"].view(np.uint32)).all()
    # and the parity mode still renders the oracle's image afterwards (the sub-streams are separate state)
    gpu_ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    m = gpu_ctx.render(cam, want_accum=True)
    ref = oracle.render(world, cam, nthreads=8)
    assert (m["rgba"] == ref["rgba"]).all()


@pytest.mark.parametrize("sid,nranks", [(1, 2), (1, 3), (9, 2)])
def test_partition_invariance(gpu_ctx, sid, nranks):
    world, cam = host.build_scene(sid, width=96, spp=9, depth=8)
    H = cam.image_height
    whole = render_tp(gpu_ctx, world, cam)
    got = np.zeros_like(whole["accum"])
    for r in range(nranks):
        out = render_tp(gpu_ctx, world, cam, part=(r, nranks, 8))
        rows = [y for y in range(H) if (y // 8) % nranks == r]
        got[rows] = out["accum"][rows]
    gpu_ctx.set_partition(0, 1, 8)
    assert (got.view(np.uint32) == whole["accum"].view(np.uint32)).all()


def test_statistically_the_same_image_as_the_parity_render(gpu_ctx):
    """Same estimator, other random numbers: the two images differ by Monte-Carlo noise only."""
    world, cam = host.build_scene(1, width=200, spp=64, depth=20)
    gpu_ctx.set_partition(0, 1, 8)
    gpu_ctx.upload_world(world)
    gpu_ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    par = gpu_ctx.render(cam, want_accum=True)["accum"].astype(np.float64)
    tp = render_tp(gpu_ctx, world, cam)["accum"].astype(np.float64)
    par2 = None
    gpu_ctx.rng_seed(12345, cam.image_width, cam.image_height)
    par2 = gpu_ctx.render(cam, want_accum=True)["accum"].astype(np.float64)
    # mean image level agrees to a fraction of a percent; pixel-wise differences are no larger than between two parity seeds
    assert abs(tp.mean() - par.mean()) < 0.01 * par.mean()
    noise = np.abs(par2 - par).mean()
    assert np.abs(tp - par).mean() < 1.25 * noise


def test_rejections(gpu_ctx, oracle):
    world, cam = host.build_scene(6, width=32, spp=4, depth=4)  # Cornell box: not a BVH-of-spheres world
    gpu_ctx.set_partition(0, 1, 8)
    gpu_ctx.upload_world(world)
    gpu_ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    with pytest.raises(hip.MortHipError) as e:
        gpu_ctx.render(cam, mode=hip.MODE_THROUGHPUT)
    assert e.value.status == -6
    world, cam = host.build_scene(1, width=32, spp=4, depth=4)
    gpu_ctx.upload_world(world)
    gpu_ctx.rng_load(oracle.seed_states(S.DEFAULT_SEED, cam.image_width, cam.image_height), cam.image_width, cam.image_height)
    with pytest.raises(hip.MortHipError) as e:  # loaded states carry no seed to derive the sub-streams from
        gpu_ctx.render(cam, mode=hip.MODE_THROUGHPUT)
    assert e.value.status == -5
