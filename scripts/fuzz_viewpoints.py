"""One-off hunt for rays on which the BVH megakernel and the oracle disagree: random cameras (position, target, field of
view, defocus, time) on the BVH scenes (or, with a third argument "all", on all ten scenes: generic kernel, this build's
trees over long runs), bit-for-bit comparison.  usage: fuzz_viewpoints.py [cases] [seed] [all]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mort_amd import host, hip  # noqa: E402
from tests import oracle_lib as oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = hip.Context(0)
bad = 0
walks = segs = 0
for k in range(cases):
    sid = (1 if k % 4 else 10) if len(sys.argv) <= 3 else int(rng.integers(1, 11))
    big = sid in (8, 9)
    world, cam = host.build_scene(sid, width=int(rng.integers(40, 90) if big else rng.integers(90, 260)), spp=int(rng.choice([4, 9]) if big else rng.choice([4, 9, 16])), depth=int(rng.choice([3, 8, 20, 50])))
    scale = {5: 1.0, 6: 45.0, 7: 45.0, 8: 45.0, 9: 45.0}.get(sid, 1.0)
    frm = (rng.uniform(-12, 12, 3) * (1, 0.25, 1) + (0, 0.6, 0)) * scale + ((278, 278, -300) if scale > 1 else (0, 0, 0))
    if k % 5 == 0: frm[1] = rng.uniform(-2, 0.19)      # under or at ground level
    at = rng.uniform(-6, 6, 3) * (1, 0.15, 1) * scale + ((278, 278, 278) if scale > 1 else (0, 0, 0))
    for i in range(3):
        cam.lookfrom.e[i] = float(frm[i]); cam.lookat.e[i] = float(at[i])
    cam.vfov = int(rng.choice([10, 20, 40, 90, 120]))
    cam.defocus_angle = float(rng.choice([0.0, 0.3, 2.0]))
    cam.focus_dist = float(rng.uniform(1, 15))
    host.lib().mort_camera_initialize(C.byref(cam))
    W, H = cam.image_width, cam.image_height
    seed = int(rng.integers(1, 1 << 31))
    ref = oracle.render(world, cam, seed=seed, nthreads=min(len(os.sched_getaffinity(0)), 16))
    ctx.set_partition(0, 1, 8); ctx.upload_world(world); ctx.rng_seed(seed, W, H)
    out = ctx.render(cam, want_accum=True, want_segments=True)
    st = ctx.rng_store(W, H, oracle.STATE_DTYPE)
    same = (out["rgba"] == ref["rgba"]).all() and (out["accum"].view(np.uint32) == ref["accum"].view(np.uint32)).all() and \
           (out["segments_px"] == ref["segments_px"]).all() and (st["v"] == ref["states"]["v"]).all()
    walks += out["stats"]["reference_walks"]; segs += out["stats"]["segments"]
    if not same:
        bad += 1
        print(f"MISMATCH case {k}: scene {sid} {W}x{H} spp {cam.samples_per_pixel} depth {cam.bounce_limit} from {frm} at {at} vfov {cam.vfov} seed {seed}", flush=True)
print(f"{cases} cases, {segs} segments, {walks} reference walks, {bad} mismatches")
sys.exit(1 if bad else 0)
