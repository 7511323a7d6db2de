#!/usr/bin/env python3
"""Debug aid (-DMORT_DEBUG_PRINT builds): prints every segment of one pixel from both kernels.  usage: gen_debug_px.py WORLD X Y DEPTH"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mort_amd import host, hip, structs as S
from tests.worlds import FLAT_WORLDS, flat_world, flat_camera
name, x, y, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
spec = FLAT_WORLDS[name]
world, ids = flat_world(spec["prims"], media=spec.get("media", ()))
cam = flat_camera(light=ids[spec["light"][1]] if spec.get("light") else None, spp=1, width=64, depth=depth)
os.environ["MORT_DEBUG_PIXEL"] = str(x + y * cam.image_width)
for env in ({"MORT_NO_GEN": "1"}, {}, {"MORT_GEN_LANE_WALK": "1"}):
    for k in ("MORT_NO_GEN", "MORT_GEN_LANE_WALK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    print("----", env, flush=True)
    with hip.Context(0) as ctx:
        ctx.upload_world(world)
        ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
        out = ctx.render(cam, want_accum=True, want_segments=True)
    print("acc", out["accum"][y, x], "seg", out["segments_px"][y, x], flush=True)
