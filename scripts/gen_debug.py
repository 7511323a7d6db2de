#!/usr/bin/env python3
"""Debug aid: unified-tree kernel vs the one-lane-per-pixel kernel on one world; prints where they differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mort_amd import host, hip, structs as S
from tests.worlds import FLAT_WORLDS, flat_world, flat_camera

def run(world, cam, env):
    for k in ("MORT_NO_GEN", "MORT_GEN_BLOCK_SIZE", "MORT_GEN_THRESHOLDS", "MORT_NO_TILE_ORDER", "MORT_GEN_LANE_WALK", "MORT_GEN_DL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with hip.Context(0) as ctx:
        ctx.upload_world(world)
        ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
        return ctx.render(cam, want_accum=True, want_segments=True)

def compare(tag, world, cam):
    ref = run(world, cam, {"MORT_NO_GEN": "1"})
    for env in ({},):
        out = run(world, cam, env)
        bad = (out["accum"].view(np.uint32) != ref["accum"].view(np.uint32)).any(axis=2)
        segbad = out["segments_px"] != ref["segments_px"]
        print(f"{tag} {env}: kernel {out['stats']['kernel_name']} pixels differing {int(bad.sum())} of {bad.size}, seg-count differing {int(segbad.sum())}, "
              f"segments {out['stats']['segments']} vs {ref['stats']['segments']}, scans {out['stats']['reference_walks']}")
        if bad.any():
            ys, xs = np.nonzero(bad)
            for y, x in list(zip(ys, xs))[:12]:
                print("    px", x, y, "seg", out["segments_px"][y, x], ref["segments_px"][y, x], "acc", out["accum"][y, x], ref["accum"][y, x])

for name in sys.argv[1:] or ["concentric_glass"]:
    if name.startswith("s"):
        sid = int(name[1:])
        world, cam = host.build_scene(sid, width=64, spp=1, depth=int(os.environ.get("DEPTH", "8")))
    else:
        spec = FLAT_WORLDS[name]
        world, ids = flat_world(spec["prims"], media=spec.get("media", ()), late_list=spec.get("late_list", False))
        cam = flat_camera(light=ids[spec["light"][1]] if spec.get("light") else None, spp=1, width=64, depth=int(os.environ.get("DEPTH", "8")))
    compare(name, world, cam)
