#!/usr/bin/env python3
"""Debug aid: full-size frame through both kernels, list the pixels that differ.  usage: gen_diff_big.py SCENE WIDTH SPP"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mort_amd import host, hip, structs as S
sid, width, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
world, cam = host.build_scene(sid, width=width, spp=spp)
res = {}
for tag, env in (("gen", {}), ("old", {"MORT_NO_GEN": "1"}), ("gen512", {"MORT_GEN_BLOCK_SIZE": "512"}), ("lane", {"MORT_GEN_LANE_WALK": "1"})):
    for k in ("MORT_NO_GEN", "MORT_GEN_BLOCK_SIZE", "MORT_GEN_LANE_WALK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with hip.Context(0) as ctx:
        ctx.upload_world(world)
        ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
        res[tag] = ctx.render(cam, want_accum=True, want_segments=True)
    print(tag, res[tag]["stats"]["kernel_name"], res[tag]["stats"]["segments"], res[tag]["stats"]["reference_walks"], flush=True)
ref = res["old"]
for tag in ("gen", "gen512", "lane"):
    o = res[tag]
    bad = (o["accum"].view(np.uint32) != ref["accum"].view(np.uint32)).any(axis=2) | (o["segments_px"] != ref["segments_px"])
    ys, xs = np.nonzero(bad)
    print(tag, "differing pixels:", len(ys), [(int(x), int(y), int(o["segments_px"][y, x]), int(ref["segments_px"][y, x])) for y, x in zip(ys, xs)][:20], flush=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"diff_s{sid}_{width}_{spp}.npz"), seg_old=ref["segments_px"], acc_old=ref["accum"], seg_gen=res["gen"]["segments_px"], acc_gen=res["gen"]["accum"])
