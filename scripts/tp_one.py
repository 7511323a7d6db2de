#!/usr/bin/env python3
"""tp_one.py SCENE WIDTH SPP NRANKS [ASPECT] : rank 0 of an N-way partition, four frames, with the longest pixel chain of the last one
(honours MORT_HIP_LIB and the tuning variables)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mort_amd import host, hip, structs as S
sid, width, spp, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
aspect = float(sys.argv[5]) if len(sys.argv) > 5 else None
world, cam = host.build_scene(sid, width=width, spp=spp, aspect=aspect)
W, H = cam.image_width, cam.image_height
with hip.Context(0) as ctx:
    ctx.set_partition(0, n, 8); ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, W, H)
    ts = []
    for f in range(4):
        out = ctx.render(cam, want_accum=False, want_segments=(f == 3))
        ts.append(round(out["stats"]["seconds"] * 1e3, 1))
    seg = out["segments_px"]
    own = seg[seg > 0]
    print(os.path.basename(os.path.dirname(os.environ.get("MORT_HIP_LIB", "default/x"))), {k: os.environ.get(k) for k in ("MORT_NO_TILE_ORDER", "MORT_GEN_BLOCK_SIZE", "MORT_GEN_THRESHOLDS") if os.environ.get(k)}, "N", n, ts, out["stats"]["kernel_name"], "px", own.size, "max seg/px", int(own.max()), "mean", round(float(own.mean()), 1), flush=True)
