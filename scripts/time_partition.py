#!/usr/bin/env python3
"""One rank of an N-way row partition timed on a single MI355X, N = 1, 2, 4, 8: the kernel-time part of strong scaling
(the gather is a 0.4 MB point-to-point copy per peer).  Prints one JSON object (kept under profiles/).
usage: time_partition.py [SCENE WIDTH SPP [mega|throughput [ASPECT]]]   (throughput = the labelled non-parity mode)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mort_amd import host, hip, structs as S
sid, width, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 1200, 500)
mode_name = sys.argv[4] if len(sys.argv) > 4 else "mega"
mode = {"mega": hip.MODE_MEGA, "throughput": hip.MODE_THROUGHPUT}[mode_name]
aspect = float(sys.argv[5]) if len(sys.argv) > 5 else None
world, cam = host.build_scene(sid, width=width, spp=spp, aspect=aspect)
W, H = cam.image_width, cam.image_height
out = {"mode": mode_name, "scene": sid, "width": W, "height": H, "spp": spp, "note": "steady-state frame (third of three) of the first and the last rank of each partition, one GPU", "ranks": {}}
for n in (1, 2, 4, 8):
    times = []
    for r in sorted({0, n - 1}):
        with hip.Context(0) as ctx:
            ctx.set_partition(r, n, 8)
            ctx.upload_world(world)
            ctx.rng_seed(S.DEFAULT_SEED, W, H)
            for f in range(3):
                st = ctx.render(cam, mode=mode, want_accum=False)["stats"]
            times.append({"rank": r, "ms": st["seconds"] * 1e3, "segments": st["segments"], "kernel": st["kernel_name"]})
    out["ranks"][str(n)] = times
t1 = max(t["ms"] for t in out["ranks"]["1"])
out["summary"] = {str(n): {"ms_slowest_rank": max(t["ms"] for t in out["ranks"][str(n)]), "efficiency_from_kernel_time": t1 / (n * max(t["ms"] for t in out["ranks"][str(n)]))} for n in (1, 2, 4, 8)}
print(json.dumps(out, indent=1))
