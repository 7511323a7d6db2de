#!/usr/bin/env python3
"""bisect_check.py MODE SCENE WIDTH SPP : for every variant library under build/bis/<N>/ (scripts/bisect_build.sh), render in a child
process with MORT_HIP_LIB set and count the pixels that differ from the oracle.  Diagnostic tool."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 5 and sys.argv[5] == "child":
    import numpy as np
    from mort_amd import host, hip
    mode, sid, width, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    world, cam = host.build_scene(sid, width=width, spp=spp)
    ref = np.load("/tmp/bisect_ref.npy")
    with hip.Context(0) as ctx:
        ctx.upload_world(world)
        ctx.rng_seed(69420, cam.image_width, cam.image_height)
        out = ctx.render(cam, mode=hip.MODE_WAVE if mode == "wave" else hip.MODE_MEGA, want_accum=True)
    bad = int((out["accum"].view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    print(json.dumps({"bad_pixels": bad, "kernel": out["stats"]["kernel_name"], "ms": out["stats"]["seconds"] * 1e3}))
    sys.exit(0)
import numpy as np
from mort_amd import host
from tests import oracle_lib as O
mode, sid, width, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
world, cam = host.build_scene(sid, width=width, spp=spp)
np.save("/tmp/bisect_ref.npy", O.render(world, cam, nthreads=16)["accum"])
for n in sorted(os.listdir(os.path.join(ROOT, "build", "bis")), key=lambda s: int(s)):
    env = dict(os.environ, MORT_HIP_LIB=os.path.join(ROOT, "build", "bis", n, "libmort_hip.so"), MORT_GEN_MIN_PRIMS="0")
    p = subprocess.run(["timeout", "-k", "10", "120", sys.executable, __file__, mode, str(sid), str(width), str(spp), "child"], env=env, capture_output=True, text=True)
    last = open(os.path.join(ROOT, "build", "bis", n, "last_pass.txt")).read().strip()
    print(n, p.stdout.strip() or ("rc=%d " % p.returncode + p.stderr.strip()[-200:]), "|", last, flush=True)
