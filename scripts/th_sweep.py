#!/usr/bin/env python3
"""th_sweep.py SCENE WIDTH SPP ENVVAR VALUE... : steady-state frame time for each value of a tuning variable of libmort_hip.so
(MORT_THRESHOLDS "s,l,k", MORT_GEN_THRESHOLDS "s,l,k,m", ...), one child process per value."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("TH_CHILD"):
    from mort_amd import host, hip, structs as S
    world, cam = host.build_scene(int(sys.argv[1]), width=int(sys.argv[2]), spp=int(sys.argv[3]), aspect=(1.7777778 if int(sys.argv[2]) == 1920 else None))
    with hip.Context(0) as ctx:
        ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
        mode = hip.MODE_WAVE if os.environ.get("TH_MODE") == "wave" else hip.MODE_MEGA
        ts = [ctx.render(cam, mode=mode, want_accum=False)["stats"]["seconds"] * 1e3 for f in range(3)]
    print(round(min(ts[1:]), 2)); sys.exit(0)
for th in sys.argv[5:]:
    env = dict(os.environ, TH_CHILD="1")
    env[sys.argv[4]] = th
    p = subprocess.run(["timeout", "-k", "10", "100", sys.executable, __file__] + sys.argv[1:4], env=env, capture_output=True, text=True)
    print(th, p.stdout.strip() or p.stderr.strip()[-200:], flush=True)
