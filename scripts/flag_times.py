#!/usr/bin/env python3
"""flag_times.py SCENE WIDTH SPP [N_RANKS] : steady-state frame time with every library under build/fv/*/ (scripts/flag_variants.sh), each in a
child process (MORT_HIP_LIB); also checks the frame's segment count and image hash against the first variant's."""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("FLAG_TIMES_CHILD"):
    from mort_amd import host, hip, structs as S
    sid, width, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    world, cam = host.build_scene(sid, width=width, spp=spp)
    with hip.Context(0) as ctx:
        ctx.set_partition(0, n, 8); ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
        ts = []
        for f in range(4):
            out = ctx.render(cam, mode=(hip.MODE_WAVE if os.environ.get("FLAG_MODE") == "wave" else hip.MODE_MEGA), want_accum=False); ts.append(out["stats"]["seconds"] * 1e3)
    print(json.dumps({"ms": round(min(ts[1:]), 2), "kernel": out["stats"]["kernel_name"], "vgprs": out["stats"]["kernel_vgprs"], "segments": out["stats"]["segments"],
                      "sha": hashlib.sha1(out["rgba"].tobytes()).hexdigest()[:12]}))
    sys.exit(0)
d = os.path.join(ROOT, "build", "fv")
for name in sorted(os.listdir(d)):
    env = dict(os.environ, MORT_HIP_LIB=os.path.join(d, name, "libmort_hip.so"), FLAG_TIMES_CHILD="1")
    p = subprocess.run(["timeout", "-k", "10", "120", sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    print(f"{name:14s}", p.stdout.strip() or ("rc=%d " % p.returncode + p.stderr.strip()[-200:]), flush=True)
