#!/usr/bin/env python3
"""wave_roles.py SCENE WIDTH SPP: when heavy and ordinary waves of a heavy-wave launch end (profile build + MORT_WAVE_LINES=1; role = wave index mod 3 < 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
from mort_amd import host, hip, structs as S
sid, width, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
world, cam = host.build_scene(sid, width=width, spp=spp)
os.environ["MORT_WAVE_LINES"] = "1"
with hip.Context(0) as ctx:
    ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    for f in range(3):
        st = ctx.render(cam, want_accum=False)["stats"]
    print(f"FRAME {st['kernel_name']} {st['seconds']*1e3:.1f} ms", flush=True)
    L = hip.lib()
    L.mort_hip_debug_wave_log.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]; L.mort_hip_debug_wave_log.restype = C.c_int
    buf = np.zeros((8192, 16), dtype=np.uint64)
    n = L.mort_hip_debug_wave_log(ctx._h, buf.ctypes.data, 8192)
a = buf[:n].astype(np.float64)
a = a[a[:, 2] > 0]
idx = (a[:, 0] * 16 + a[:, 1]).astype(int)
end = a[:, 2] * 1e-5
for name, sel in (("heavy waves", idx % 3 < 2), ("ordinary waves", idx % 3 == 2)):
    e = end[sel]
    if len(e): print(f"{name:15s} {len(e):5d}: end p10 {np.percentile(e,10):7.1f} p50 {np.percentile(e,50):7.1f} p90 {np.percentile(e,90):7.1f} p99 {np.percentile(e,99):7.1f} max {e.max():7.1f} ms; S steps per wave {a[sel,6].mean():.0f}")
