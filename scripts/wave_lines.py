#!/usr/bin/env python3
"""wave_lines.py SCENE WIDTH SPP [ASPECT]: per-wave lifetimes of a state-machine megakernel (profile build + MORT_WAVE_LINES=1): how much of the
frame is throughput (all waves alive) and how much is tail, and what a round costs in the waves that end last."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
from mort_amd import host, hip, structs as S
sid, width, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
aspect = float(sys.argv[4]) if len(sys.argv) > 4 else None
world, cam = host.build_scene(sid, width=width, spp=spp, aspect=aspect)
os.environ["MORT_WAVE_LINES"] = "1"
with hip.Context(0) as ctx:
    ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    for f in range(3):
        st = ctx.render(cam, want_accum=False)["stats"]
    print(f"FRAME {st['kernel_name']} {st['seconds']*1e3:.1f} ms segments {st['segments']}", flush=True)
    L = hip.lib()
    L.mort_hip_debug_wave_log.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]; L.mort_hip_debug_wave_log.restype = C.c_int
    buf = np.zeros((8192, 16), dtype=np.uint64)
    n = L.mort_hip_debug_wave_log(ctx._h, buf.ctypes.data, 8192)
a = buf[:n].astype(np.float64)
a = a[a[:, 2] > 0]
print("waves", len(a))
if len(a):
    end_ms = a[:, 2] * 1e-5
    print("wave end (ms): p10 %.1f p25 %.1f p50 %.1f p75 %.1f p90 %.1f p99 %.1f max %.1f; mean %.1f = %.0f %% of the frame's wave slots busy" % (tuple(np.percentile(end_ms, [10, 25, 50, 75, 90, 99, 100])) + (end_ms.mean(), 100 * end_ms.mean() / end_ms.max())))
    order = np.argsort(end_ms)
    for name, sel in (("all waves", order), ("first half to end", order[:len(order) // 2]), ("last 5 % to end", order[-max(1, len(order) // 20):]), ("last 8 waves", order[-8:])):
        b = a[sel]
        Ssteps = b[:, 6].sum()
        cyc = b[:, 7:12].sum()
        print(f"{name:18s}: end {b[:,2].mean()*1e-5:7.1f} ms  S steps/wave {b[:,6].mean():8.0f}  per S step: T {b[:,3].sum()/Ssteps:5.1f} L {b[:,4].sum()/Ssteps:4.2f} M {b[:,5].sum()/Ssteps:4.2f} steps; cycles per S-round {cyc/Ssteps:8.0f} "
              f"(T {b[:,7].sum()/Ssteps:6.0f} L {b[:,8].sum()/Ssteps:6.0f} M {b[:,9].sum()/Ssteps:6.0f} S {b[:,10].sum()/Ssteps:6.0f} sched {b[:,11].sum()/Ssteps:6.0f}); us per round {(b[:,2]*1e-2).sum()/Ssteps:6.1f}")
