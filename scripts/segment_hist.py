#!/usr/bin/env python3
"""segment_hist.py SCENE WIDTH SPP [ASPECT]: how the per-pixel segment counts of a frame are distributed -- how many pixels carry chains close to the longest one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mort_amd import host, hip, structs as S
sid, width, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
aspect = float(sys.argv[4]) if len(sys.argv) > 4 else None
world, cam = host.build_scene(sid, width=width, spp=spp, aspect=aspect)
with hip.Context(0) as ctx:
    ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    out = ctx.render(cam, want_accum=False, want_segments=True)
seg = out["segments_px"].astype(np.float64)
mx, mean, tot = seg.max(), seg.mean(), seg.sum()
print(f"pixels {seg.size}, segments {int(tot)}, mean {mean:.1f}, max {int(mx)} = {mx/mean:.1f} x mean; frame {out['stats']['seconds']*1e3:.1f} ms")
for f in (0.9, 0.75, 0.5, 0.35, 0.25, 0.15):
    m = seg >= f * mx
    print(f"  pixels >= {f:.2f} x max: {int(m.sum()):7d} ({100*m.mean():.2f} %), carrying {100*seg[m].sum()/tot:.1f} % of the segments")
T = (seg.shape[0] // 8) * 8, (seg.shape[1] // 8) * 8
t = seg[:T[0], :T[1]].reshape(T[0] // 8, 8, T[1] // 8, 8).max(axis=(1, 3))
for f in (0.75, 0.5, 0.35, 0.25):
    print(f"  8x8 tiles whose longest pixel >= {f:.2f} x max: {int((t >= f * mx).sum())} of {t.size}")
