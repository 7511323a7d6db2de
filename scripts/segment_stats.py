"""Per-pixel segment counts of the headline frame: how long the longest pixel chains are next to the mean
(the frame cannot finish before its longest chain: DESIGN.md 6).  usage: segment_stats.py [scene] [spp]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mort_amd import host, hip
scene = int(sys.argv[1]) if len(sys.argv) > 1 else 1
world, cam = host.build_scene(scene, spp=int(sys.argv[2]) if len(sys.argv) > 2 else 500)
ctx = hip.Context(0)
ctx.upload_world(world)
ctx.rng_seed(69420, cam.image_width, cam.image_height)
out = ctx.render(cam, want_accum=False, want_segments=True)
out = ctx.render(cam, want_accum=False, want_segments=True)
seg = out["segments_px"].astype(np.int64)
st = out["stats"]
print(f"frame {st['seconds']*1e3:.1f} ms, {seg.sum()} segments, mean {seg.mean():.0f} per pixel")
for q in (50, 90, 99, 99.9, 99.99, 100):
    print(f"  p{q}: {np.percentile(seg, q):.0f}")
tiles = seg[: seg.shape[0] // 8 * 8, : seg.shape[1] // 8 * 8].reshape(seg.shape[0] // 8, 8, seg.shape[1] // 8, 8).sum(axis=(1, 3))
print(f"tiles: mean {tiles.mean():.0f}, max {tiles.max()}, tiles above 3x mean: {(tiles > 3 * tiles.mean()).sum()} of {tiles.size}")
top = np.sort(seg.ravel())[::-1]
print("time per segment if the frame were bound by the longest pixel: %.2f us" % (st['seconds'] * 1e6 / top[0]))
for k in (64, 1024, 16384): print(f"  pixels {k}: >= {top[k-1]} segments")
