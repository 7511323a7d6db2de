"""Frame times of consecutive frames of the headline scene (first frame: probe-ordered tiles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mort_amd import host, hip
world, cam = host.build_scene(1, spp=500)
ctx = hip.Context(0)
ctx.upload_world(world)
ctx.rng_seed(int(os.environ.get("SEED", "69420")), cam.image_width, cam.image_height)
for f in range(5):
    out = ctx.render(cam, want_accum=bool(int(os.environ.get("ACCUM", "0"))))
    print(f"frame {f}: {out['stats']['seconds']*1e3:.1f} ms, segments {out['stats']['segments']}", flush=True)
