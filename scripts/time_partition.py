"""Time one rank's share of the headline frame for N-way partitions on a single GPU (rehearsal of the
multi-GPU scaling run: the slowest rank bounds the frame).  usage: time_partition.py [N ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mort_amd import host, hip
world, cam = host.build_scene(1, spp=int(os.environ.get("SPP", "500")))
ctx = hip.Context(0)
ctx.upload_world(world)
ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
base = float(os.environ.get("BASE_MS", "0")) * 1e-3 or None
for n in ns:
    worst = 0.0
    ranks = range(n) if os.environ.get("ALL_RANKS") else [0, n - 1] if n > 1 else [0]
    for r in ranks:
        ctx.set_partition(r, n, 8)
        ctx.rng_seed(69420, cam.image_width, cam.image_height)
        out = ctx.render(cam, want_accum=False)  # first frame: tiles ordered by the one-sample probe
        out = ctx.render(cam, want_accum=False)  # steady state: ordered by the previous frame's costs
        worst = max(worst, out["stats"]["seconds"])
    base = base or worst * n
    print(f"N={n}: slowest rank {worst*1e3:.1f} ms  -> scaling efficiency {base / (n * worst) * 100:.1f}%  lds={out['stats']['kernel_lds_bytes']}", flush=True)
