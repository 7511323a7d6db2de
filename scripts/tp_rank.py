import os, sys, json
sys.path.insert(0, '/root/repo')
from mort_amd import host, hip, structs as S
world, cam = host.build_scene(1, width=1200, spp=500)
W, H = cam.image_width, cam.image_height
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
res = []
for r in (0, n - 1):
    with hip.Context(0) as ctx:
        ctx.set_partition(r, n, 8); ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, W, H)
        for f in range(3): st = ctx.render(cam, want_accum=False)["stats"]
        res.append((r, round(st["seconds"] * 1e3, 2), st["kernel_name"], st["kernel_lds_bytes"]))
print(os.environ.get("MORT_FAST_BLOCK_SIZE"), os.environ.get("MORT_FAST_BLOCKS_PER_CU"), os.environ.get("MORT_CHAIN_BOUND"), n, res, flush=True)
