import sys
sys.path.insert(0,'/root/repo')
from mort_amd import host, hip, structs as S
sid,w,spp=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
world, cam = host.build_scene(sid, width=w, spp=spp)
with hip.Context(0) as ctx:
    ctx.upload_world(world); ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    for f in range(2):
        st = ctx.render(cam, mode=hip.MODE_WAVE, want_accum=False)["stats"]
        print(f, st["kernel_name"], round(st["seconds"]*1e3,1), flush=True)
