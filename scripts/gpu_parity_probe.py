"""Ad-hoc GPU probe: HIP render vs oracle on small configs, prints mismatch counts and timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mort_amd import host, hip
from tests import oracle_lib as O

cases = [(1, 200, 4, None), (10, 200, 1, None), (2, 160, 4, None), (5, 96, 16, None), (6, 96, 16, None),
         (7, 64, 16, None), (4, 160, 4, None), (3, 160, 4, None), (9, 64, 16, None)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) if v != 'None' else None for v in a.split(',')) for a in sys.argv[1:]]
MODE = int(os.environ.get('MORT_PROBE_MODE', '0'))
ctx = hip.Context(0)
ok = True
for sid, w, spp, depth in cases:
    world, cam = host.build_scene(sid, width=w, spp=spp, depth=depth)
    t0 = time.time()
    ref = O.render(world, cam, nthreads=16)
    t_cpu = time.time() - t0
    ctx.upload_world(world)
    ctx.rng_seed(69420, cam.image_width, cam.image_height)
    st0 = ctx.rng_store(cam.image_width, cam.image_height, O.STATE_DTYPE)
    seeds = O.seed_states(69420, cam.image_width, cam.image_height)
    seed_ok = bool((st0['d'] == seeds['d']).all() and (st0['v'] == seeds['v']).all())
    out = ctx.render(cam, mode=MODE, want_accum=True, want_segments=True)
    st1 = ctx.rng_store(cam.image_width, cam.image_height, O.STATE_DTYPE)
    rg = (out['rgba'] != ref['rgba']).any(axis=-1).sum()
    ac = (out['accum'].view(np.uint32) != ref['accum'].view(np.uint32)).any(axis=-1).sum()
    sg = (out['segments_px'] != ref['segments_px']).sum()
    stt = ((st1['d'] != ref['states']['d']) | (st1['v'] != ref['states']['v']).any(axis=-1)).sum()
    s = out['stats']
    print(f"scene {sid} {cam.image_width}x{cam.image_height} spp={spp}: seed_ok={seed_ok} rgba_mismatch={rg} accum_mismatch={ac} "
          f"seg_mismatch={sg} state_mismatch={stt} segs gpu={s['segments']} cpu={ref['segments']} draws gpu={s['rng_draws']} cpu={ref['rng_draws']} "
          f"gpu_s={s['seconds']:.4f} cpu_s={t_cpu:.2f} vgprs={s['kernel_vgprs']}", flush=True)
    if rg or ac or sg or stt or not seed_ok:
        ok = False
        bad = np.argwhere((out['accum'].view(np.uint32) != ref['accum'].view(np.uint32)).any(axis=-1))[:5]
        for y, x in bad:
            print("   px", x, y, out['accum'][y, x], ref['accum'][y, x], out['segments_px'][y, x], ref['segments_px'][y, x])
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
