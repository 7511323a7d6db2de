#!/usr/bin/env python3
"""Per-state step / lane / cycle counts of the state-machine megakernels (needs a -DMORT_PROFILE_STATES build:
make hip LIBDIR=build/prof/lib OBJDIR=build/prof/obj HIPFLAGS_EXTRA=-DMORT_PROFILE_STATES; MORT_HIP_LIB=build/prof/lib/libmort_hip.so).
usage: gen_profile.py SCENE WIDTH SPP [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mort_amd import host, hip, structs as S
sid, width, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2
world, cam = host.build_scene(sid, width=width, spp=spp, aspect=(1.0 if width == 4096 else 1.7777778 if width == 1920 else None))
with hip.Context(0) as ctx:
    ctx.upload_world(world)
    ctx.rng_seed(S.DEFAULT_SEED, cam.image_width, cam.image_height)
    for f in range(frames):
        out = ctx.render(cam, want_accum=False)
        st = out["stats"]
        print(f"frame {f}: {st['kernel_name']} {st['seconds']*1e3:.1f} ms, {st['segments']} segments, {st['segments']/st['seconds']/1e9:.3f} Gseg/s, scans {st['reference_walks']}, vgprs {st['kernel_vgprs']} lds {st['kernel_lds_bytes']}", flush=True)
