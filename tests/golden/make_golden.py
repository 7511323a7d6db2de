#!/usr/bin/env python3
"""Generates the committed golden fixtures.  Run from the repo root in the build container:

    python tests/golden/make_golden.py

1. earthmap_rgb.npz -- texels of the reference's imgs/earthmap.jpg decoded with the reference's
   vendored stb_image.h (oracle/_ref/stb_decode, compiled from /root/reference where it lies).
   Needs /root/reference; skipped (existing file kept) when it is absent.
2. oracle_golden.npz -- outputs of the CPU oracle (oracle/mort_oracle.c) for small configurations
   of every scene family: uchar4 image, fp32 accumulators, per-pixel segment counts, and the final
   RNG words of a few pixels.  BASELINE config 1 (Scene 1, 200x112, 4 spp) is case "s1_c1".

The reference ships no golden data of its own, so these vectors pin the oracle against
regressions and give the GPU tests a fixed target; they do not pin it against the CUDA render
("parity unpinned").
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {  # name: (scene, width, spp, depth)
    "s1_c1": (1, 200, 4, None),
    "s2": (2, 96, 4, None),
    "s3": (3, 96, 4, None),
    "s4": (4, 96, 4, None),
    "s5": (5, 64, 9, None),
    "s6": (6, 64, 9, None),
    "s7": (7, 48, 9, None),
    "s9": (9, 48, 4, None),
    "s10": (10, 160, 1, None),
    "s1_depth3": (1, 96, 4, 3),
}


def make_earth():
    ref = "/root/reference/imgs/earthmap.jpg"
    if not os.path.exists(ref):
        print("reference absent: keeping existing earthmap fixture")
        return
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    tmp = "/tmp/mort_earthmap.ppm"
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "stb_decode"), ref, tmp])
    data = open(tmp, "rb").read()
    hdr, rest = data.split(b"255\n", 1)
    w, h = [int(t) for t in hdr.split()[1:3]]
    rgb = np.frombuffer(rest, dtype=np.uint8).reshape(h, w, 3)
    np.savez_compressed(os.path.join(HERE, "earthmap_rgb.npz"), rgb=rgb)
    print("earthmap", w, h)


def make_oracle():
    from mort_amd import host
    from tests import oracle_lib as O
    out = {}
    for name, (sid, width, spp, depth) in CASES.items():
        world, cam = host.build_scene(sid, width=width, spp=spp, depth=depth)
        r = O.render(world, cam, nthreads=8)
        out[name + "_rgba"] = r["rgba"]
        out[name + "_accum"] = r["accum"]
        out[name + "_segpx"] = r["segments_px"].astype(np.uint16)
        out[name + "_states"] = np.stack([r["states"]["d"][:64], *[r["states"]["v"][:64, k] for k in range(5)]], axis=1)
        out[name + "_meta"] = np.array([sid, cam.image_width, cam.image_height, spp, cam.bounce_limit, r["segments"], r["rng_draws"]], dtype=np.int64)
        print(name, cam.image_width, cam.image_height, r["segments"])
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)


if __name__ == "__main__":
    make_earth()
    make_oracle()
