#!/usr/bin/env python3
"""Compare two renders (the reference only draws into a GL window).

  tools/mort_diff.py a.ppm b.ppm                 uchar4 images written by `mort --out`
  tools/mort_diff.py a.raw b.raw --f32 W H       fp32 accumulators written by `mort --dump-f32`

Prints: differing pixels, max abs difference, per-pixel RMSE, and for fp32 inputs the ULP histogram --
the report format for the north_star's "per-pixel RMSE vs the CUDA reference" once a CUDA dump exists.
Exit status 0 when identical, 1 otherwise.
"""
import argparse
import sys

import numpy as np


def read_ppm(path):
    data = open(path, "rb").read()
    hdr, rest = data.split(b"255\n", 1)
    toks = hdr.split()
    assert toks[0] == b"P6", "not a binary PPM"
    w, h = int(toks[1]), int(toks[2])
    return np.frombuffer(rest, dtype=np.uint8, count=w * h * 3).reshape(h, w, 3)


def ulp_distance(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--f32", nargs=2, type=int, metavar=("W", "H"))
    args = ap.parse_args()
    if args.f32:
        w, h = args.f32
        a = np.fromfile(args.a, dtype=np.float32).reshape(h, w, 3)
        b = np.fromfile(args.b, dtype=np.float32).reshape(h, w, 3)
        ulp = ulp_distance(a, b)
        diff_px = int((ulp > 0).any(axis=-1).sum())
        err = a.astype(np.float64) - b.astype(np.float64)
        rmse_px = np.sqrt((err ** 2).mean(axis=-1))
        print(f"pixels differing: {diff_px} of {w * h}")
        print(f"max abs diff: {np.abs(err).max():.9g}   image RMSE: {np.sqrt((err ** 2).mean()):.9g}   worst per-pixel RMSE: {rmse_px.max():.9g}")
        for lim in (0, 1, 2, 4, 16, 256):
            print(f"  channels with ULP distance > {lim}: {int((ulp > lim).sum())}")
        print(f"pixels with per-pixel RMSE >= 1e-5 (north_star bound): {int((rmse_px >= 1e-5).sum())}")
        return 0 if diff_px == 0 else 1
    a, b = read_ppm(args.a), read_ppm(args.b)
    if a.shape != b.shape:
        print("shapes differ", a.shape, b.shape)
        return 1
    d = np.abs(a.astype(int) - b.astype(int))
    diff_px = int((d > 0).any(axis=-1).sum())
    print(f"pixels differing: {diff_px} of {a.shape[0] * a.shape[1]}   max abs diff: {d.max()}   RMSE (8-bit): {np.sqrt((d.astype(float) ** 2).mean()):.6g}")
    return 0 if diff_px == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
