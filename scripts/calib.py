#!/usr/bin/env python3
"""Roofline calibration on the GPU box (include/mort_hip.h mort_hip_calib_*): cycles per wave64 VALU instruction per SIMD at
1..8 resident waves for five instruction mixes, and the HBM copy rate.  Prints one JSON object (kept under profiles/).
usage: calib.py [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mort_amd import hip
KINDS = {0: "independent v_fma_f32", 1: "dependent v_fma_f32 chain", 2: "independent v_fma_f64", 3: "3 v_fma_f32 : 1 s_add_u32", 4: "independent v_pk_fma_f32"}
out = {"valu": [], "hbm_copy_GBs": None}
with hip.Context(0) as ctx:
    for kind in KINDS:
        for w in (1, 2, 3, 4, 6, 8):
            r = ctx.calib_valu(w, kind)
            r["mix"] = KINDS[kind]
            out["valu"].append(r)
            print(f"{KINDS[kind]:28s} asked {w} waves/SIMD, resident {r['resident_waves_per_simd']:.0f} on {r['simds_seen']} SIMDs: {r['cycles_per_valu_per_simd']:.3f} cycles per VALU per SIMD, "
                  f"{r['cycles_per_valu_per_wave']:.3f} per wave, clock {r['clock_ghz']:.2f} GHz, {r['seconds']*1e3:.2f} ms", file=sys.stderr, flush=True)
    out["hbm_copy_GBs"] = ctx.calib_hbm_copy(1 << 30, 5)
    print(f"float4 copy, 1 GiB per buffer: {out['hbm_copy_GBs']:.0f} GB/s (read + write)", file=sys.stderr)
s = json.dumps(out, indent=1)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(s + "\n")
print(s)
