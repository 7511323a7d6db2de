#!/usr/bin/env python3
"""pmc_hbm_json.py RAW.csv SCENE WIDTH SPP MODE "mort args" -> the *_pmc_hbm.json bench.py reads.

RAW.csv is scripts/pmc_summary.py output of the FETCH_SIZE and WRITE_SIZE passes (values in KB, summed over XCDs).
Per launch of the dominant kernel (the one with the largest total): read bytes = FETCH_SIZE x 1024 (lower) and x 2 (the
guide's gfx950 correction for wide streams, upper), write bytes = WRITE_SIZE x 1024; traffic = upper read + write."""
import csv, json, sys

raw, scene, width, spp, mode, margs = sys.argv[1:7]
rows = list(csv.DictReader(open(raw)))
tot = {}
for r in rows:
    targs = r["kernel"].split("<")[-1].split(">")[0].replace(",", " ").split()
    if "mega_bvh_kernel" in r["kernel"] and "true" in targs[1:2] + targs[3:4]:
        continue  # mega_bvh_kernel<BLOCK, PROBE, DRAIN, SUB>: the one-sample cost probe / the non-parity launch
    # (value, divisor): the frame's own dispatch = the largest one (a one-sample cost probe of the same kernel may precede it)
    tot.setdefault(r["kernel"], {})[r["counter"]] = (float(r.get("max_dispatch") or r["sum_over_dispatches"]), 1 if r.get("max_dispatch") else int(r["dispatches"]))
kern = max(tot, key=lambda k: sum(v[0] for v in tot[k].values()))
f, nf = tot[kern].get("FETCH_SIZE", (0.0, 1))
w, nw = tot[kern].get("WRITE_SIZE", (0.0, 1))
wave = mode == "wave"
# wavefront mode: one render = many launches; report per render (all launches of all wf_ kernels)
if wave:
    f = sum(float(r["sum_over_dispatches"]) for r in rows if "wf_" in r["kernel"] and r["counter"] == "FETCH_SIZE")
    w = sum(float(r["sum_over_dispatches"]) for r in rows if "wf_" in r["kernel"] and r["counter"] == "WRITE_SIZE")
    nf = nw = 1
    kern = "wf_* (all launches of one render)"
# the frame's geometry and depth as the host scene layer resolves them (bench.py quotes a counter set only for the same frame)
import os, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mort_amd import host
m_as = re.search(r"--aspect\s+(\S+)", margs); m_dp = re.search(r"--depth\s+(\S+)", margs)
_, cam = host.build_scene(int(scene), width=int(width), spp=int(spp), depth=int(m_dp.group(1)) if m_dp else None, aspect=float(m_as.group(1)) if m_as else None)
out = {
    "FETCH_SIZE_KB": f / nf, "WRITE_SIZE_KB": w / nw,
    "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- ./mort_amd/bin/mort {margs}  (scripts/profile_set.sh)",
    "kernel": kern,
    "config": {"scene": int(scene), "width": int(width), "height": cam.image_height, "spp": int(spp), "depth": cam.bounce_limit, "gpus": 1, "mode": mode},
    "hbm_read_bytes_lower": f / nf * 1024, "hbm_read_bytes_upper": f / nf * 2048, "hbm_write_bytes": w / nw * 1024,
    "traffic_bytes_per_launch": f / nf * 2048 + w / nw * 1024,
    "note": "L2 <-> fabric bytes (upper bound of HBM traffic: the 256 MB MALL sits behind these counters); FETCH_SIZE x2 is the guide's gfx950 correction for wide coalesced streams",
}
print(json.dumps(out, indent=1))
