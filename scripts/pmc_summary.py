#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per (kernel, counter).

usage: pmc_summary.py DIR [DIR ...] > summary.csv
Each DIR is a rocprofv3 -d output directory of one --pmc pass.  A counter's rows of one dispatch (one per XCD / SE
dimension) are summed; only kernels whose name contains 'mega' or 'wf_' are kept.  per_dispatch = mean over the
dispatches of that kernel, max_dispatch = the largest one (the frame itself when a one-sample cost probe of the same
kernel runs first).
"""
import csv, glob, os, sys, collections

per = collections.defaultdict(float)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                if "mega" not in k and "wf_" not in k:
                    continue
                k = k.split("(")[0].replace(",", "")
                per[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
tot = collections.defaultdict(list)
for (k, c, _), v in per.items():
    tot[(k, c)].append(v)
print("kernel,counter,sum_over_dispatches,dispatches,per_dispatch,max_dispatch")
for (k, c), vs in sorted(tot.items()):
    print(f"{k},{c},{sum(vs):.6g},{len(vs)},{sum(vs) / len(vs):.6g},{max(vs):.6g}")
