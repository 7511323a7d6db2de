#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per (kernel, counter).

usage: pmc_summary.py DIR [DIR ...] > summary.csv
Each DIR is a rocprofv3 -d output directory of one --pmc pass.  Values are summed over dispatches' dimensions
(XCDs / SEs); only kernels whose name contains 'mega' or 'wf_' are kept, the one-sample probe is listed apart.
"""
import csv, glob, os, sys, collections

tot = collections.defaultdict(float)
calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"]
                if "mega" not in k and "wf_" not in k:
                    continue
                k = k.split("(")[0].replace(",", "")
                tot[(k, r["Counter_Name"])] += float(r["Counter_Value"])
                calls[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("kernel,counter,sum_over_dispatches,dispatches,per_dispatch")
for (k, c), v in sorted(tot.items()):
    n = len(calls[(k, c)])
    print(f"{k},{c},{v:.6g},{n},{v / n:.6g}")
