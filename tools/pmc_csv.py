#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (tools/pmc_run.sh): per kernel, mean counter value per dispatch and the mean
dispatch duration.  usage: pmc_csv.py <outdir> [kernel-name substring ...]"""
import csv
import glob
import os
import re
import sys

out, filt = sys.argv[1], sys.argv[2:]
agg, dur = {}, {}
for path in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*$", "", k)[:120]
        if filt and not any(f in k for f in filt):
            continue
        a = agg.setdefault((k, row["Counter_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += float(row["Counter_Value"])
        if "Start_Timestamp" in row and row.get("End_Timestamp"):
            d = dur.setdefault(k, [0, 0.0])
            d[0] += 1
            d[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
for k in sorted({k for k, _ in agg}):
    n, t = dur.get(k, [1, 0.0])
    print(f"== {k[:150]}  (mean dispatch {t / max(n, 1):.1f} us under the profiler)")
    for (kk, c), (m, tot) in sorted(agg.items()):
        if kk == k:
            print(f"   {c:34s} n={m:4d} mean={tot / m:18.1f}")
