#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration table from a counter pass over tools/probes/fetch_calib.bin:
    python tools/fetch_calib_summary.py <rocprof out dir> [<second dir with the other counter>]
Prints, per access pattern, counter KB x 1024 / bytes really moved (1 GiB per kernel)."""
import csv
import glob
import os
import re
import sys

BYTES = float(1 << 30)
rows = {}
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", ""))
            rows.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(f"{'kernel':34s} {'counter':12s} {'KB':>14s} {'x1024/bytes':>12s}")
for k in sorted(rows):
    for c, v in sorted(rows[k].items()):
        kb = sum(v) / len(v)
        print(f"{k:34s} {c:12s} {kb:14.0f} {kb * 1024 / BYTES:12.3f}")
