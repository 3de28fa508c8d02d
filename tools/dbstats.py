#!/usr/bin/env python3
"""Per-kernel statistics (rocprofv3 --stats layout) from a rocprofv3 results .db -> CSV on stdout."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, end - start from kernels").fetchall()
agg = {}
for name, dur in rows:
    name = re.sub(r"\(.*$", "", name) if len(sys.argv) > 2 and sys.argv[2] == "--short" else name
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
total = sum(a[1] for a in agg.values())
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'"{name}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100.0 * a[1] / total:.2f},{a[2]},{a[3]}')
