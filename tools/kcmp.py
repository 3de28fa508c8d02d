#!/usr/bin/env python3
"""Side-by-side of kbench logs: python tools/kcmp.py a.log b.log ... (times in us)."""
import re, sys
tabs = []
for f in sys.argv[1:]:
    d = {}
    for line in open(f):
        m = re.match(r"^(.*?):\s+([0-9.]+) us", line)
        if m:
            d[re.sub(r"\s+", " ", m.group(1))] = float(m.group(2))
    tabs.append(d)
keys = [k for k in tabs[0] if all(k in t for t in tabs)]
tot = [0.0] * len(tabs)
for k in keys:
    vals = [t[k] for t in tabs]
    for i, v in enumerate(vals):
        tot[i] += v
    print(f"{k[:70]:70s} " + " ".join(f"{v:9.1f}" for v in vals))
print(f"{'TOTAL':70s} " + " ".join(f"{v:9.1f}" for v in tot))
