#!/usr/bin/env python3
"""What one train step launches, from a rocprofv3 kernel trace of bench.py (graph step, default streams).

    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 5 --warmup 1 --no-legs --config5 0 ...
    python tools/step_kernels.py <dir> [label]          -> JSON on stdout

A step = the launches between two consecutive `adamw_kernel` launches (the optimiser closes every train step); the
steps whose launch count equals the most frequent count are the steady-state graph steps.  Reports launches per step,
how many of them run under 10 us (and their summed time), the summed kernel time, and every kernel that does NOT come
from libpromptir_hip.so (runtime blits, ATen kernels; RCCL at N > 1) with its count per step."""
import collections
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else "step"
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if not rows:
    sys.exit("no kernel trace rows under " + d)


def short(name):
    return re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:90]


def library(name):
    return "(anonymous namespace)::" in name      # every kernel of libpromptir_hip.so lives in an anonymous namespace


cuts = [i for i, r in enumerate(rows) if short(r[2]).startswith("adamw_kernel")]
steps = [rows[a + 1:b + 1] for a, b in zip(cuts, cuts[1:])]
if not steps:
    sys.exit("fewer than two optimiser launches in the trace")
mode = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
steady = [s for s in steps if len(s) == mode]
out = {"label": label, "steps_in_trace": len(steps), "steady_steps": len(steady), "launches_per_step": mode}
n = len(steady)
dur = [[(e - s) * 1e-3 for s, e, _ in st] for st in steady]
out["kernel_time_ms_per_step"] = round(sum(sum(x) for x in dur) / n / 1e3, 3)
out["under_10us_per_step"] = round(sum(sum(1 for v in x if v < 10.0) for x in dur) / n, 1)
out["under_10us_share_of_launches"] = round(out["under_10us_per_step"] / mode, 4)
out["under_10us_ms_per_step"] = round(sum(sum(v for v in x if v < 10.0) for x in dur) / n / 1e3, 3)
out["wall_ms_per_step"] = round(sum((st[-1][1] - st[0][0]) for st in steady) / n * 1e-6, 3)
foreign = collections.Counter()
byname = collections.Counter()
tname = collections.Counter()
for st in steady:
    for s0, e0, name in st:
        byname[short(name).split("<")[0]] += 1
        tname[short(name)] += (e0 - s0) * 1e-6
        if not library(name):
            foreign[short(name)] += 1
out["non_library_kernels_per_step"] = {k: round(v / n, 2) for k, v in foreign.most_common()}
out["launches_by_kernel_per_step"] = {k: round(v / n, 1) for k, v in byname.most_common(40)}
out["kernel_ms_per_step_by_instantiation"] = {k: round(v / n, 3) for k, v in tname.most_common(45)}
print(json.dumps(out, indent=1))
