#!/usr/bin/env python3
"""Runs bench.main() with pir_tune_set knobs applied first: tools/knob_bench.py KNOB=VALUE [...] -- [bench.py args]."""
import contextlib
import io
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (first: the library must bind to the HIP runtime torch loads)
from promptir_amd import _lib  # noqa: E402

args = sys.argv[1:]
rest = args[args.index("--") + 1:] if "--" in args else []
knobs = [a for a in (args[:args.index("--")] if "--" in args else args)]
for kv in knobs:
    k, v = kv.split("=")
    assert _lib.lib.pir_tune_set(int(k), int(v)) == 0, kv
sys.argv = ["bench.py"] + rest
import bench  # noqa: E402

buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(" ".join(knobs) or "default", "ms_per_step", d["ms_per_step"], "median", d["ms_per_step_median"], "value", d["value"], flush=True)
