#!/usr/bin/env python3
"""gemm_nt tile-configuration sweep (knob 1) on the 1x1 weight gradients of every level: auto vs every fixed config."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import LEVELS, r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
T = _lib.lib.pir_tune_set
NAMES = {-1: "auto", 0: "64x64", 1: "128x64", 2: "128x96", 3: "128x128", 4: "256x96", 5: "256x128"}
tot = {}
ONLY = os.environ.get("ONLY", "")
for name, C, S, heads in LEVELS:
    if ONLY and ONLY not in name:
        continue
    hid = int(C * 2.66)
    for tag, cin, cout in (("qkv", C, 3 * C), ("proj", C, C), ("ffn_in", C, 2 * hid), ("ffn_out", hid, C)):
        x, dy, w = r(B, cin, S, S), r(B, cout, S, S), r(cout, cin, 1, 1)
        out = torch.empty_like(w)
        fn = lambda: ops.conv1x1_wgrad(dy, x, w, out=out)
        row, best = [], None
        T(1, 3)
        fn()
        ref = out.clone()
        for cfg in (-1, 0, 1, 2, 3, 4, 5):
            T(1, cfg)
            out.zero_()
            fn()
            err = float((out - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
            assert err <= 2e-5, (name, tag, cfg, err)      # a tile plan may only change the summation order
            t = timeit(fn)
            row.append(f"{NAMES[cfg]} {t*1e6:6.1f}")
            if cfg >= 0 and (best is None or t < best[0]):
                best = (t, cfg)
            if cfg == -1:
                tot["auto"] = tot.get("auto", 0.0) + t
        T(1, -1)
        tot["best"] = tot.get("best", 0.0) + best[0]
        print(f"{name:18s} wgrad {tag:8s} M1={cout:4d} M2={cin:4d} N={S*S*B:7d}: " + " | ".join(row) + f" | best {NAMES[best[1]]}", flush=True)
print({k: round(v * 1e3, 3) for k, v in tot.items()})
