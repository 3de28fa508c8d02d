#!/usr/bin/env python3
"""gemm_nn_x3 pipeline-depth A/B (knob 5) per tile configuration (knob 0) on the 1x1-convolution shapes of the step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
T = _lib.lib.pir_tune_set
CFG = {-1: "auto", 3: "128x128", 4: "64x128", 7: "96x128"}
SHAPES = [(96, 510, 128, 0), (510, 96, 128, 1), (96, 288, 128, 0), (255, 96, 128, 1), (96, 96, 128, 1), (48, 254, 128, 0),
          (96, 510, 64, 0), (510, 96, 64, 1), (192, 1020, 32, 0), (1020, 192, 32, 1), (192, 576, 32, 0), (510, 192, 32, 1),
          (384, 2042, 16, 0), (2042, 384, 16, 1), (384, 1152, 16, 0)]
tot = {}
for cin, cout, S, res in SHAPES:
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    res_t = r(B, cout, S, S) if res else None
    out = torch.empty(B, cout, S, S, device="cuda:0")
    fn = lambda: ops.conv1x1_forward(x, w, res_t, out=out)
    T(0, -1); T(5, 0)
    base = timeit(fn)
    cells, best = [], (base, "auto/2")
    for cfg in (3, 7, 4):
        if cfg == 7 and cout > 96 and cout % 96:
            continue
        for d in (0, 4, 6):
            T(0, cfg); T(5, d)
            t = timeit(fn)
            cells.append(f"{CFG[cfg]}/d{d or 2} {t*1e6:6.1f}")
            if t < best[0]:
                best = (t, f"{CFG[cfg]}/d{d or 2}")
    T(0, -1); T(5, 0)
    tot["auto"] = tot.get("auto", 0) + base
    tot["best"] = tot.get("best", 0) + best[0]
    print(f"M={cout:4d} K={cin:4d} N={S*S:5d} R={res}: auto {base*1e6:6.1f} | " + " | ".join(cells) + f" | best {best[1]} {best[0]*1e6:.1f}", flush=True)
print({k: round(v * 1e3, 3) for k, v in tot.items()})
