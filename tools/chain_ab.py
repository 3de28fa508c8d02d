#!/usr/bin/env python3
"""gemm_nn_x3 chained tiles (knob 12 = tiles per workgroup) vs one tile per workgroup: equality and time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
T = _lib.lib.pir_tune_set
SHAPES = [(96, 510, 128, 0), (510, 96, 128, 1), (96, 288, 128, 0), (255, 96, 128, 1), (96, 96, 128, 1), (48, 254, 128, 0), (127, 48, 128, 1),
          (96, 510, 64, 0), (510, 96, 64, 1), (192, 1020, 32, 0), (1020, 192, 32, 1), (192, 576, 32, 0), (510, 192, 32, 1),
          (384, 2042, 16, 0), (2042, 384, 16, 1), (384, 1152, 16, 0), (100, 70, 24, 1)]
CH = (2, 3, 4, 8, 16)
tot = {}
for cin, cout, S, res in SHAPES:
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    res_t = r(B, cout, S, S) if res else None
    out = torch.empty(B, cout, S, S, device="cuda:0")
    fn = lambda: ops.conv1x1_forward(x, w, res_t, out=out)
    T(12, 0)
    t0 = timeit(fn)
    ref = out.clone()
    cells, best = [], t0
    for c in CH:
        T(12, c)
        out.zero_()
        t1 = timeit(fn)
        err = (out - ref).abs().max().item()
        cells.append(f"x{c} {t1*1e6:6.1f}" + ("" if err == 0 else f" ERR {err:.1e}"))
        tot[c] = tot.get(c, 0) + t1
        best = min(best, t1)
    T(12, 0)
    tot[1] = tot.get(1, 0) + t0
    tot["best"] = tot.get("best", 0) + best
    print(f"M={cout:4d} K={cin:4d} N={S*S:5d} R={res}: single {t0*1e6:6.1f} | " + " | ".join(cells), flush=True)
print({k: round(v * 1e3, 3) for k, v in tot.items()})
