#!/usr/bin/env python3
"""gemm_nn_x3: transposed accumulators + 16-byte stores (knob 11) vs the 4-byte C-layout epilogue: equality and time."""
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
tot = {}
for cin, cout, S, res in SHAPES:
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    res_t = r(B, cout, S, S) if res else None
    out = torch.empty(B, cout, S, S, device="cuda:0")
    fn = lambda: ops.conv1x1_forward(x, w, res_t, out=out)
    T(11, 0)
    t0 = timeit(fn)
    ref = out.clone()
    T(11, 1)
    t1 = timeit(fn)
    err = (out - ref).abs().max().item()
    T(11, -1)
    tot["c"] = tot.get("c", 0) + t0
    tot["t"] = tot.get("t", 0) + t1
    print(f"M={cout:4d} K={cin:4d} N={S*S:5d} R={res}: C-layout {t0*1e6:6.1f}  transposed {t1*1e6:6.1f}  ({t1/t0:.3f})  max|diff| {err:.1e}", flush=True)
print({k: round(v * 1e3, 3) for k, v in tot.items()})
