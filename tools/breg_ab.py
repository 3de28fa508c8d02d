#!/usr/bin/env python3
"""gemm_nn_x3 96x128 tile: activations kept in registers (knob 18 = 1) vs staged through LDS: equality and time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
T = _lib.lib.pir_tune_set
SHAPES = [(510, 96, 128, 1), (288, 96, 128, 0), (255, 96, 128, 1), (96, 96, 128, 1), (510, 96, 64, 1), (96, 288, 64, 0), (192, 576, 32, 0), (576, 192, 32, 0),
          (1020, 192, 32, 1), (192, 192, 32, 1), (384, 1152, 16, 0), (2042, 384, 16, 1), (1152, 384, 16, 0), (704, 2112, 16, 0), (100, 90, 24, 1)]
tot = {0: 0.0, 1: 0.0}
for cin, cout, S, res in SHAPES:
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    res_t = r(B, cout, S, S) if res else None
    out = torch.empty(B, cout, S, S, device="cuda:0")
    fn = lambda: ops.conv1x1_forward(x, w, res_t, out=out)
    T(0, 7)      # the 96 x 128 tile for every shape
    T(18, 0); t0 = timeit(fn); ref = out.clone()
    T(18, 1); out.zero_(); t1 = timeit(fn)
    same = torch.equal(out, ref)
    T(18, -1); T(0, -1)
    tot[0] += t0; tot[1] += t1
    print(f"M={cout:4d} K={cin:4d} N={S*S:5d} R={res}: LDS {t0*1e6:6.1f}  registers {t1*1e6:6.1f}  ({t1/t0:.3f})  identical {same}", flush=True)
print({k: round(v * 1e3, 3) for k, v in tot.items()})
