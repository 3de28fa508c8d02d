#!/usr/bin/env python3
"""Times the 1x1 convolutions WITH a residual (project_out of both branches, every level) with the library named by PIR_LIB."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
SHAPES = [(96, 96, 128), (255, 96, 128), (48, 48, 128), (127, 48, 128), (96, 96, 64), (255, 96, 64), (192, 192, 32), (510, 192, 32),
          (384, 384, 16), (1021, 384, 16)]
cells, tot = [], 0.0
for cin, cout, S in SHAPES:
    x, w, res = r(B, cin, S, S), r(cout, cin, 1, 1), r(B, cout, S, S)
    out = torch.empty(B, cout, S, S, device="cuda:0")
    t = timeit(lambda: ops.conv1x1_forward(x, w, res, out=out), rounds=7, inner=5)
    tot += t
    cells.append(f"{cout}x{cin}x{S*S}: {t*1e6:6.1f}")
print(os.environ.get("PIR_LIB", "product"), " | ".join(cells), f"| sum {tot*1e6:.0f}", flush=True)
