#!/usr/bin/env python3
"""Times the 1x1-convolution GEMMs with the library named by PIR_LIB (abtest/libabl<N>.so: one pipeline component
removed, results are garbage) - what a component costs on the critical path = time(0) - time(N)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
SHAPES = [(96, 510, 128, 0), (510, 96, 128, 0), (96, 288, 128, 0), (96, 96, 128, 0), (96, 510, 64, 0), (192, 1020, 32, 0),
          (1020, 192, 32, 0), (384, 2042, 16, 0), (2042, 384, 16, 0)]
out_line = []
for cin, cout, S, res in SHAPES:
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    out = torch.empty(B, cout, S, S, device="cuda:0")
    t = timeit(lambda: ops.conv1x1_forward(x, w, None, out=out), rounds=7, inner=5)
    out_line.append(f"{cout}x{cin}x{S*S}: {t*1e6:6.1f}")
print(os.environ.get("PIR_LIB", "product"), " | ".join(out_line), flush=True)
