#!/usr/bin/env python3
"""Times the 1x1 weight-gradient GEMMs (gemm_nt_x3 + its split-K reduction) with the library named by PIR_LIB
(abtest/libnt<N>.so: one pipeline component removed, results are garbage)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
SHAPES = [(96, 510, 128), (96, 288, 128), (96, 96, 128), (255, 96, 128), (96, 510, 64), (192, 1020, 32), (192, 576, 32), (384, 2042, 16), (384, 1152, 16)]
cells = []
for cin, cout, S in SHAPES:
    x, dy, w = r(B, cin, S, S), r(B, cout, S, S), r(cout, cin, 1, 1)
    out = torch.empty_like(w)
    t = timeit(lambda: ops.conv1x1_wgrad(dy, x, w, out=out), rounds=7, inner=5)
    cells.append(f"{cout}x{cin}x{S*S*B}: {t*1e6:6.1f}")
print(os.environ.get("PIR_LIB", "product"), " | ".join(cells), flush=True)
