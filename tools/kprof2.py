#!/usr/bin/env python3
"""Two gemm_nn shapes, 3 launches each, for rocprofv3 --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.kbench import r  # noqa: E402

B, DEV = 32, "cuda:0"
for cin, cout, S in ((96, 510, 128), (384, 2042, 16)):
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    out = torch.empty(B, cout, S, S, device=DEV)
    for _ in range(3):
        ops.conv1x1_forward(x, w, None, out=out)
    torch.cuda.synchronize()
