#!/usr/bin/env python3
"""Ping-pong gemm_nn kernel on one MFMA-bound shape, 3 launches, for rocprofv3 --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from promptir_amd._lib import lib as rawlib  # noqa: E402
from tools.kbench import r  # noqa: E402

B, DEV = 32, "cuda:0"
rawlib.pir_tune_set(0, int(os.environ.get("CFG", "5")))
for cin, cout, S in ((704, 3744, 16),):
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    out = torch.empty(B, cout, S, S, device=DEV)
    for _ in range(3):
        ops.conv1x1_forward(x, w, None, out=out)
    torch.cuda.synchronize()
