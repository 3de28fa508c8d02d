#!/usr/bin/env python3
"""One 1x1-convolution GEMM shape, a few launches, for rocprofv3 --pmc passes (tools/pmc_run.sh).
   SHAPE=cin,cout,S  MODE=fwd|dgrd|wgrd  CFG=<tile cfg or -1>  B=32"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r  # noqa: E402

B = int(os.environ.get("B", "32"))
cin, cout, S = (int(v) for v in os.environ.get("SHAPE", "96,510,128").split(","))
_lib.lib.pir_tune_set(0, int(os.environ.get("CFG", "-1")))
x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
out = torch.empty(B, cout, S, S, device="cuda:0")
dy, dx = r(B, cout, S, S), torch.empty(B, cin, S, S, device="cuda:0")
for _ in range(int(os.environ.get("REPS", "4"))):
    mode = os.environ.get("MODE", "fwd")
    if mode == "fwd":
        ops.conv1x1_forward(x, w, None, out=out)
    elif mode == "wgrd":
        ops.conv1x1_wgrad(dy, x, w)
    else:
        ops.conv1x1_dgrad(dy, w, out=dx)
torch.cuda.synchronize()
