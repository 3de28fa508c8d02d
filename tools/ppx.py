#!/usr/bin/env python3
"""Time the ping-pong gemm_nn kernel (tune knob 0 = 5) on a few shapes; build selected by PIR_LIB."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from promptir_amd._lib import lib as rawlib  # noqa: E402
from tools.kbench import r, timeit, DEV  # noqa: E402

B = 32
tag = os.environ.get("PIR_LIB", "default").split("/")[-1]
rawlib.pir_tune_set(0, int(os.environ.get("CFG", "5")))
line = f"{tag:14s}"
for name, cin, cout, S in (("n3", 704, 3744, 16), ("L4", 384, 2042, 16), ("L3", 192, 1020, 32), ("L1'", 96, 510, 128)):
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    out = torch.empty(B, cout, S, S, device=DEV)
    t = timeit(lambda: ops.conv1x1_forward(x, w, None, out=out))
    line += f" {name} {t*1e6:7.1f}"
print(line, flush=True)
