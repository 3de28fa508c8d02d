#!/usr/bin/env python3
"""96x128 tile (tune knob 0 = 7) vs the automatic choice on the full-resolution 1x1 convolutions."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from promptir_amd._lib import lib as rawlib  # noqa: E402
from tools.kbench import r, timeit, DEV  # noqa: E402

B = 32
for name, cin, cout, S in (("ffn_in L1'", 96, 510, 128), ("qkv L1'", 96, 288, 128), ("proj L1'", 96, 96, 128), ("ffn_out L1'", 255, 96, 128),
                           ("ffn_in L1", 48, 254, 128), ("qkv L1", 48, 144, 128), ("ffn_in L2", 96, 510, 64), ("qkv L2", 96, 288, 64)):
    x, w, dy = r(B, cin, S, S), r(cout, cin, 1, 1), r(B, cout, S, S)
    line = f"{name:12s}"
    for knob in (-1, 7):
        rawlib.pir_tune_set(0, knob)
        out = torch.empty(B, cout, S, S, device=DEV); dx = torch.empty(B, cin, S, S, device=DEV)
        t = timeit(lambda: ops.conv1x1_forward(x, w, None, out=out))
        t2 = timeit(lambda: ops.conv1x1_dgrad(dy, w, out=dx))
        line += f" | cfg {knob:2d}: fwd {t*1e6:7.1f} dgrad {t2*1e6:7.1f}"
    rawlib.pir_tune_set(0, -1)
    print(line, flush=True)
