#!/usr/bin/env python3
"""A/B of the ping-pong gemm_nn kernel (tune knob 0 = 5) against the automatic tile choice, per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from promptir_amd._lib import lib as rawlib  # noqa: E402
from tools.kbench import r, timeit, DEV  # noqa: E402

B = int(os.environ.get("B", "32"))
SHAPES = [("ffn_in L1'", 96, 510, 128), ("qkv L1'", 96, 288, 128), ("proj L1'", 96, 96, 128), ("ffn_out L1'", 255, 96, 128),
          ("ffn_in L1", 48, 254, 128), ("qkv L1", 48, 144, 128),
          ("ffn_in L2", 96, 510, 64), ("qkv L2", 96, 288, 64), ("ffn_in L3", 192, 1020, 32), ("qkv L3", 192, 576, 32),
          ("ffn_out L3", 510, 192, 32), ("ffn_in L4", 384, 2042, 16), ("qkv L4", 384, 1152, 16), ("ffn_out L4", 1021, 384, 16),
          ("ffn_in n3", 704, 3744, 16)]
for name, cin, cout, S in SHAPES:
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    res = r(B, cout, S, S)
    dy = r(B, cout, S, S)
    outs = {}
    line = f"{name:12s} M={cout:4d} K={cin:4d} N={S*S:5d}"
    for knob in (-1, int(os.environ.get("CFG", "5"))):
        rawlib.pir_tune_set(0, knob)
        out = torch.empty(B, cout, S, S, device=DEV)
        ops.conv1x1_forward(x, w, res, out=out)
        dx = torch.empty(B, cin, S, S, device=DEV)
        ops.conv1x1_dgrad(dy, w, out=dx)
        outs[knob] = (out.clone(), dx.clone())
        t = timeit(lambda: ops.conv1x1_forward(x, w, res, out=out))
        t2 = timeit(lambda: ops.conv1x1_dgrad(dy, w, out=dx))
        line += f" | cfg {knob:2d}: fwd {t*1e6:7.1f} dgrad {t2*1e6:7.1f}"
    rawlib.pir_tune_set(0, -1)
    e1 = (outs[-1][0] - outs[int(os.environ.get("CFG", "5"))][0]).abs().max().item()
    e2 = (outs[-1][1] - outs[int(os.environ.get("CFG", "5"))][1]).abs().max().item()
    print(line + f" | maxdiff {e1:.2e} {e2:.2e}", flush=True)
