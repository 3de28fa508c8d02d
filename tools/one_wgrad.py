#!/usr/bin/env python3
"""One 1x1 weight gradient, repeated: target for rocprofv3 counter passes.
    python tools/one_wgrad.py COUT CIN SIDE [--ln] [--batch 16] [--iters 6] [--knob K=V ...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("cout", type=int); ap.add_argument("cin", type=int); ap.add_argument("side", type=int)
ap.add_argument("--ln", action="store_true"); ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--iters", type=int, default=6); ap.add_argument("--knob", action="append", default=[])
a = ap.parse_args()
for kv in a.knob:
    k, v = kv.split("=")
    _lib.lib.pir_tune_set(int(k), int(v))
dev = "cuda:0"
x = torch.randn(a.batch, a.cin, a.side, a.side, device=dev)
dy = torch.randn(a.batch, a.cout, a.side, a.side, device=dev)
w = torch.empty(a.cout, a.cin, 1, 1, device=dev)
out = torch.empty_like(w)
if a.ln:
    gam, bet = torch.randn(a.cin, device=dev), torch.randn(a.cin, device=dev)
    _, mean, rstd = ops.layernorm_forward(x, gam, bet)
    fn = lambda: ops.conv1x1_wgrad_ln(dy, x, mean, rstd, gam, bet, w, out=out)
else:
    fn = lambda: ops.conv1x1_wgrad(dy, x, w, out=out)
for _ in range(a.iters):
    fn()
torch.cuda.synchronize()
