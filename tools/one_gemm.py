#!/usr/bin/env python3
"""One 1x1-convolution GEMM shape, repeated: target for rocprofv3 counter passes.
    python tools/one_gemm.py COUT CIN SIDE [--res 0|1] [--dgrad] [--batch 32] [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("cout", type=int); ap.add_argument("cin", type=int); ap.add_argument("side", type=int)
ap.add_argument("--res", type=int, default=-1); ap.add_argument("--dgrad", action="store_true")
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--tm", type=int, default=0); ap.add_argument("--wgs", type=int, default=0); ap.add_argument("--bst", type=int, default=-1)
a = ap.parse_args()
_lib.lib.pir_tune_set(20, a.res); _lib.lib.pir_tune_set(21, a.tm); _lib.lib.pir_tune_set(23, a.wgs); _lib.lib.pir_tune_set(24, a.bst)
dev = "cuda:0"
w = torch.randn(a.cout, a.cin, 1, 1, device=dev)
if a.dgrad:
    x = torch.randn(a.batch, a.cout, a.side, a.side, device=dev)
    out = torch.empty(a.batch, a.cin, a.side, a.side, device=dev)
    fn = lambda: ops.conv1x1_dgrad(x, w, out=out)
else:
    x = torch.randn(a.batch, a.cin, a.side, a.side, device=dev)
    out = torch.empty(a.batch, a.cout, a.side, a.side, device=dev)
    fn = lambda: ops.conv1x1_forward(x, w, None, out=out)
for _ in range(a.iters):
    fn()
torch.cuda.synchronize()
