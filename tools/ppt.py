#!/usr/bin/env python3
"""Phase timing of the instrumented ping-pong kernel (abtest/libt.so): per wave of workgroup 0, cycles per step in
matrix / barrier-after-matrix / stash / issue / read_frags / barrier-after-staging."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r  # noqa: E402

B, cin, cout, S = 32, 704, 3744, 16
_lib.lib.pir_tune_set(0, 5)
x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
out = torch.empty(B, cout, S, S, device="cuda:0")
a3, kp = ops._split_weight(w, dgrad=False)
dbg = torch.zeros(64, device="cuda:0")
hw = S * S
for _ in range(2):
    ops.gemm_nn(dbg, (0, 0), cin, 1, x, 0, (cin * hw, 0), hw, out, 0, (cout * hw, 0), hw, cout, cin, hw, B, 1, A3=a3, a3_kp=kp)
torch.cuda.synchronize()
d = dbg.cpu().view(8, 8)
print("wave  matrix  bar_m  stash  issue  reads  bar_s  vmwait  (cycles per step; steps =", int(d[0, 7]), ")")
for wv in range(8):
    n = max(float(d[wv, 7]), 1.0)
    print(f"{wv:4d} " + " ".join(f"{float(d[wv, i]) / n:7.0f}" for i in range(7)))
