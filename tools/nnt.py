#!/usr/bin/env python3
"""Phase timing of the instrumented default gemm_nn kernel (abtest/libt.so, built from a patched copy): cycles per
k-step of workgroup 0's waves in: load issue / fragment reads / MFMA issue / vmcnt wait / split+LDS writes / barrier."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.kbench import r  # noqa: E402

B = 32
for cin, cout, S in ((384, 2042, 16), (704, 3744, 16), (96, 510, 128), (510, 96, 128)):
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    out = torch.empty(B, cout, S, S, device="cuda:0")
    a3, kp = ops._split_weight(w, dgrad=False)
    dbg = torch.zeros(64, device="cuda:0")
    hw = S * S
    for _ in range(2):
        ops.gemm_nn(dbg, (0, 0), cin, 1, x, 0, (cin * hw, 0), hw, out, 0, (cout * hw, 0), hw, cout, cin, hw, B, 1, A3=a3, a3_kp=kp)
    torch.cuda.synchronize()
    d = dbg.cpu().view(8, 8)
    print(f"M={cout} K={cin} N={hw}: k-steps {int(d[0, 6])}; cycles per k-step: issue  reads  mfma  vmwait  stash  barrier")
    for wv in range(4):
        n = max(float(d[wv, 6]), 1.0)
        print(f"  wave {wv}: " + " ".join(f"{float(d[wv, i]) / n:7.0f}" for i in range(6)) + f"   total {float(d[wv, :6].sum()) / n:7.0f}")
