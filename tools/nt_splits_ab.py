#!/usr/bin/env python3
"""gemm_nt split-K sweep (knob 2) on the low-resolution weight gradients: automatic split count vs fixed ones."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.resident_ab import timeit, r  # noqa: E402

B = int(os.environ.get("B", "16"))
T = _lib.lib.pir_tune_set
for C, S in ((192, 32), (384, 16), (704, 16), (320, 32), (96, 64)):
    hid = int(C * 2.66)
    for tag, cin, cout in (("qkv", C, 3 * C), ("proj", C, C), ("ffn_in", C, 2 * hid), ("ffn_out", hid, C)):
        x, dy, w = r(B, cin, S, S), r(B, cout, S, S), r(cout, cin, 1, 1)
        out = torch.empty_like(w)
        fn = lambda: ops.conv1x1_wgrad(dy, x, w, out=out)
        fns, names = [], []
        for sp in (0, 4, 8, 12, 16, 24, 32, 48, 64, 96):
            def f(sp=sp):
                T(2, sp)
                fn()
            fns.append(f); names.append("auto" if sp == 0 else str(sp))
        ts = timeit(fns, rounds=5, inner=3)
        T(2, 0)
        best = min(range(len(ts)), key=lambda i: ts[i])
        print(f"C{C} {S}^2 B={B} wgrad {tag:8s} {cout:4d}x{cin:4d}: " + " | ".join(f"{n} {t*1e6:6.1f}" for n, t in zip(names, ts)) +
              f" | best {names[best]} ({ts[best]/ts[0]:.2f} of auto)", flush=True)
