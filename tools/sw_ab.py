#!/usr/bin/env python3
"""A/B of the register-only stencils (stencil_wave.hip, knob 8 = 0) against the LDS-tiled ones (knob 8 = 1), with
sweeps of columns per lane (knob 10) and band height (knob 9); interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
T = _lib.lib.pir_tune_set
for name, C, S in (("L1' C96 128^2", 96, 128), ("L1 C48 128^2", 48, 128), ("L2 C96 64^2", 96, 64), ("L3 C192 32^2", 192, 32),
                   ("L4 C384 16^2", 384, 16)):
    hid = int(C * 2.66)
    x, w = r(B, 3 * C, S, S), r(3 * C, 1, 3, 3)
    y = torch.empty_like(x)
    x2, w2 = r(B, 2 * hid, S, S), r(2 * hid, 1, 3, 3)
    for label, fn, by in (("dw fwd", lambda: ops.dwconv_forward(x, w, out=y), 8.0 * x.numel()),
                          ("dw bwd", lambda: ops.dwconv_backward(y, x, w), 12.0 * x.numel()),
                          ("gate fwd", lambda: ops.dwconv_gate_forward(x2, w2), 6.0 * x2.numel())):
        res = []
        for tag, off, vec, rb in (("lds", 1, 0, 0), ("auto", 0, 0, 0), ("v4", 0, 4, 0), ("v2", 0, 2, 0), ("v1", 0, 1, 0),
                                  ("rb16", 0, 0, 16), ("rb32", 0, 0, 32), ("rb64", 0, 0, 64)):
            if rb > S or (vec and S // vec > 64):
                continue
            T(8, off); T(10, vec); T(9, rb)
            t = timeit(fn)
            res.append(f"{tag} {t * 1e6:6.1f}us {by / t / 1e9:5.0f}")
        T(8, 0); T(10, 0); T(9, 0)
        print(f"B={B} {name:14s} {label:8s} | " + " | ".join(res), flush=True)
