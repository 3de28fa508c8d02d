#!/usr/bin/env python3
"""A/B of one tuning knob on the forward 1x1 convolutions of a part batch: plain (pir_gemm_nn) and with the LayerNorm applied
on load (pir_ln_conv1x1_fwd, statistics out).  Bit-equality of the two settings and time per call.
    B=16 KNOB=41 V0=0 V1=1 python tools/knob_fwd_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.resident_ab import r, timeit  # noqa: E402

B = int(os.environ.get("B", "16"))
KNOB, V0, V1 = int(os.environ.get("KNOB", "41")), int(os.environ.get("V0", "0")), int(os.environ.get("V1", "1"))
T = _lib.lib.pir_tune_set
print(f"knob {KNOB}: {V0} against {V1}, batch {B}\n{'shape':34s} | {'v0 us':>9s} {'v1 us':>9s} {'ratio':>6s} | equal")
for c, S in ((48, 128), (96, 128), (96, 64)):
    hid = int(c * 2.66)
    for tag, M, ln in (("qkv+ln", 3 * c, True), ("ffn_in+ln", 2 * hid, True), ("qkv", 3 * c, False), ("ffn_in", 2 * hid, False)):
        x, w, gam, bet = r(B, c, S, S), r(M, c, 1, 1), r(c), r(c)

        def run(v):
            T(KNOB, v)
            if ln:
                return ops.ln_conv1x1_forward(x, gam, bet, w, stats=True)
            return ops.conv1x1_forward(x, w)

        a, b = run(V0), run(V1)
        if a is None:
            continue
        ya, yb = (a[0], b[0]) if ln else (a, b)
        torch.cuda.synchronize()
        eq = bool(torch.equal(ya, yb))
        t0, t1 = timeit([lambda: run(V0), lambda: run(V1)])
        print(f"C{c} {S}^2 {tag:10s} M={M:4d}       | {t0*1e6:9.1f} {t1*1e6:9.1f} {t1/t0:6.2f} | {eq}", flush=True)
T(KNOB, V0)
