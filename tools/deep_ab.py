#!/usr/bin/env python3
"""A/B of one knob (KNOB / V0 / V1 from the environment; default knob 43: 32 x 128 tiles for underfilled launches, 0 against 100) on the 1x1 convolutions of the 16^2 / 32^2 / 64^2 levels:
forward and input gradient, part batches of 1 / 4 / 16 images."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.resident_ab import r, timeit  # noqa: E402

T = _lib.lib.pir_tune_set
KNOB, V0, V1 = int(os.environ.get('KNOB', '43')), int(os.environ.get('V0', '0')), int(os.environ.get('V1', '100'))
for B in (1, 4, 16):
    print(f"batch {B}\n{'shape':36s} | {'v0 us':>11s} {'v1 us':>11s} {'ratio':>6s}")
    tot = [0.0, 0.0]
    for C, S in ((96, 64), (192, 32), (384, 16)):
        hid = int(C * 2.66)
        for tag, cin, cout, dgrad in (("qkv fwd", C, 3 * C, False), ("ffn_in fwd", C, 2 * hid, False), ("ffn_out fwd", hid, C, False),
                                      ("proj fwd", C, C, False), ("ffn_out dgrad", hid, C, True), ("qkv dgrad", C, 3 * C, True),
                                      ("ffn_in dgrad", C, 2 * hid, True)):
            w = r(cout, cin, 1, 1)
            x = r(B, cout if dgrad else cin, S, S)
            out = torch.empty(B, cin if dgrad else cout, S, S, device="cuda:0")

            def run(v):
                T(KNOB, v)
                if dgrad:
                    ops.conv1x1_dgrad(x, w, out=out)
                else:
                    ops.conv1x1_forward(x, w, None, out=out)

            t0, t1 = timeit([lambda: run(V0), lambda: run(V1)])
            tot[0] += t0; tot[1] += t1
            print(f"C{C} {S}^2 {tag:14s} K={(cout if dgrad else cin):4d} M={(cin if dgrad else cout):4d} | {t0*1e6:11.1f} {t1*1e6:11.1f} {t1/t0:6.2f}", flush=True)
    print("sum %.1f -> %.1f us" % (tot[0] * 1e6, tot[1] * 1e6))
T(KNOB, V0)
