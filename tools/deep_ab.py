#!/usr/bin/env python3
"""A/B of knob 42 (four register stages in the 96 x 128 BREG tile) on the 1x1 convolutions of the 16^2 / 32^2 / 64^2 levels:
forward and input gradient, part batches of 1 / 4 / 16 images."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.resident_ab import r, timeit  # noqa: E402

T = _lib.lib.pir_tune_set
for B in (1, 4, 16):
    print(f"batch {B}\n{'shape':36s} | {'2 stages us':>11s} {'4 stages us':>11s} {'ratio':>6s}")
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
                T(42, v)
                if dgrad:
                    ops.conv1x1_dgrad(x, w, out=out)
                else:
                    ops.conv1x1_forward(x, w, None, out=out)

            t0, t1 = timeit([lambda: run(0), lambda: run(1)])
            tot[0] += t0; tot[1] += t1
            print(f"C{C} {S}^2 {tag:14s} K={(cout if dgrad else cin):4d} M={(cin if dgrad else cout):4d} | {t0*1e6:11.1f} {t1*1e6:11.1f} {t1/t0:6.2f}", flush=True)
    print("sum %.1f -> %.1f us" % (tot[0] * 1e6, tot[1] * 1e6))
T(42, -1)
