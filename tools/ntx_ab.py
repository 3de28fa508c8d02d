#!/usr/bin/env python3
"""A/B of the X-private weight-gradient kernel (gemm_ntx.hip, knob 25) against the tiled gemm_nt on the 1x1 weight
gradients of a batch-B 128x128 train step: agreement (summation order differs: tolerance) and time per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.resident_ab import timeit, r  # noqa: E402

B = int(os.environ.get("B", "32"))
T = _lib.lib.pir_tune_set
tot = [0.0, 0.0]
print(f"{'shape':34s} | {'tiled us':>9s} {'xp us':>9s} {'ratio':>6s} | {'GB/s':>7s} {'TF/s':>7s} | rel.err")
for C, S in ((48, 128), (96, 128), (96, 64), (192, 32), (384, 16)):
    hid = int(C * 2.66)
    for tag, cin, cout in (("qkv", C, 3 * C), ("proj", C, C), ("ffn_in", C, 2 * hid), ("ffn_out", hid, C)):
        x, dy, w = r(B, cin, S, S), r(B, cout, S, S), r(cout, cin, 1, 1)
        outs = [torch.empty_like(w), torch.empty_like(w)]

        def tiled():
            T(25, 0)
            ops.conv1x1_wgrad(dy, x, w, out=outs[0])

        def xp():
            T(25, 1)
            ops.conv1x1_wgrad(dy, x, w, out=outs[1])

        tiled(); xp()
        torch.cuda.synchronize()
        err = float((outs[0] - outs[1]).abs().max()) / max(float(outs[0].abs().max()), 1e-30)
        t0, t1 = timeit([tiled, xp])
        tot[0] += t0; tot[1] += t1
        by = 4.0 * S * S * B * (cin + cout)
        fl = 2.0 * cin * cout * S * S * B
        print(f"C{C} {S}^2 wgrad {tag:8s} {cout:4d}x{cin:4d} | {t0*1e6:9.1f} {t1*1e6:9.1f} {t1/t0:6.2f} | {by/t1/1e9:7.0f} {fl/t1/1e12:7.1f} | {err:.1e}",
              flush=True)
T(25, -1)
print("sum tiled %.3f ms, xp %.3f ms" % (tot[0] * 1e3, tot[1] * 1e3))
