#!/usr/bin/env python3
"""A/B of the C-stationary gemm_nn (gemm_cst.hip, knob 26) against the automatic choice without it on the M = 96 / long-k
1x1 convolution shapes of the train step: bit-equality of the results and time per call.

    python tools/cst_ab.py [--batch 32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.resident_ab import r, timeit  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    B = args.batch
    L = _lib.lib
    shapes = []   # (tag, cout, cin, side, residual, dgrad)
    for C, S in ((48, 128), (96, 128), (96, 64), (192, 32)):
        hid = int(C * 2.66)
        shapes += [(f"C{C} {S}^2 ffn_out fwd+R", C, hid, S, True, False), (f"C{C} {S}^2 qkv dgrad", 3 * C, C, S, False, True),
                   (f"C{C} {S}^2 ffn_in dgrad", 2 * hid, C, S, False, True)]
    print(f"batch {B}\n{'shape':28s} {'M':>5s} {'K':>5s} {'N':>6s} | {'auto us':>9s} {'cst us':>9s} {'ratio':>6s} | {'GB/s':>8s} {'TF/s':>8s} | plan equal")
    for tag, cout, cin, S, res, dgrad in shapes:
        w = r(cout, cin, 1, 1)
        if dgrad:
            x, M, K = r(B, cout, S, S), cin, cout
            out = [torch.empty(B, cin, S, S, device=DEV) for _ in range(2)]
            call = lambda o: ops.conv1x1_dgrad(x, w, out=o)
        else:
            x, M, K = r(B, cin, S, S), cout, cin
            rt = r(B, cout, S, S) if res else None
            out = [torch.empty(B, cout, S, S, device=DEV) for _ in range(2)]
            call = lambda o: ops.conv1x1_forward(x, w, rt, out=o)

        def auto():
            L.pir_tune_set(26, 0)
            call(out[0])

        def cst():
            L.pir_tune_set(26, 1)
            call(out[1])

        auto(); cst()
        torch.cuda.synchronize()
        equal = torch.equal(out[0], out[1])
        g = _lib.GemmNN()
        a3, kp = ops._split_weight(w, dgrad=dgrad)
        g.A, g.A3, g.a3_kp, g.X, g.Y = w.data_ptr(), a3.data_ptr(), kp, x.data_ptr(), out[1].data_ptr()
        g.M, g.K, g.N, g.O1, g.O2, g.ldx, g.ldy = M, K, S * S, B, 1, S * S, S * S
        L.pir_tune_set(26, 1)
        plan = L.pir_gemm_nn_plan(g)
        t_a, t_c = timeit([auto, cst])
        by = 4.0 * S * S * B * (K + M * (2 if res else 1))
        fl = 2.0 * M * K * S * S * B
        print(f"{tag:28s} {M:5d} {K:5d} {S*S:6d} | {t_a*1e6:9.1f} {t_c*1e6:9.1f} {t_c/t_a:6.2f} | {by/t_c/1e9:8.0f} {fl/t_c/1e12:8.1f} | {plan} {equal}",
              flush=True)
    L.pir_tune_set(26, -1)


if __name__ == "__main__":
    main()
