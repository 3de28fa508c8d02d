#!/usr/bin/env python3
"""A/B of the resident-panel gemm_nn (gemm_res.hip, knob 20) against the tiled bf16x3 kernel (gemm_x3.hip) on the 1x1
convolution shapes of a batch-32 128x128 train step: bit-equality of the results and time per call.

    python tools/resident_ab.py [--batch 32] [--tm 0] [--wgs 0]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def timeit(fns, rounds=7, inner=4):
    """Interleaved rounds of several variants in one process; median per variant."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    ts = [[] for _ in fns]
    for _ in range(rounds):
        for i, f in enumerate(fns):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(inner):
                f()
            e.record()
            torch.cuda.synchronize()
            ts[i].append(s.elapsed_time(e) / inner * 1e-3)
    return [sorted(t)[len(t) // 2] for t in ts]


def r(*shape):
    return torch.randn(*shape, device=DEV)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tm", type=int, default=0)
    ap.add_argument("--wgs", type=int, default=0)
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    B = args.batch
    L = _lib.lib
    L.pir_tune_set(21, args.tm)
    L.pir_tune_set(23, args.wgs)
    shapes = []   # (tag, cout, cin, side, residual, dgrad)
    for C, S in ((48, 128), (96, 128), (96, 64), (192, 64), (192, 32), (384, 32)):
        hid = int(C * 2.66)
        shapes += [(f"C{C} {S}^2 qkv fwd", 3 * C, C, S, False, False), (f"C{C} {S}^2 ffn_in fwd", 2 * hid, C, S, False, False),
                   (f"C{C} {S}^2 ffn_out fwd+R", C, hid, S, True, False), (f"C{C} {S}^2 ffn_out dgrad", C, hid, S, False, True),
                   (f"C{C} {S}^2 qkv dgrad", 3 * C, C, S, False, True), (f"C{C} {S}^2 ffn_in dgrad", 2 * hid, C, S, False, True),
                   (f"C{C} {S}^2 proj fwd+R", C, C, S, True, False)]
    print(f"{'shape':28s} {'M':>5s} {'K':>5s} {'N':>6s} | {'tiled us':>9s} {'res us':>9s} {'bst us':>9s} {'best/t':>6s} | {'GB/s':>8s} {'TF/s':>8s} | plan  equal(res, bst)")
    for tag, cout, cin, S, res, dgrad in shapes:
        if args.only and args.only not in tag:
            continue
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

        def tiled():
            L.pir_tune_set(20, 0); L.pir_tune_set(24, 0)
            call(out[0])

        def resident():
            L.pir_tune_set(20, 1); L.pir_tune_set(24, 0)
            call(out[1])

        def bstat():
            L.pir_tune_set(20, 1); L.pir_tune_set(24, 1)
            call(out[2])

        out.append(torch.empty_like(out[0]))
        L.pir_tune_set(20, 1)
        g = _lib.GemmNN()
        tiled(); resident(); bstat()
        torch.cuda.synchronize()
        equal = (torch.equal(out[0], out[1]), torch.equal(out[0], out[2]))
        served = None
        # was the resident kernel actually taken?  (plan 9000)
        a3, kp = ops._split_weight(w, dgrad=dgrad)
        g.A, g.A3, g.a3_kp, g.X, g.Y = w.data_ptr(), a3.data_ptr(), kp, x.data_ptr(), out[1].data_ptr()
        g.M, g.K, g.N, g.O1, g.O2, g.ldx, g.ldy = M, K, S * S, B, 1, S * S, S * S
        L.pir_tune_set(20, 1)
        served = L.pir_gemm_nn_plan(g)
        t_t, t_r, t_b = timeit([tiled, resident, bstat])
        by = 4.0 * S * S * B * (K + M * (2 if res else 1))
        fl = 2.0 * M * K * S * S * B
        best = min(t_r, t_b)
        print(f"{tag:28s} {M:5d} {K:5d} {S*S:6d} | {t_t*1e6:9.1f} {t_r*1e6:9.1f} {t_b*1e6:9.1f} {best/t_t:6.2f} | {by/best/1e9:8.0f} {fl/best/1e12:8.1f} | {served}  {equal}",
              flush=True)
    L.pir_tune_set(20, -1); L.pir_tune_set(24, -1)


if __name__ == "__main__":
    main()
