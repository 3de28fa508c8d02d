#!/usr/bin/env python3
"""LayerNorm + 1x1 convolution forward: the LayerNorm-on-load B-stationary kernel (pir_ln_conv1x1_fwd) against the pair
pir_layernorm_fwd + pir_gemm_nn on the shapes of the train step.

    python tools/ln_fwd_ab.py [--batch 32]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.resident_ab import r, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    B = args.batch
    print(f"batch {B}\n{'shape':28s} | {'pair us':>9s} {'fused us':>9s} {'ratio':>6s}")
    for c, S in ((48, 128), (96, 128), (96, 64)):
        hid = int(c * 2.66)
        for tag, M in (("qkv (LN1)", 3 * c), ("ffn_in (LN2)", 2 * hid)):
            if args.only and args.only not in f"C{c} {S}^2 {tag}":
                continue
            x, w, gam, bet = r(B, c, S, S), r(M, c, 1, 1), r(c), r(c)

            def pair():
                xn, _, _ = ops.layernorm_forward(x, gam, bet)
                ops.conv1x1_forward(xn, w)

            if ops.ln_conv1x1_forward(x, gam, bet, w) is None:
                print(f"C{c} {S}^2 {tag:14s} M={M:4d} | not served at this batch", flush=True)
                continue

            def fused():
                ops.ln_conv1x1_forward(x, gam, bet, w)

            t_p, t_f = timeit([pair, fused])
            print(f"C{c} {S}^2 {tag:14s} M={M:4d} | {t_p*1e6:9.1f} {t_f*1e6:9.1f} {t_f/t_p:6.2f}", flush=True)


if __name__ == "__main__":
    main()
