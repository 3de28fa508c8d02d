#!/usr/bin/env python3
"""Dense 3x3 convolutions of the PromptIR path (patch embed, down / upsample bodies, prompt convolutions, output) at the
shapes of a part batch: forward, input gradient, weight gradient - time, and each against its own roofline bound.

    python tools/conv3x3_bench.py [--batch 16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.resident_ab import r, timeit  # noqa: E402

MFMA, HBM = 2500e12 / 6, 6.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    B = ap.parse_args().batch
    shapes = [("patch_embed", 3, 48, 128), ("down1_2", 48, 24, 128), ("down2_3", 96, 48, 64), ("down3_4", 192, 96, 32),
              ("up4_3", 384, 768, 16), ("up3_2", 192, 384, 32), ("up2_1", 96, 192, 64), ("prompt3 conv", 320, 320, 16),
              ("prompt2 conv", 128, 128, 32), ("prompt1 conv", 64, 64, 64), ("output", 96, 3, 128)]
    print(f"batch {B}\n{'conv':14s} {'cin':>4s} {'cout':>4s} {'HW':>6s} | {'fwd us':>8s} {'eff':>5s} | {'dgrad us':>8s} {'eff':>5s} | {'wgrad us':>8s} {'eff':>5s}")
    tot = [0.0, 0.0, 0.0]
    for name, cin, cout, S in shapes:
        x, w, dy = r(B, cin, S, S), r(cout, cin, 3, 3), r(B, cout, S, S)
        fl = 2.0 * 9 * cin * cout * S * S * B
        by = 4.0 * B * S * S * (cin + cout)
        bound = max(fl / MFMA, by / HBM)
        t = timeit([lambda: ops.conv3x3_forward(x, w), lambda: ops.conv3x3_dgrad(dy, w), lambda: ops.conv3x3_wgrad(dy, x, w)])
        for i in range(3):
            tot[i] += t[i]
        print(f"{name:14s} {cin:4d} {cout:4d} {S*S:6d} | {t[0]*1e6:8.1f} {bound/t[0]:5.2f} | {t[1]*1e6:8.1f} {bound/t[1]:5.2f} | {t[2]*1e6:8.1f} {bound/t[2]:5.2f}", flush=True)
    print(f"sum: fwd {tot[0]*1e3:.2f} ms, dgrad {tot[1]*1e3:.2f} ms, wgrad {tot[2]*1e3:.2f} ms")


if __name__ == "__main__":
    main()
