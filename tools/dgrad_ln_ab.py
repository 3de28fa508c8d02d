#!/usr/bin/env python3
"""Input gradient + LayerNorm backward: the fused C-stationary kernel (pir_conv1x1_dgrad_ln_bwd) against the pair
pir_gemm_nn + pir_layernorm_bwd on the 96-channel shapes of the train step.

    python tools/dgrad_ln_ab.py [--batch 32]
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
    B = ap.parse_args().batch
    print(f"batch {B}\n{'shape':24s} | {'pair us':>9s} {'fused us':>9s} {'ratio':>6s} | fused GB/s (alg.)")
    for c, S in ((48, 128), (96, 128), (96, 64), (192, 32)):
        for tag, K in (("qkv (LN1)", 3 * c), ("ffn_in (LN2)", 2 * int(c * 2.66))):
            x, w, dy, dres = r(B, c, S, S), r(K, c, 1, 1), r(B, K, S, S), r(B, c, S, S)
            gam, bet = r(c), r(c)
            _, mean, rstd = ops.layernorm_forward(x, gam, bet)
            sw, sb = torch.empty_like(gam), torch.empty_like(gam)

            def pair():
                dxn = ops.conv1x1_dgrad(dy, w)
                ops.layernorm_backward(dxn, x, gam, True, mean, rstd, sw, sb, dres=dres)

            def fused():
                assert ops.conv1x1_dgrad_ln_backward(dy, w, x, gam, mean, rstd, sw, sb, dres=dres) is not None

            t_p, t_f = timeit([pair, fused])
            by = 4.0 * S * S * B * (K + 3 * c)
            print(f"C{c} {S}^2 {tag:14s} | {t_p*1e6:9.1f} {t_f*1e6:9.1f} {t_f/t_p:6.2f} | {by/t_f/1e9:8.0f}", flush=True)


if __name__ == "__main__":
    main()
