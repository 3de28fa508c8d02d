#!/usr/bin/env python3
"""Tile-configuration sweep (knob 0) on the folded MDTA GEMMs (per-image effective weights, not pre-split):
x1 = W_eff v + x and dv = W_eff^T dx1, M = K = C, N = HW."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "16"))
T = _lib.lib.pir_tune_set
NAMES = {-1: "auto", 0: "32x256", 1: "64x256", 2: "96x256", 3: "128x128", 4: "64x128", 7: "96x128"}
for C, S in ((48, 128), (96, 128), (96, 64)):
    hw = S * S
    weff, qkv, x, dx1 = r(B, C, C), r(B, 3 * C, S, S), r(B, C, S, S), r(B, C, S, S)
    x1, dqkv = torch.empty_like(x), torch.empty_like(qkv)
    bs = 3 * C * hw
    fwd = lambda: ops.gemm_nn(weff, (C * C, 0), C, 1, qkv, 2 * C * hw, (bs, 0), hw, x1, 0, (C * hw, 0), hw, C, C, hw, B, 1,
                              R=x, r_batch=(C * hw, 0), ldr=hw)
    bwd = lambda: ops.gemm_nn(weff, (C * C, 0), 1, C, dx1, 0, (C * hw, 0), hw, dqkv, 2 * C * hw, (bs, 0), hw, C, C, hw, B, 1)
    for tag, fn in (("x1 = W_eff v + x", fwd), ("dv = W_eff^T dx1", bwd)):
        row = []
        for cfg in (-1, 0, 1, 2, 3, 4, 7):
            T(0, cfg)
            row.append(f"{NAMES[cfg]} {timeit(fn)*1e6:6.1f}")
        T(0, -1)
        print(f"C={C:3d} HW={hw:5d} B={B} {tag:18s}: " + " | ".join(row), flush=True)
