#!/usr/bin/env python3
"""gemm_nn tile-configuration sweep (knob 0) on the 1x1-convolution shapes of every level: auto vs every fixed config."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
T = _lib.lib.pir_tune_set
NAMES = {-1: "auto", 0: "32x256", 1: "64x256", 2: "96x256", 3: "128x128", 4: "64x128", 7: "96x128"}
LEVELS = [("L1 enc C48 128^2", 48, 128), ("L1 dec C96 128^2", 96, 128), ("L2 C96 64^2", 96, 64), ("L3 C192 32^2", 192, 32),
          ("L4 C384 16^2", 384, 16), ("noise3 C704 16^2", 704, 16)]
if os.environ.get("LEVELS"):
    LEVELS = [LEVELS[int(i)] for i in os.environ["LEVELS"].split(",")]
tot = {}
for name, C, S in LEVELS:
    hid = int(C * 2.66)
    for tag, cin, cout, res in (("qkv", C, 3 * C, False), ("proj", C, C, True), ("ffn_in", C, 2 * hid, False),
                                ("ffn_out", hid, C, True)):
        x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
        res_t = r(B, cout, S, S) if res else None
        out = torch.empty(B, cout, S, S, device="cuda:0")
        dy, dx = r(B, cout, S, S), torch.empty(B, cin, S, S, device="cuda:0")
        for mode, fn, M, K in (("fwd", lambda: ops.conv1x1_forward(x, w, res_t, out=out), cout, cin),
                               ("dgrd", lambda: ops.conv1x1_dgrad(dy, w, out=dx), cin, cout)):
            row, best = [], None
            for cfg in (-1, 0, 1, 2, 3, 4, 7):
                T(0, cfg)
                t = timeit(fn)
                row.append(f"{NAMES[cfg]} {t*1e6:6.1f}")
                if cfg >= 0 and (best is None or t < best[0]):
                    best = (t, cfg)
                if cfg == -1:
                    tot["auto"] = tot.get("auto", 0.0) + t
            tot["best"] = tot.get("best", 0.0) + best[0]
            T(0, -1)
            print(f"{name:18s} {mode:4s} {tag:8s} M={M:4d} K={K:4d}: " + " | ".join(row) + f" | best {NAMES[best[1]]}", flush=True)
print({k: round(v * 1e3, 3) for k, v in tot.items()})
