#!/usr/bin/env python3
"""Fused GDFN forward (pir_gdfn_fused_fwd) against the pair it replaces (pir_ln_conv1x1_fwd + pir_dwconv3x3_gate) at the
shapes of the network's 128^2 and 64^2 levels; HIP-event medians, microseconds."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402

ops.GDFN_FUSED = True

dev = torch.device("cuda", 0)


def med(fn, reps=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


print(f"{'B':>3} {'C':>3} {'HxW':>9} | {'pair us':>9} {'fused us':>9} {'ratio':>6} | fused GB/s (x + g)")
for b in (8, 16, 25):
    for c, h, w in ((96, 128, 128), (48, 128, 128), (96, 64, 64)):
        hid = int(c * 2.66)
        x = torch.randn(b, c, h, w, device=dev)
        lw, lb = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        win = torch.randn(2 * hid, c, 1, 1, device=dev) * 0.1
        wdw = torch.randn(2 * hid, 1, 3, 3, device=dev) * 0.1

        def pair():
            return ops.dwconv_gate_forward(ops.ln_conv1x1_forward(x, lw, lb, win), wdw)

        def fused():
            return ops.gdfn_fused_forward(x, lw, lb, win, wdw)

        assert fused() is not None
        tp, tf = med(pair), med(fused)
        print(f"{b:3d} {c:3d} {h:4d}x{w:<4d} | {tp:9.1f} {tf:9.1f} {tf / tp:6.2f} | {4.0 * b * h * w * (c + hid) / tf / 1e3:8.0f}")
