#!/usr/bin/env python3
"""The four 1x1 weight gradients of one low-resolution block as ONE grouped launch, repeated: target for counter passes.
    python tools/one_group.py C SIDE [--batch 16] [--iters 6] [--knob K=V ...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("c", type=int); ap.add_argument("side", type=int)
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--knob", action="append", default=[])
a = ap.parse_args()
for kv in a.knob:
    k, v = kv.split("=")
    _lib.lib.pir_tune_set(int(k), int(v))
dev, C, S, B = "cuda:0", a.c, a.side, a.batch
hid = int(C * 2.66)
r = lambda *s: torch.randn(*s, device=dev)
items = []
for cin, cout in ((C, 3 * C), (C, C), (C, 2 * hid), (hid, C)):
    items.append((r(B, cout, S, S), r(B, cin, S, S), torch.empty(cout, cin, 1, 1, device=dev)))
import time
for _ in range(a.iters):
    with ops.deferred_reductions():
        ops.conv1x1_wgrad_group(items)
    ops.flush_reductions()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    with ops.deferred_reductions():
        ops.conv1x1_wgrad_group(items)
    ops.flush_reductions()
torch.cuda.synchronize()
print("C%d %d^2 batch %d knobs %s: %.1f us per group (+ reductions)" % (C, S, B, a.knob, (time.perf_counter() - t0) / 20 * 1e6))
