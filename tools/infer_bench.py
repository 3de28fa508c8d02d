#!/usr/bin/env python3
"""Inference throughput, eager vs hipGraph replay (BASELINE config 2: batch 8 x 3x128x128)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from promptir_amd.infer import GraphedForward  # noqa: E402

dev = torch.device("cuda", 0)
net, _ = bench.build_model(dev)
net.eval()
for B in (1, 8, 32):
    x, _ = bench.build_batch(B, 128, 0, dev)
    g = GraphedForward(net)
    with torch.no_grad():
        for fn, tag in ((lambda: net(x), "eager"), (lambda: g(x), "graph")):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print(f"batch {B:2d} {tag}: {dt * 1e3:7.2f} ms  {B / dt:8.1f} patches/s", flush=True)
