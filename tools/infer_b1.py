#!/usr/bin/env python3
"""Batch-B graph-replayed forward, repeated: target for a kernel trace of the batch-independent chain.  B=1 python tools/infer_b1.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from promptir_amd.infer import GraphedForward  # noqa: E402

dev = torch.device("cuda", 0)
net, _ = bench.build_model(dev)
net.eval()
B = int(os.environ.get("B", "1"))
x, _ = bench.build_batch(B, 128, 0, dev)
g = GraphedForward(net)
with torch.no_grad():
    for _ in range(int(os.environ.get("ITERS", "20"))):
        g(x)
torch.cuda.synchronize()
