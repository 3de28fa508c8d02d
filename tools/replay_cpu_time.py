#!/usr/bin/env python3
"""Host time of one train_step call (graph launch + optimiser launches, no synchronisation) against the device time of
the step: is the step bound by the host's launch rate?   B=8 python tools/replay_cpu_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from promptir_amd.train import DataParallelTrainer  # noqa: E402

B = int(os.environ.get("B", "8"))
dev = torch.device("cuda:0")
net, _sd = bench.build_model(dev)
trainer = DataParallelTrainer(net, lr=2e-4)
x, t = bench.build_batch(B, 128, 0, dev)
for _ in range(3):
    trainer.train_step(x, t)
torch.cuda.synchronize()
host, wall = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.train_step(x, t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
# back-to-back: the host runs ahead
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    trainer.train_step(x, t)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("batch %d: host per call %.2f ms (median), device-complete after %.2f ms; 10 back to back: host %.2f ms per step, wall %.2f ms per step"
      % (B, sorted(host)[5], sorted(wall)[5], (t1 - t0) * 100, (t2 - t0) * 100))
