#!/usr/bin/env python3
"""Splits the train step of bench.py into its graph replay (forward + loss + backward) and the rest (gradient merge of
the part-batch buffers, AdamW, weight re-split): HIP-event times over 10 steps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from promptir_amd.train import DataParallelTrainer  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
net, sd = bench.build_model(dev)
tr = DataParallelTrainer(net, lr=2e-4)
x, t = bench.build_batch(32, 128, 0, dev)
tr.prepare(x, t)
for _ in range(3):
    tr.train_step(x, t)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
acc = [0.0, 0.0, 0.0]
N = 10
for _ in range(N):
    ev[0].record()
    tr.forward_backward(x, t)
    ev[1].record()
    tr.opt.step(lr=2e-4, grad_scale=1.0)
    ev[2].record()
    from promptir_amd import ops
    ops.refresh_split_weights()
    tr._split_sig = ops.split_weights_signature()
    ev[3].record()
    torch.cuda.synchronize()
    for i in range(3):
        acc[i] += ev[i].elapsed_time(ev[i + 1])
print("ms per step: forward+backward (graph replay + part-gradient merge) %.3f | AdamW %.3f | weight re-split %.3f" % tuple(a / N for a in acc))
