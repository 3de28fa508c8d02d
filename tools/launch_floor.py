#!/usr/bin/env python3
"""Floor cost of a dependent kernel launch inside a replayed hipGraph: chains of N tiny pir_add launches on one and on two
forked streams (the step's execution mode), time per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "2000"))
a = [torch.zeros(256, device=dev) for _ in range(2)]
b = torch.ones(256, device=dev)


def chain(t, n):
    for _ in range(n):
        ops.add_(t, b)


def build(streams):
    g = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream() for _ in range(streams)]
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        if streams == 1:
            chain(a[0], N)
        else:
            ev = torch.cuda.Event(); ev.record(main)
            for i, st in enumerate(side):
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    chain(a[i], N // streams)
            for st in side:
                e2 = torch.cuda.Event(); e2.record(st); main.wait_event(e2)
    return g


for streams in (1, 2):
    g = build(streams)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print("%d stream(s): %d launches in %.3f ms -> %.2f us per launch of the graph, %.2f us per launch of a stream"
          % (streams, N, ms, ms * 1e3 / N, ms * 1e3 / (N // streams)))

# two SEPARATE single-chain graphs replayed on two streams at the same time
gs, sts = [], [torch.cuda.Stream() for _ in range(2)]
for i in range(2):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(a[i], N // 2)
    gs.append(g)
torch.cuda.synchronize()


def both():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    for g, st in zip(gs, sts):
        st.wait_event(ev)
        with torch.cuda.stream(st):
            g.replay()
        e2 = torch.cuda.Event(); e2.record(st); main.wait_event(e2)


both(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    both()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
print("two graphs on two streams: %d launches in %.3f ms -> %.2f us per launch of a stream" % (N, ms, ms * 1e3 / (N // 2)))
