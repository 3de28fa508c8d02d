"""Graph-replayed inference for fixed input shapes (serving / tiled restoration).

A PromptIR forward at small batch is launch-bound from Python (~600 C-ABI launches); capturing it once into a
hipGraph (`torch.cuda.CUDAGraph`) removes the host from the loop.  No entry point of the C ABI allocates or
synchronises, so the whole forward is capturable; the bf16x3 weight pieces must be current before the capture
(`ops.refresh_split_weights()` after any weight update).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


class GraphedForward:
    """`y = GraphedForward(net)(x)`: one captured graph per input shape; outputs are fresh copies."""

    def __init__(self, net: torch.nn.Module, max_graphs: int = 4, part_streams: int = None):
        import os

        self.net = net
        self.max_graphs = max_graphs
        self._graphs: Dict[Tuple[int, ...], tuple] = {}
        # as in the trainer: from eight images on the batch runs as two halves on two streams inside the graph (the small
        # kernels of one half fill the gaps of the other)
        self.part_streams = int(os.environ.get("PIR_INFER_STREAMS", "2")) if part_streams is None else part_streams
        self.min_part = max(1, int(os.environ.get("PIR_INFER_MIN_PART", "4")))    # images per part stream at least
        self._streams = None

    def _forward_parts(self, sx: torch.Tensor, sy: torch.Tensor = None):
        from . import ops

        b = sx.shape[0]
        n = min(self.part_streams, max(1, b // self.min_part))
        if n <= 1:
            return self.net(sx)
        if self._streams is None or len(self._streams) < n:
            self._streams = [torch.cuda.Stream(sx.device) for _ in range(n)]
        main = torch.cuda.current_stream(sx.device)
        ready = torch.cuda.Event()
        ready.record(main)
        if sy is None:
            sy = torch.empty_like(sx)
        for i in range(n):
            lo, hi = b * i // n, b * (i + 1) // n
            st = self._streams[i]
            st.wait_event(ready)
            with torch.cuda.stream(st):
                ops.copy_planes(self.net(sx[lo:hi]), sy[lo:hi])
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
        return sy

    @torch.no_grad()
    def _capture(self, x: torch.Tensor):
        sx = x.clone()
        side = torch.cuda.Stream(x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                      # warm-up: weight pieces, workspaces, allocator pools
                self._forward_parts(sx)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            sy = self._forward_parts(sx)
        return graph, sx, sy

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("GraphedForward needs a ROCm tensor (no CPU fallback)")
        key = tuple(x.shape)
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = self._capture(x)
        from . import ops

        graph, sx, sy = ent
        ops.copy_flat(x if x.is_contiguous() else ops._gather_contiguous(x), sx)   # library kernels only: no runtime blits
        graph.replay()
        return ops.copy_flat(sy, torch.empty_like(sy))
