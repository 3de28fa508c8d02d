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

    def __init__(self, net: torch.nn.Module, max_graphs: int = 4):
        self.net = net
        self.max_graphs = max_graphs
        self._graphs: Dict[Tuple[int, ...], tuple] = {}

    @torch.no_grad()
    def _capture(self, x: torch.Tensor):
        sx = x.clone()
        side = torch.cuda.Stream(x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                      # warm-up: weight pieces, workspaces, allocator pools
                self.net(sx)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            sy = self.net(sx)
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
