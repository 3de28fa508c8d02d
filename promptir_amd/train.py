"""Training engine for the PromptIR path: plain torch.distributed data parallelism.

Replaces the reference's Lightning wrapper (train.py:28-56: PromptIRModel with nn.L1Loss,
AdamW(lr=2e-4), LinearWarmupCosineAnnealingLR(15, 150) stepped per epoch) and its
`pl.Trainer(strategy="ddp_find_unused_parameters_true")` (train.py:339).

MI355X-first layout: all live parameters are views into ONE flat fp32 buffer, with matching
flat gradient / Adam-moment buffers.  The HIP weight-gradient kernels write straight into
the flat gradient buffer (gradient sinks, promptir_amd/ops.py), so a step is
    forward -> L1 -> backward -> ONE all-reduce of 141.5 MB over RCCL/xGMI -> ONE AdamW kernel.
The six parameters the reference never uses in forward stay outside the flat
buffers: they get no gradient and no update — the same end state as DDP's
find_unused_parameters=True with torch's AdamW skipping grad-less parameters.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

UNUSED_PREFIXES = ("chnl_reduce1.", "chnl_reduce2.", "chnl_reduce3.",
                   "reduce_noise_channel_1.", "reduce_noise_channel_2.", "reduce_noise_channel_3.")


def warmup_cosine_lr(epoch: int, base_lr: float = 2e-4, warmup_epochs: int = 15, max_epochs: int = 150,
                     warmup_start_lr: float = 0.0, eta_min: float = 0.0) -> float:
    """Closed form of the reference scheduler when `step(epoch)` is called (utils/schedulers.py:332-346,
    driven from train.py:48-50).  Note lr == 0 during epoch 0."""
    if epoch < warmup_epochs:
        return warmup_start_lr + epoch * (base_lr - warmup_start_lr) / (warmup_epochs - 1)
    return eta_min + 0.5 * (base_lr - eta_min) * (
        1 + math.cos(math.pi * (epoch - warmup_epochs) / (max_epochs - warmup_epochs)))


def live_parameters(net: nn.Module, unused_prefixes: Iterable[str] = UNUSED_PREFIXES):
    pref = tuple(unused_prefixes)
    return [(n, p) for n, p in net.named_parameters() if not n.startswith(pref)]


class FlatAdamW:
    """Flat parameter / gradient / moment buffers + the fused AdamW step (torch.optim.AdamW defaults)."""

    ALIGN = 64  # floats; keeps every parameter 256-byte aligned inside the flat buffers

    def __init__(self, net: nn.Module, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, unused_prefixes: Iterable[str] = UNUSED_PREFIXES):
        self.named = live_parameters(net, unused_prefixes)
        if not self.named:
            raise ValueError("no parameters")
        dev = self.named[0][1].device
        self.offsets: Dict[str, int] = {}
        off = 0
        for n, p in self.named:
            self.offsets[n] = off
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.live_numel = sum(p.numel() for _, p in self.named)
        self.param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        for n, p in self.named:
            o = self.offsets[n]
            view = self.param[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            gview = self.grad[o:o + p.numel()].view_as(p)
            p._grad_sink = gview     # HIP wgrad kernels write here (promptir_amd/ops.py)
            p.grad = gview           # optimiser-style access for callers / checkpoints
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.steps = 0

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0) -> None:
        self.steps += 1
        if torch.device(self.param.device).type == "cuda":
            from . import ops

            ops.adamw_step(self.param, self.grad, self.exp_avg, self.exp_avg_sq, self.steps,
                           lr=self.lr if lr is None else lr, betas=self.betas, eps=self.eps,
                           weight_decay=self.weight_decay, grad_scale=grad_scale)
            ops.weights_changed()   # cached bf16x3 weight pieces are stale now
        else:
            raise RuntimeError("FlatAdamW.step: the AdamW kernel is HIP-only (no CPU fallback)")

    def state_dict(self):
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "offsets": dict(self.offsets)}

    def load_state_dict(self, sd):
        self.steps = int(sd["steps"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


def allreduce_mean_(flat_grad: torch.Tensor, world_size: int) -> float:
    """One SUM all-reduce over the whole flat gradient (RCCL on GPUs, gloo on CPU tests).
    Returns the scale (1/world) the optimiser kernel folds into its gradient read."""
    if world_size > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / world_size


class DataParallelTrainer:
    """One process per GPU; the batch is sharded, the model is replicated."""

    def __init__(self, net: nn.Module, lr: float = 2e-4, loss_fn=None, micro_streams: Optional[int] = None,
                 graph: Optional[bool] = None):
        self.net = net
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.opt = FlatAdamW(net, lr=lr)
        if self.world > 1:  # DDP's initial parameter broadcast from rank 0
            dist.broadcast(self.opt.param, src=0)
        if loss_fn is None:
            from .ops import l1_loss as loss_fn
        self.loss_fn = loss_fn
        # The batch is cut into parts that run on their own HIP streams: the kernels of one part (say a
        # bandwidth-bound stencil) run beside those of another (an MFMA-bound GEMM) and fill each other's idle
        # units.  Each part writes its weight gradients into its own flat buffer (the sinks are captured at
        # forward time); the parts are summed once.  Measured at batch 32 inside the hipGraph: 1 stream 131.9 ms,
        # 2: 125.0, 3: 122.9, 4: 122.3 (eager, two streams: 142.9 ms - the doubled launch count makes the CPU
        # the bottleneck, which is why this lives inside the graph).
        if graph is None:
            graph = os.environ.get("PIR_GRAPH", "1") != "0"
        self.graph, self._graph, self._graph_shape = bool(graph) and self.opt.param.is_cuda, None, None
        if micro_streams is None:   # part streams only pay inside the graph (eagerly the extra launches bind the CPU)
            micro_streams = int(os.environ.get("PIR_MICRO_STREAMS", "4" if self.graph else "1"))
        self.micro_streams = max(1, micro_streams) if self.opt.param.is_cuda else 1
        if self.micro_streams > 1:
            from . import ops

            ops.USE_SIDE_STREAM = False   # the parts already overlap; side streams inside them do not mix with capture
            dev = self.opt.param.device
            n = self.micro_streams
            self._streams = [torch.cuda.Stream(dev) for _ in range(n)]
            self._grads = [self.opt.grad] + [torch.zeros_like(self.opt.grad) for _ in range(n - 1)]
            self._sinks = [[] for _ in range(n)]
            for name, p in self.opt.named:
                o = self.opt.offsets[name]
                for i in range(n):
                    self._sinks[i].append((p, self._grads[i][o:o + p.numel()].view_as(p)))

    def _use_sinks_default(self) -> None:
        for n, p in self.opt.named:
            o = self.opt.offsets[n]
            p._grad_sink = self.opt.grad[o:o + p.numel()].view_as(p)

    def _use_sinks(self, which: int) -> None:
        for p, view in self._sinks[which]:
            p._grad_sink = view

    def _fwd_bwd(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor, two_streams: bool = True) -> torch.Tensor:
        """forward + L1 + backward; leaves the batch-mean gradient in opt.grad and returns the loss (device scalar)."""
        b = degrad_patch.shape[0]
        n = min(self.micro_streams, max(1, b // 4))      # parts of at least four samples
        if not (two_streams and n > 1):
            loss = self.loss_fn(self.net(degrad_patch), clean_patch)
            loss.backward()
            return loss.detach()
        from . import ops

        bounds = [b * i // n for i in range(n + 1)]
        main = torch.cuda.current_stream(degrad_patch.device)
        ready = torch.cuda.Event()
        ready.record(main)
        losses = []
        for i in range(n):
            lo, hi = bounds[i], bounds[i + 1]
            st = self._streams[i]
            st.wait_event(ready)
            self._use_sinks(i)                       # captured by the autograd nodes of this forward
            with torch.cuda.stream(st):
                w = torch.full((), (hi - lo) / b, dtype=torch.float32, device=degrad_patch.device)
                losses.append((self.loss_fn(self.net(degrad_patch[lo:hi]), clean_patch[lo:hi]), w))
        for i, (loss, w) in enumerate(losses):
            with torch.cuda.stream(self._streams[i]):
                loss.backward(gradient=w)
        for st in self._streams[:n]:
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
        self._use_sinks(0)
        total = None
        for i, (loss, w) in enumerate(losses):      # mean over the batch = sum of the weighted parts
            if i:
                ops.add_(self.opt.grad, self._grads[i])
            total = loss.detach() * w if total is None else total + loss.detach() * w
        return total

    def _capture(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor) -> None:
        """hipGraph of forward + loss + backward on static input buffers.  The optimiser, the
        all-reduce and the weight re-split stay outside (their arguments change from step to step)."""
        self._sx, self._st = degrad_patch.clone(), clean_patch.clone()
        side = torch.cuda.Stream(degrad_patch.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                # warm-up on a side stream: caches, workspaces, allocator pools
            for _ in range(2):
                self._fwd_bwd(self._sx, self._st)
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread of a multi-GPU job may query its events while this thread captures
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
            self._sloss = self._fwd_bwd(self._sx, self._st)
        self._graph_shape = tuple(degrad_patch.shape)

    def _ensure_graph(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor) -> None:
        if self._graph is not None and self._graph_shape == tuple(degrad_patch.shape):
            return
        try:
            self._capture(degrad_patch, clean_patch)
        except Exception as exc:   # e.g. a runtime that refuses the capture: keep training, eagerly on one stream
            import warnings

            warnings.warn(f"hipGraph capture failed ({exc!r}); falling back to the eager single-stream step")
            torch.cuda.synchronize()
            self.graph, self._graph, self.micro_streams = False, None, 1
            self._use_sinks_default()

    def prepare(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor) -> None:
        """Optional: build the graph for this batch shape now (otherwise the first train_step does it)."""
        if self.graph and degrad_patch.is_cuda:
            self._ensure_graph(degrad_patch, clean_patch)

    def train_step(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor, lr: Optional[float] = None):
        """reference train.py:37-46 (+ optimizer.step of Lightning's loop)."""
        from . import ops

        if self.graph and degrad_patch.is_cuda and ops.lib.records is None:
            self._ensure_graph(degrad_patch, clean_patch)
        if self.graph and degrad_patch.is_cuda and ops.lib.records is None:
            self._sx.copy_(degrad_patch)
            self._st.copy_(clean_patch)
            self._graph.replay()
            loss = self._sloss.clone()
        else:   # eager; the instrumented (per-kernel timed) step stays on one stream
            loss = self._fwd_bwd(degrad_patch, clean_patch, two_streams=ops.lib.records is None)
        scale = allreduce_mean_(self.opt.grad, self.world)
        self.opt.step(lr=lr, grad_scale=scale)
        if self.graph:
            ops.refresh_split_weights()
        return loss

    def checkpoint(self, epoch: int) -> dict:
        """Lightning-compatible dict: state_dict keys are `net.<PromptIR key>`."""
        return {"epoch": epoch, "global_step": self.opt.steps,
                "state_dict": {"net." + k: v.detach().clone() for k, v in self.net.state_dict().items()},
                "optimizer_states": [self.opt.state_dict()]}


def load_lightning_checkpoint(net: nn.Module, ckpt: dict) -> None:
    sd = ckpt.get("state_dict", ckpt)
    net.load_state_dict({k[4:] if k.startswith("net.") else k: v for k, v in sd.items()})


def init_distributed() -> tuple:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("PIR_FORCE_PG") == "1") and not dist.is_initialized():   # PIR_FORCE_PG: 1-rank test
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world
