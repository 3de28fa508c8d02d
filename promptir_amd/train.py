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


def lightning_epoch_lr(epoch: int, **kw) -> float:
    """Learning rate the reference actually TRAINS epoch `epoch` with under Lightning 2.0.1.

    `lr_scheduler_step` (train.py:48-50) calls `scheduler.step(self.current_epoch)`.  For an epoch-interval,
    non-plateau scheduler Lightning 2.0.1 makes that call from `_TrainingEpochLoop.advance`, right after the LAST batch
    of the epoch (`if self._num_ready_batches_reached(): self.update_lr_schedulers("epoch",
    update_plateau_schedulers=False)`), i.e. BEFORE `FitLoop.on_advance_end` fires `on_train_epoch_end` /
    ModelCheckpoint; the later call in `on_advance_end` has `update_plateau_schedulers=True` and skips a cosine
    scheduler.  `trainer.current_epoch` is `epoch_progress.current.completed`, still N while epoch N runs: the call at
    the end of epoch N passes N, so epoch N+1 trains with closed_form(N), and epoch 0 trains with the scheduler's
    construction-time value closed_form(0) = warmup_start_lr = 0.  Epochs 0 and 1 therefore both run with lr 0.
    (Lightning is not installed in the build container; this ordering is restated from the 2.0.1 sources as recorded in
    ADVICE rounds 1 and 3 and is not executed anywhere here.)"""
    return warmup_cosine_lr(max(epoch - 1, 0), **kw)


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
        # Gradient segments: when the module can say which piece of its forward owns a parameter (PromptIR.stage_of),
        # the flat buffers are laid out piece by piece, so the gradients a backward segment completes form ONE
        # contiguous range that can be all-reduced while the next segment runs.  (Checkpoints address parameters by
        # name, so the order inside the flat buffers is free.)
        stage_of = getattr(net, "stage_of", None)
        self.stages = None
        if callable(stage_of):
            order = sorted(range(len(self.named)), key=lambda i: (stage_of(self.named[i][0]), i))
            self.named = [self.named[i] for i in order]
        dev = self.named[0][1].device
        self.offsets: Dict[str, int] = {}
        off = 0
        for n, p in self.named:
            self.offsets[n] = off
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.live_numel = sum(p.numel() for _, p in self.named)
        if callable(stage_of):   # [begin, end) of every stage's slice of the flat buffers
            self.stages, begin, cur = [], 0, stage_of(self.named[0][0])
            for n, _ in self.named:
                if stage_of(n) != cur:
                    self.stages.append((begin, self.offsets[n]))
                    begin, cur = self.offsets[n], stage_of(n)
            self.stages.append((begin, off))
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
        """Flat layout (this repo's own, compact)."""
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "offsets": dict(self.offsets)}

    def torch_state_dict(self, net: nn.Module, lr: Optional[float] = None) -> dict:
        """The same state in torch.optim.AdamW.state_dict() layout, as Lightning stores it in
        `checkpoint["optimizer_states"][0]` for the reference's `optim.AdamW(self.parameters(), lr=2e-4)`
        (train.py:53): parameter indices count ALL parameters in `net.parameters()` order; the six parameters that
        never receive a gradient have no state entry (torch creates state lazily, on the first step with a grad)."""
        state = {}
        names = [n for n, _ in net.named_parameters()]
        for i, n in enumerate(names):
            if n not in self.offsets or self.steps == 0:
                continue
            o, k = self.offsets[n], dict(self.named)[n].numel()
            shape = dict(self.named)[n].shape
            state[i] = {"step": torch.tensor(float(self.steps)),
                        "exp_avg": self.exp_avg[o:o + k].view(shape).detach().clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + k].view(shape).detach().clone()}
        group = {"lr": self.lr if lr is None else lr, "initial_lr": self.lr, "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd, net: Optional[nn.Module] = None):
        """Accepts the flat layout or a torch.optim.AdamW state_dict (a reference / Lightning checkpoint); the
        latter needs `net` to map parameter indices to names."""
        if "state" in sd and "param_groups" in sd:
            if net is None:
                raise ValueError("a torch-layout optimizer state needs the module to map parameter indices")
            names = [n for n, _ in net.named_parameters()]
            order = []
            for g in sd["param_groups"]:
                order += list(g["params"])
            if len(order) != len(names):
                raise ValueError(f"optimizer state has {len(order)} parameters, the module {len(names)}")
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            steps = set()
            shapes = {n: p for n, p in self.named}
            for pos, idx in enumerate(order):
                st = sd["state"].get(idx)
                n = names[pos]
                if st is None:
                    continue
                if n not in self.offsets:
                    raise ValueError(f"optimizer state for {n}, which this engine never updates")
                o, k = self.offsets[n], shapes[n].numel()
                self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
            if len(steps) > 1:
                raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a plain AdamW run")
            self.steps = steps.pop() if steps else 0
            return
        self.steps = int(sd["steps"])
        theirs = sd.get("offsets")
        if theirs is None or dict(theirs) == self.offsets:
            if sd["exp_avg"].numel() != self.numel:
                raise ValueError(f"flat optimizer state has {sd['exp_avg'].numel()} entries, this engine {self.numel}")
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            return
        # a flat state written with another parameter order (round-1 checkpoints were laid out in named_parameters()
        # order, this engine sorts by backward stage): the sizes agree, the positions do not - remap by NAME
        sizes = {n: p.numel() for n, p in self.named}
        missing = sorted(set(sizes) - set(theirs))
        extra = sorted(set(theirs) - set(sizes))
        if missing or extra:
            raise ValueError(f"flat optimizer state does not cover this module: missing {missing[:3]}, unknown {extra[:3]}")
        total = sd["exp_avg"].numel()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for n, k in sizes.items():
            src = int(theirs[n])
            if src < 0 or src + k > total:
                raise ValueError(f"flat optimizer state: {n} at {src} (+{k}) lies outside its {total} entries")
            o = self.offsets[n]
            self.exp_avg[o:o + k].copy_(sd["exp_avg"][src:src + k])
            self.exp_avg_sq[o:o + k].copy_(sd["exp_avg_sq"][src:src + k])


def allreduce_mean_(flat_grad: torch.Tensor, world_size: int) -> float:
    """One SUM all-reduce over the whole flat gradient (RCCL on GPUs, gloo on CPU tests).
    Returns the scale (1/world) the optimiser kernel folds into its gradient read."""
    if dist.is_initialized():   # also in a 1-rank group (PIR_FORCE_PG=1): the collective path is then exercised on one GPU
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / world_size


class _CaptureRefused(RuntimeError):
    """The HIP runtime would not capture / instantiate the step as a graph (as opposed to the step itself failing)."""


class DataParallelTrainer:
    """One process per GPU; the batch is sharded, the model is replicated."""

    def __init__(self, net: nn.Module, lr: float = 2e-4, loss_fn=None, micro_streams: Optional[int] = None,
                 graph: Optional[bool] = None):
        self.net = net
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.opt = FlatAdamW(net, lr=lr)
        if dist.is_initialized():  # DDP's initial parameter broadcast from rank 0 (a 1-rank group runs it too)
            dist.broadcast(self.opt.param, src=0)
        # the library's L1 takes a part batch's share of the batch BY VALUE (no device scalar, no ATen fill / multiply
        # in the step); a caller-supplied loss_fn(y, t) is weighted through the upstream gradient instead
        self._own_loss = loss_fn is None
        if loss_fn is None:
            from .ops import l1_loss as loss_fn
        self.loss_fn = loss_fn
        # The batch is cut into parts that run on their own HIP streams: the kernels of one part (say a
        # bandwidth-bound stencil) run beside those of another (an MFMA-bound GEMM) and fill each other's idle
        # units.  Each part writes its weight gradients into its own flat buffer (the sinks are captured at
        # forward time); the parts are summed once.  Measured at batch 32 inside the hipGraph, round 2 kernels: 1 stream
        # 106.4 ms, 2: 101.8, 3: 102.1, 4: 103.7, 6: 124.2 (round 1, slower streaming kernels: 131.9 / 125.0 / 122.9 /
        # 122.3 - the better the bandwidth-bound kernels fill the chip on their own, the less a second part helps).
        # Eagerly the doubled launch count makes the CPU the bottleneck, which is why this lives inside the graph.
        if graph is None:
            graph = os.environ.get("PIR_GRAPH", "1") != "0"
        self.graph, self._graph, self._graph_shape = bool(graph) and self.opt.param.is_cuda, None, None
        if micro_streams is None:   # part streams only pay inside the graph (eagerly the extra launches bind the CPU)
            # (PIR_MICRO_STREAMS=4: four parts of eight at batch 32 measured -0.5 ... -0.9 ms per step, round 4, for twice the
            # launches - 6211 instead of 3131 per step - and two more flat gradient buffers; the default stays at two)
            micro_streams = int(os.environ.get("PIR_MICRO_STREAMS", "2" if self.graph else "1"))
        self.micro_streams = max(1, micro_streams) if self.opt.param.is_cuda else 1
        self.min_part = max(0, int(os.environ.get("PIR_MIN_PART", "0")))     # 0: the rule of _nparts
        self._split_sig = None
        self._seg_state = None
        # backward in three segments with the gradient all-reduce of each finished range overlapping the next segment.
        # OPT-IN (PIR_STAGED=1): RCCL kernels beside hipGraph replays have only run under a 1-rank group and a 2-rank
        # gloo rehearsal so far; until an N > 1 RCCL run shows it equal to and faster than the single-graph step
        # (+1.3 ms of graph seams at batch 8 against <= 1.6 ms of exposed all-reduce) the default for every world size
        # is ONE graph followed by ONE all-reduce of the flat gradient.
        self.staged = (os.environ.get("PIR_STAGED", "0") != "0" and self.opt.param.is_cuda and self._staged_ok())
        if self.micro_streams > 1:
            dev = self.opt.param.device
            n = self.micro_streams
            self._streams = [torch.cuda.Stream(dev) for _ in range(n)]
            self._grads = [self.opt.grad] + [torch.zeros_like(self.opt.grad) for _ in range(n - 1)]
            self._sinks = [[] for _ in range(n)]
            for name, p in self.opt.named:
                o = self.opt.offsets[name]
                for i in range(n):
                    self._sinks[i].append((p, self._grads[i][o:o + p.numel()].view_as(p)))

    def _part_loss(self, y: torch.Tensor, t: torch.Tensor, share: float):
        """(loss node, upstream gradient): with the library's own L1 the node already holds share * mean|y - t| and the
        seed is a persistent device 1.0; otherwise the share travels as the upstream gradient."""
        if self._own_loss and y.is_cuda:
            from . import ops

            return self.loss_fn(y, t, share), ops.unit_gradient(y.device)
        return self.loss_fn(y, t), torch.full((), share, dtype=torch.float32, device=y.device)

    def _total_loss(self, parts) -> torch.Tensor:
        """Sum of the parts' shares of the batch loss (mean over the batch), on the device."""
        total = None
        for loss, w in parts:
            if self._own_loss and loss.is_cuda:
                from . import ops

                total = loss.detach() if total is None else ops.add_(total, loss.detach())
            else:
                total = loss.detach() * w if total is None else total + loss.detach() * w
        return total

    def set_staged(self, staged: bool) -> bool:
        """Switch between ONE graph + ONE all-reduce of the flat gradient (False) and the backward in three segment
        graphs with the all-reduce of each finished gradient range overlapping the next segment (True).  Drops the
        captured graph: the next step re-captures.  Returns the mode now in effect (the staged mode needs the module's
        three-piece forward)."""
        staged = bool(staged) and self.opt.param.is_cuda and self._staged_ok()
        if staged != self.staged:
            self.staged, self._graph, self._graph_shape, self._seg_state = staged, None, None, None
        return self.staged

    def _use_sinks_default(self) -> None:
        for n, p in self.opt.named:
            o = self.opt.offsets[n]
            p._grad_sink = self.opt.grad[o:o + p.numel()].view_as(p)

    def _use_sinks(self, which: int) -> None:
        for p, view in self._sinks[which]:
            p._grad_sink = view

    def _fwd_bwd(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor, two_streams: bool = True) -> torch.Tensor:
        """forward + L1 + backward; leaves the batch-mean gradient in opt.grad and returns the loss (device scalar)."""
        if not degrad_patch.is_cuda:
            return self._fwd_bwd_parts(degrad_patch, clean_patch, two_streams)
        from . import ops

        # the part streams already overlap; side streams inside them do not mix with capture (per-trainer choice,
        # scoped to this trainer's own forward/backward)
        with ops.side_streams(self.micro_streams == 1 and ops.USE_SIDE_STREAM):
            return self._fwd_bwd_parts(degrad_patch, clean_patch, two_streams)

    def _fwd_bwd_parts(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor, two_streams: bool) -> torch.Tensor:
        b = degrad_patch.shape[0]
        n = self._nparts(b)
        if not (two_streams and n > 1):
            loss, seed = self._part_loss(self.net(degrad_patch), clean_patch, 1.0)
            loss.backward(gradient=seed)
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
                losses.append(self._part_loss(self.net(degrad_patch[lo:hi]), clean_patch[lo:hi], (hi - lo) / b))
        for i, (loss, w) in enumerate(losses):
            with torch.cuda.stream(self._streams[i]):
                loss.backward(gradient=w)
        for st in self._streams[:n]:
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
        self._use_sinks(0)
        for i in range(1, len(losses)):
            ops.add_(self.opt.grad, self._grads[i])
        return self._total_loss(losses)             # mean over the batch = sum of the weighted parts

    # ---- gradient all-reduce overlapped with backward (reference: DDP's bucketed all-reduce, train.py:339) ----------
    # The forward is cut into three pieces (PromptIR.encode_levels | run_latent | decode); the backward then runs as
    # three segments - decoder side, latent, encoder levels - each leaving ONE contiguous range of the flat gradient
    # complete (FlatAdamW lays the parameters out piece by piece).  The all-reduce of a finished range is issued
    # asynchronously (RCCL works on its own stream) and overlaps the next segment; only the last and smallest range
    # (encoder levels: 9 % of the parameters, 13 MB) is exposed.  Every segment is its own hipGraph (the first one also
    # holds the forward); eager mode runs the same segments.
    def _staged_ok(self) -> bool:
        return self.opt.stages is not None and len(self.opt.stages) == 3 and hasattr(self.net, "encode_levels")

    def _nparts(self, b: int) -> int:
        """Part-batch streams for a batch of b (at most `micro_streams`): parts of eight samples, but two parts from eight
        samples on (measured in the graph, round 4, same box: batch 32 in 2 / 3 / 4 parts 92.6 / 92.6 / 91.7 ms; batch 8 in
        2 x 4 35.6 ms, 4 x 2 37.1 ms, one part 36.9 ms).  PIR_MIN_PART overrides the part size."""
        if self.min_part:
            return min(self.micro_streams, max(1, b // self.min_part))
        return min(self.micro_streams, max(b // 8, 2 if b >= 8 else 1))

    def _parts(self, b: int):
        n = self._nparts(b)
        return n, [b * i // n for i in range(n + 1)]

    def _fork(self, n: int, device):
        main = torch.cuda.current_stream(device)
        if n == 1:
            return main, [main]
        ev = torch.cuda.Event()
        ev.record(main)
        for st in self._streams[:n]:
            st.wait_event(ev)
        return main, self._streams[:n]

    @staticmethod
    def _join(main, streams) -> None:
        for st in streams:
            if st is not main:
                ev = torch.cuda.Event()
                ev.record(st)
                main.wait_event(ev)

    def _sum_range(self, stage: int, n: int) -> None:
        """Part gradients of one stage's range summed into the primary flat buffer (mean over the batch: the parts'
        losses carry their share of the batch as upstream gradient)."""
        from . import ops

        lo, hi = self.opt.stages[stage]
        for i in range(1, n):
            ops.add_(self.opt.grad[lo:hi], self._grads[i][lo:hi])

    def _seg_forward_and_decoder(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """Segment 0: forward of every part, loss, backward of the decoder piece.  Leaves stage 2's gradient range
        complete and returns the batch loss."""
        b = x.shape[0]
        n, bounds = self._parts(b)
        main, streams = self._fork(n, x.device)
        self._seg_state = []
        for i, st in enumerate(streams):
            lo, hi = bounds[i], bounds[i + 1]
            if n > 1:
                self._use_sinks(i)                   # captured by the autograd nodes of this forward
            with torch.cuda.stream(st):
                enc = self.net.encode_levels(x[lo:hi])
                enc_cut = [e.detach().requires_grad_() for e in enc]
                lat = self.net.run_latent(enc_cut[3])
                lat_cut = lat.detach().requires_grad_()
                y = self.net.decode(x[lo:hi], lat_cut, enc_cut[2], enc_cut[1], enc_cut[0])
                loss, w = self._part_loss(y, t[lo:hi], (hi - lo) / b)
                self._seg_state.append({"enc": enc, "enc_cut": enc_cut, "lat": lat, "lat_cut": lat_cut,
                                        "loss": loss, "w": w})
        for p, st in zip(self._seg_state, streams):
            with torch.cuda.stream(st):
                p["loss"].backward(gradient=p["w"])
        self._join(main, streams)
        if n > 1:
            self._use_sinks(0)
        self._sum_range(2, n)
        return self._total_loss([(p["loss"], p["w"]) for p in self._seg_state])   # mean over the batch

    def _seg_latent(self, device) -> None:
        """Segment 1: backward of the latent blocks (stage 1's range)."""
        n = len(self._seg_state)
        main, streams = self._fork(n, device)
        for p, st in zip(self._seg_state, streams):
            with torch.cuda.stream(st):
                p["lat"].backward(p["lat_cut"].grad)
        self._join(main, streams)
        self._sum_range(1, n)

    def _seg_encoder(self, device) -> None:
        """Segment 2: backward of the encoder levels (stage 0's range)."""
        n = len(self._seg_state)
        main, streams = self._fork(n, device)
        for p, st in zip(self._seg_state, streams):
            with torch.cuda.stream(st):
                torch.autograd.backward(list(p["enc"]), [c.grad for c in p["enc_cut"]])
        self._join(main, streams)
        self._sum_range(0, n)

    def _staged_eager(self, x: torch.Tensor, t: torch.Tensor, reduce_async=None) -> torch.Tensor:
        from . import ops

        with ops.side_streams(False):
            loss = self._seg_forward_and_decoder(x, t)
            if reduce_async:
                reduce_async(2)
            self._seg_latent(x.device)
            if reduce_async:
                reduce_async(1)
            self._seg_encoder(x.device)
        self._seg_state = None
        return loss.detach()

    def _capture_staged(self, x: torch.Tensor, t: torch.Tensor) -> None:
        from . import ops

        self._sx, self._st = x.clone(), t.clone()
        side = torch.cuda.Stream(x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                # warm-up outside any error handling (see _capture)
            for _ in range(2):
                self._staged_eager(self._sx, self._st)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
        try:
            with ops.side_streams(False):
                with torch.cuda.graph(graphs[0], capture_error_mode="thread_local"):
                    sloss = self._seg_forward_and_decoder(self._sx, self._st)
                pool = graphs[0].pool()              # later segments read tensors the first one produced
                with torch.cuda.graph(graphs[1], pool=pool, capture_error_mode="thread_local"):
                    self._seg_latent(x.device)
                with torch.cuda.graph(graphs[2], pool=pool, capture_error_mode="thread_local"):
                    self._seg_encoder(x.device)
        except RuntimeError as exc:
            msg = str(exc).lower()
            if not any(k in msg for k in ("captur", "graph")):
                raise
            raise _CaptureRefused(str(exc)) from exc
        self._seg_state = None
        self._graph, self._sloss = graphs, sloss
        self._graph_shape = tuple(x.shape)

    def _capture(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor) -> None:
        if self.staged:
            return self._capture_staged(degrad_patch, clean_patch)
        self._capture_whole(degrad_patch, clean_patch)

    def _capture_whole(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor) -> None:
        """hipGraph of forward + loss + backward on static input buffers.  The optimiser, the
        all-reduce and the weight re-split stay outside (their arguments change from step to step)."""
        self._sx, self._st = degrad_patch.clone(), clean_patch.clone()
        side = torch.cuda.Stream(degrad_patch.device)
        side.wait_stream(torch.cuda.current_stream())
        # Warm-up OUTSIDE any error handling: argument-check, workspace, out-of-memory or asynchronous HIP errors
        # of the step itself are defects and must propagate.
        with torch.cuda.stream(side):                # on a side stream: caches, workspaces, allocator pools
            for _ in range(2):
                self._fwd_bwd(self._sx, self._st)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            # thread_local: the RCCL watchdog thread of a multi-GPU job may query its events while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                sloss = self._fwd_bwd(self._sx, self._st)
        except RuntimeError as exc:
            # Only a capture-specific refusal (the runtime cannot capture / instantiate this stream work) falls back;
            # anything else is re-raised.  torch.cuda.graph.__exit__ has already ended the capture on every stream
            # that joined it.
            msg = str(exc).lower()
            if not any(k in msg for k in ("captur", "graph")):
                raise
            raise _CaptureRefused(str(exc)) from exc
        self._graph, self._sloss = graph, sloss
        self._graph_shape = tuple(degrad_patch.shape)

    def _ensure_graph(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor) -> None:
        if self._graph is not None and self._graph_shape == tuple(degrad_patch.shape):
            return
        try:
            self._capture(degrad_patch, clean_patch)
        except _CaptureRefused as exc:   # keep training, eagerly on one stream
            import warnings

            warnings.warn(f"hipGraph capture refused ({exc}); falling back to the eager single-stream step")
            torch.cuda.synchronize()
            self.graph, self._graph, self.micro_streams = False, None, 1
            self._seg_state = None
            self._use_sinks_default()

    def prepare(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor) -> None:
        """Optional: build the graph for this batch shape now (otherwise the first train_step does it)."""
        if self.graph and degrad_patch.is_cuda:
            self._ensure_graph(degrad_patch, clean_patch)

    def forward_backward(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor, between=None) -> torch.Tensor:
        """forward + L1 + backward of one batch (reference train.py:37-46 + Lightning's loss.backward()): leaves the
        local batch-mean gradient in the flat buffer `opt.grad` and returns the loss (device scalar).  hipGraph
        replay when enabled, eager launches otherwise.  Staged mode: `between(stage)` is called as soon as the
        backward segment that completes gradient range `stage` has been enqueued (stages 2 and 1; the caller handles
        the final range 0 itself)."""
        if not degrad_patch.is_cuda:
            return self._fwd_bwd(degrad_patch, clean_patch)
        from . import ops

        if self.graph and ops.lib.records is None:
            self._ensure_graph(degrad_patch, clean_patch)
        if self.graph and ops.lib.records is None:
            # No Python runs inside a replay, so `_split_weight`'s version check cannot: if any weight changed since
            # the pieces were last refreshed (load_state_dict, a checkpoint, dist.broadcast, an external optimiser),
            # re-split them all now (one launch).
            sig = ops.split_weights_signature()
            if sig != self._split_sig:
                ops.refresh_split_weights()
                self._split_sig = ops.split_weights_signature()
            if degrad_patch.data_ptr() != self._sx.data_ptr():   # (a caller may fill the static buffers itself)
                ops.copy_flat(ops._contig4(degrad_patch), self._sx)
            if clean_patch.data_ptr() != self._st.data_ptr():
                ops.copy_flat(ops._contig4(clean_patch), self._st)
            if self.staged:        # three segment graphs; `between(stage)` runs after the segment that completes `stage`
                self._graph[0].replay()
                loss = self._loss_out()
                if between:
                    between(2)
                self._graph[1].replay()
                if between:
                    between(1)
                self._graph[2].replay()
                return loss
            self._graph.replay()
            return self._loss_out()
        if self.staged and ops.lib.records is None:
            return self._staged_eager(degrad_patch, clean_patch, between)
        # eager; the instrumented (per-kernel timed) step stays on one stream
        return self._fwd_bwd(degrad_patch, clean_patch, two_streams=ops.lib.records is None)

    def static_inputs(self):
        """The captured graph's own input buffers (degraded, clean) once a graph exists: a loader that writes its batch
        straight into them (device-side degradation, crop and augmentation kernels) saves the per-step copy."""
        return (self._sx, self._st) if self._graph is not None else None

    def _loss_out(self) -> torch.Tensor:
        """The replayed step's loss in a tensor of its own, copied out of the graph's static scalar by the library's
        copy kernel (a `.clone()` is a runtime blit outside the library)."""
        from . import ops

        return ops.copy_flat(self._sloss, torch.empty_like(self._sloss))

    def train_step(self, degrad_patch: torch.Tensor, clean_patch: torch.Tensor, lr: Optional[float] = None):
        """reference train.py:37-46 (+ optimizer.step of Lightning's loop)."""
        if self.staged and degrad_patch.is_cuda and dist.is_initialized():
            from . import ops

            staged_now = ops.lib.records is None
        else:
            staged_now = False
        if staged_now:
            # asynchronous all-reduce of each gradient range as soon as its backward segment has been enqueued: RCCL
            # orders it behind the work already on this stream and runs it on its own stream beside the next segment
            pending = []

            def reduce_range(stage: int) -> None:
                lo, hi = self.opt.stages[stage]
                pending.append(dist.all_reduce(self.opt.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

            loss = self.forward_backward(degrad_patch, clean_patch, between=reduce_range)
            reduce_range(0)
            for work in pending:
                work.wait()
            scale = 1.0 / self.world
        else:
            loss = self.forward_backward(degrad_patch, clean_patch)
            scale = allreduce_mean_(self.opt.grad, self.world)
        self.opt.step(lr=lr, grad_scale=scale)
        if self.graph and self.opt.param.is_cuda:
            from . import ops

            ops.refresh_split_weights()
            self._split_sig = ops.split_weights_signature()
        return loss

    def checkpoint(self, epoch: int, lr: Optional[float] = None) -> dict:
        """A dict in the layout Lightning 2.0.1 writes for the reference's PromptIRModel (train.py:28-56, :334):
        `state_dict` keys are `net.<PromptIR key>`, `optimizer_states[0]` is a torch.optim.AdamW state_dict,
        `lr_schedulers[0]` the LinearWarmupCosineAnnealingLR state, plus the version / loops / callbacks keys
        `load_from_checkpoint` and `trainer.fit(ckpt_path=...)` look up (loop progress is not tracked here: `loops`
        and `callbacks` are empty, so Lightning restarts its counters from `epoch` / `global_step`)."""
        # ModelCheckpoint saves in on_train_epoch_end, AFTER the epoch's scheduler call (`_TrainingEpochLoop.advance`
        # steps an epoch-interval scheduler behind the last batch, see lightning_epoch_lr): at the epoch-N save the
        # scheduler has seen its construction-time step plus step(0) .. step(N), so last_epoch = N, _step_count = N + 2,
        # and `_last_lr` / the optimizer's lr hold closed_form(N) - the rate epoch N+1 will train with.  `lr` overrides
        # that rate (callers with their own schedule).
        cur_lr = warmup_cosine_lr(epoch, base_lr=self.opt.lr) if lr is None else lr
        sched = {"warmup_epochs": 15, "max_epochs": 150, "warmup_start_lr": 0.0, "eta_min": 0.0,
                 "base_lrs": [self.opt.lr], "last_epoch": epoch, "_step_count": epoch + 2, "verbose": False,
                 "_get_lr_called_within_step": False, "_last_lr": [cur_lr]}
        return {"epoch": epoch, "global_step": self.opt.steps, "pytorch-lightning_version": "2.0.1",
                "state_dict": {"net." + k: v.detach().clone() for k, v in self.net.state_dict().items()},
                "loops": {}, "callbacks": {}, "hparams_name": None, "hyper_parameters": {},
                "optimizer_states": [self.opt.torch_state_dict(self.net, cur_lr)], "lr_schedulers": [sched]}


def load_checkpoint_file(path: str) -> dict:
    """torch.load for trusted Lightning checkpoints (they hold non-tensor payload the `weights_only` default of recent
    torch releases rejects)."""
    return torch.load(path, map_location="cpu", weights_only=False)


def load_lightning_checkpoint(net: nn.Module, ckpt: dict) -> None:
    sd = ckpt.get("state_dict", ckpt)
    net.load_state_dict({k[4:] if k.startswith("net.") else k: v for k, v in sd.items()})


def init_distributed() -> tuple:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and os.environ.get("PIR_SHARE_GPU") == "1":
        # rehearsal of the N>1 path on a box with fewer GPUs than ranks: ranks share devices (RCCL refuses that, so
        # such a run also sets PIR_DIST_BACKEND=gloo; tests/test_drivers_gpu.py)
        local = local % torch.cuda.device_count()
    if (world > 1 or os.environ.get("PIR_FORCE_PG") == "1") and not dist.is_initialized():   # PIR_FORCE_PG: 1-rank test
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PIR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world
