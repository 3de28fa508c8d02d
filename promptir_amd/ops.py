"""Operator layer: torch tensors in, HIP kernels (C ABI, promptir_amd/_lib.py) underneath.

PyTorch supplies device memory, the current HIP stream and autograd bookkeeping only;
every FLOP of the PromptIR path runs in libpromptir_hip.so.  Tensors must live on a
ROCm device — there is no CPU or eager fallback (see `_require_gpu`).

Layout contract for activations: fp32 NCHW with stride(3)==1, stride(2)==W,
stride(1)==H*W; the batch stride is free, so channel slices of a larger buffer
(q/k/v inside qkv, halves of a concat) are passed without copies.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check


class _TimedLib:
    """Pass-through to the C ABI.  With `start_timing()` every entry point is bracketed by HIP
    events recorded on the stream the kernels are launched on (torch's current stream), which
    is how bench.py measures per-kernel launch durations live."""

    def __init__(self, raw):
        self._raw = raw
        self.records = None

    def start_timing(self):
        self.records = []

    def stop_timing(self):
        recs, self.records = self.records, None
        torch.cuda.synchronize()
        out = []
        for name, start, end, work, nbytes in recs:
            out.append((name, start.elapsed_time(end) * 1e-3, work, nbytes))
        return out

    @staticmethod
    def _bytes(name, args):
        """Algorithmic HBM bytes of one call of the GEMM families (operands read once, result written once)."""
        if name in ("pir_gemm_nn", "pir_gemm_nn_ws"):
            g = args[0]._obj
            o = g.O1 * g.O2
            shared = g.a_s1 == 0 and g.a_s2 == 0
            return 4.0 * o * (g.K * g.N + g.M * g.N * (2 if g.R else 1)) + 4.0 * g.M * g.K * (1 if shared else o)
        if name in ("pir_gemm_nt", "pir_gemm_nt_partials"):
            g = args[0]._obj
            return 4.0 * g.O1 * g.O2 * g.BR * (g.M1 + g.M2) * g.N
        if name == "pir_gemm_nt_group":
            return sum(4.0 * g.BR * (g.M1 + g.M2) * g.N for g in list(args[0])[:args[1]])
        if name == "pir_conv1x1_dgrad_ln_bwd":     # dy read, x and the residual gradient read, dx written, statistics read
            k, (b, c, hw) = args[4], args[18:21]
            return 4.0 * b * hw * (k + 3 * c + 2)
        if name == "pir_ln_conv1x1_fwd":           # x read, y written (+ statistics)
            b, m, k, hw = args[10:14]
            return 4.0 * b * hw * (k + m + 2)
        if name == "pir_conv1x1_wgrad_ln":         # dy and x read (+ statistics)
            b, cout, cin, hw = args[11:15]
            return 4.0 * b * hw * (cout + cin + 2)
        if name == "pir_conv3x3_wgrad":            # dy and x read once
            b, cout, cin, h, w = args[5:10]
            return 4.0 * b * h * w * (cout + cin)
        if name in ("pir_conv3x3", "pir_conv3x3_x3", "pir_conv3x3_x3_ws"):   # x read, y written (+ residual)
            b, m, k, h, w = args[11:16] if name == "pir_conv3x3" else args[8:13]
            return 4.0 * b * h * w * (k + m)
        # ---- depthwise stencils (HBM roofline): algorithmic planes x 4 bytes, SURVEY 8(d) / DESIGN 4
        if name == "pir_dwconv3x3_sumsq":          # x read, y written
            b, c, h, w = args[9:13]
            return 4.0 * 2 * b * c * h * w
        if name == "pir_dwconv3x3":
            b, c, h, w = args[6:10]
            return 4.0 * 2 * b * c * h * w
        if name == "pir_dwconv3x3_gate":           # both halves of x read, g written: 3 planes per hidden channel
            b, hid, h, w = args[5:9]
            return 4.0 * 3 * b * hid * h * w
        if name == "pir_dwconv3x3_bwd":            # dy and x read, dx written
            b, c, h, w = args[10:14]
            return 4.0 * 3 * b * c * h * w
        if name == "pir_gdfn_dwconv_bwd":          # x (2 planes) and dg read, dx (2 planes) written: 5 per hidden channel
            b, hid, h, w = args[10:14]
            return 4.0 * 5 * b * hid * h * w
        if name == "pir_mdta_dqk":                 # q, k read; dq, dk written
            b, heads, c, hw = args[9:13]
            return 4.0 * 4 * b * heads * c * hw
        return 0.0

    @staticmethod
    def _work(name, args):
        """Algorithmic FLOPs of one call for the MFMA families (0 for the streaming kernels)."""
        if name in ("pir_gemm_nn", "pir_gemm_nn_ws"):
            g = args[0]._obj
            return 2.0 * g.M * g.K * g.N * g.O1 * g.O2
        if name in ("pir_gemm_nt", "pir_gemm_nt_partials"):
            g = args[0]._obj
            return 2.0 * g.M1 * g.M2 * g.N * g.O1 * g.O2 * g.BR
        if name == "pir_gemm_nt_group":
            return sum(2.0 * g.M1 * g.M2 * g.N * g.BR for g in list(args[0])[:args[1]])
        if name == "pir_conv1x1_dgrad_ln_bwd":
            k, (b, c, hw) = args[4], args[18:21]
            return 2.0 * c * k * hw * b
        if name == "pir_ln_conv1x1_fwd":
            b, m, k, hw = args[10:14]
            return 2.0 * m * k * hw * b
        if name == "pir_conv1x1_wgrad_ln":
            b, cout, cin, hw = args[11:15]
            return 2.0 * cout * cin * hw * b
        if name == "pir_conv3x3":
            b, m, k, h, w = args[11:16]
            return 2.0 * 9 * m * k * h * w * b
        if name in ("pir_conv3x3_x3", "pir_conv3x3_x3_ws"):
            b, m, k, h, w = args[8:13]
            return 2.0 * 9 * m * k * h * w * b
        if name == "pir_conv3x3_wgrad":
            b, cout, cin, h, w = args[5:10]
            return 2.0 * 9 * cout * cin * h * w * b
        if name == "pir_mdta_dqk":                 # dq = dG k and dk = dG^T q
            b, heads, c, hw = args[9:13]
            return 2.0 * 2 * c * c * hw * b * heads
        return 0.0

    def __getattr__(self, name):
        fn = getattr(self._raw, name)
        if self.records is None or name.endswith("_floats") or name.endswith("_plan") or name.endswith("_needed"):   # host-only queries
            return fn

        def timed(*args):
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            status = fn(*args)
            end.record()
            if status != 1000:      # 1000 = shape not served, nothing launched (the caller runs the unfused pair)
                base = {"pir_gemm_nn_ws": "pir_gemm_nn", "pir_conv3x3_x3_ws": "pir_conv3x3_x3"}.get(name, name)   # same op, with scratch
                label = base + "@" + _TAG[-1] if _TAG else base
                self.records.append((label, start, end, self._work(name, args), self._bytes(name, args)))
            return status

        return timed


_TAG = []   # innermost `tagged(...)` scope: the instrumented step files a call under "<entry point>@<tag>"


class tagged:
    """Scope whose C-ABI calls the instrumented step reports as a family of their own (bench.py: the MDTA contractions
    are generic pir_gemm_nn / pir_gemm_nt calls; north_star wants their MFMA fraction by itself)."""

    def __init__(self, tag: str):
        self.tag = tag

    def __enter__(self):
        _TAG.append(self.tag)

    def __exit__(self, *exc):
        _TAG.pop()
        return False


def _tag_calls(tag: str):
    def deco(fn):
        def wrapped(*a, **k):
            with tagged(tag):
                return fn(*a, **k)
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        return wrapped
    return deco


lib = _TimedLib(_lib.lib)
_KNOBS_SET = {}   # knob -> value of every pir_tune_set this process made through `tune_set` (reported by bench.py)


def tune_set(knob: int, value: int) -> None:
    check(_lib.lib.pir_tune_set(int(knob), int(value)), "pir_tune_set(%s=%s)" % (knob, value))
    _KNOBS_SET[int(knob)] = int(value)


def effective_switches() -> dict:
    """Everything that can change which kernels the product path runs: the module switches below, every PIR_*
    environment variable and every tuning knob set in this process.  bench.py prints it with its line."""
    mod = {k: globals()[k] for k in ("USE_X3", "MDTA_FOLD", "LN_FOLD", "LN_TRAIN", "DGRAD_LN", "MDTA_DQK", "MDTA_FOLD_MIN_HW",
                                      "USE_SIDE_STREAM", "SOFTMAX_PARTS", "CAT_INPLACE", "SHUFFLE_FOLD", "NT_GROUP") if k in globals()}
    return {"switches": mod, "env": {k: v for k, v in sorted(_os.environ.items()) if k.startswith("PIR_")},
            "knobs_set": {str(k): v for k, v in sorted(_KNOBS_SET.items())}}

# A/B switches for development (tools/, tests): PIR_NN_X3 = 0 | 1 forces the fp32-MFMA / bf16x3 gemm_nn path
import os as _os

if _os.environ.get("PIR_NN_X3") is not None:
    tune_set(3, int(_os.environ["PIR_NN_X3"]))
if _os.environ.get("PIR_NT_X3") is not None:
    tune_set(4, int(_os.environ["PIR_NT_X3"]))
for _kv in filter(None, _os.environ.get("PIR_KNOBS", "").split(",")):   # "20=0,25=0": any pir_tune_set knob (tools A/B)
    _k, _v = _kv.split("=")
    tune_set(int(_k), int(_v))


# ----------------------------------------------------------------------------- plumbing
def _require_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "promptir_amd: the HIP path needs tensors on a ROCm device (got %s); there is no CPU "
                "fallback — use oracle/ for CPU reference results" % t.device)
        if t.dtype != torch.float32:
            raise RuntimeError(f"promptir_amd: fp32 only (the reference uses no AMP), got {t.dtype}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _planes(t: torch.Tensor) -> torch.Tensor:
    """Return `t` if it satisfies the plane layout (free batch stride), else a contiguous copy made by the library's
    own gather kernel (pir_copy_strided4) - never an ATen copy: the step holds pir_* kernels only."""
    if t.dim() != 4:
        raise RuntimeError(f"expected a 4-d NCHW tensor, got shape {tuple(t.shape)}")
    b, c, h, w = t.shape
    st = t.stride()
    ok = (w == 1 or st[3] == 1) and (h == 1 or st[2] == w) and (c == 1 or st[1] == h * w)
    if ok and (b == 1 or st[0] >= c * h * w):
        return t
    return _gather_contiguous(t)


def _gather_contiguous(t: torch.Tensor) -> torch.Tensor:
    _require_gpu(t)
    b, c, h, w = t.shape
    out = torch.empty((b, c, h, w), dtype=torch.float32, device=t.device)
    st = t.stride()
    check(lib.pir_copy_strided4(t.data_ptr(), st[0], st[1], st[2], st[3], out.data_ptr(), b, c, h, w, _stream()),
          "pir_copy_strided4")
    return out


def _contig4(t: torch.Tensor) -> torch.Tensor:
    """Fully contiguous NCHW (the flat-indexed kernels: loss)."""
    return t if t.is_contiguous() else _gather_contiguous(t)


def copy_flat(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst[:] = src for contiguous fp32 buffers of equal size through pir_copy_planes (a tensor.copy_() would be a
    runtime blit kernel outside the library)."""
    _require_gpu(src, dst)
    if not (src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()):
        raise RuntimeError("copy_flat: contiguous buffers of equal size expected")
    check(lib.pir_copy_planes(src.data_ptr(), src.numel(), dst.data_ptr(), dst.numel(), 0, 1, src.numel(), _stream()),
          "pir_copy_planes")
    return dst


def _bs(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else t.shape[1] * t.shape[2] * t.shape[3]


_WS = {}
_WS_RETIRED = []   # outgrown buffers stay alive: a captured graph may still hold their addresses


def workspace(nfloats: int, device: torch.device, slot: str = "main", own: bool = False) -> torch.Tensor:
    """Grow-only scratch buffer per (device, slot).  All kernels of one op are enqueued on the
    current stream in order, so one buffer per slot is race-free on a single stream.
    Inside a `deferred_reductions()` scope the partial sums a kernel leaves here are read only at the flush: every
    request then gets its own piece of a bump-allocated arena instead of the one reused buffer."""
    st = _stream()
    # (large partial sets are reduced at once by the library even inside the scope - pir_reduce_defer_limit - and keep the
    # reused, cache-resident buffer; `own`: the caller needs a piece nobody else gets before the flush regardless)
    if slot == "main" and _DEFER.get(st, 0) > 0:
        if own or 4 * int(nfloats) <= DEFER_LIMIT_BYTES:
            _defer_switch(st, True)
            return _arena_take(int(nfloats), device, st)
        _defer_switch(st, False)      # the reduction that uses the reused buffer must run before the next call overwrites it
    key = (device.index, slot, st)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = torch.empty(max(int(nfloats), 1 << 16), dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


# ---- deferred, batched second stages of the parameter-gradient reductions (csrc/reduce_batch.hip) ----------------
DEFER_REDUCE = _os.environ.get("PIR_DEFER_REDUCE", "1") != "0"
DEFER_LIMIT_BYTES = int(_os.environ.get("PIR_DEFER_LIMIT", str(4 << 20)))
_lib.lib.pir_reduce_defer_limit(DEFER_LIMIT_BYTES)
_DEFER = {}     # stream handle -> nesting depth of open deferral scopes
_ARENAS = {}    # (device index, stream) -> {"bufs": [tensors], "cur": index, "off": floats used in bufs[cur]}
_DEFER_KEEP = {}   # stream -> tensors that hold queued partial sums (kept alive until the flush)


_DEFER_ON = {}   # stream -> what the library was last told (pir_reduce_defer)


def _defer_switch(st: int, on: bool) -> None:
    if _DEFER_ON.get(st, False) != on:
        check(_lib.lib.pir_reduce_defer(st, int(on)), "pir_reduce_defer")
        _DEFER_ON[st] = on


def _arena_take(nfloats: int, device: torch.device, st: int) -> torch.Tensor:
    a = _ARENAS.setdefault((device.index, st), {"bufs": [], "cur": 0, "off": 0})
    n = (max(nfloats, 1) + 63) // 64 * 64
    while True:
        if a["cur"] < len(a["bufs"]):
            buf = a["bufs"][a["cur"]]
            if a["off"] + n <= buf.numel():
                out = buf[a["off"]:a["off"] + n]
                a["off"] += n
                return out
            a["cur"] += 1
            a["off"] = 0
            continue
        # chunks are never freed or replaced (captured graphs hold their addresses); sized for a block's worth of partials
        a["bufs"].append(torch.empty(max(n, 1 << 24), dtype=torch.float32, device=device))


class deferred_reductions:
    """Scope in which the reductions behind parameter gradients are queued by the library instead of launched
    (pir_reduce_defer); `flush_reductions()` then runs them in batched launches.  Only for results nothing on the
    device reads before the flush."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled) and DEFER_REDUCE

    def __enter__(self):
        if self.enabled:
            self.st = _stream()
            depth = _DEFER.get(self.st, 0)
            if depth == 0:
                _defer_switch(self.st, True)
            _DEFER[self.st] = depth + 1
        return self

    def __exit__(self, *exc):
        if self.enabled:
            depth = _DEFER[self.st] - 1
            _DEFER[self.st] = depth
            if depth == 0:
                _defer_switch(self.st, False)
        return False


class immediate_reductions:
    """Inside a deferral scope: reductions whose result the NEXT kernel reads (dW_eff, dattn of the MDTA backward)."""

    def __enter__(self):
        self.st = _stream()
        self.depth = _DEFER.get(self.st, 0)
        if self.depth:
            _defer_switch(self.st, False)
            _DEFER[self.st] = 0
        return self

    def __exit__(self, *exc):
        if self.depth:
            _DEFER[self.st] = self.depth
            _defer_switch(self.st, True)
        return False


def keep_until_flush(t: torch.Tensor) -> torch.Tensor:
    """A tensor of partial sums allocated by the caller (not the arena) whose reduction may be queued."""
    st = _stream()
    if _DEFER.get(st, 0) > 0:
        _defer_switch(st, True)
        _DEFER_KEEP.setdefault(st, []).append(t)
    return t


def flush_reductions() -> None:
    """Run every reduction queued on the current stream (batched launches) and recycle the partial-sum arena."""
    st = _stream()
    check(lib.pir_reduce_flush(st), "pir_reduce_flush")
    _DEFER_KEEP.pop(st, None)
    for (dev, s), a in _ARENAS.items():
        if s == st:
            a["cur"], a["off"] = 0, 0


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---- producers that write straight into a concat buffer (net/model.py:341,347,353,359,365,370: torch.cat along C) ----
CAT_INPLACE = _os.environ.get("PIR_CAT_INPLACE", "1") != "0"


def _alias(buf: torch.Tensor, c0: int, c: int) -> torch.Tensor:
    """Channels [c0, c0 + c) of the NCHW buffer `buf` as a tensor that SHARES its memory without being an autograd
    view of it (Tensor.set_): safe to return from an autograd Function's forward as a fresh output."""
    b, ctot, h, w = buf.shape
    return torch.empty(0, dtype=buf.dtype, device=buf.device).set_(
        buf.untyped_storage(), buf.storage_offset() + c0 * h * w, (b, c, h, w), (ctot * h * w, h * w, w, 1))


class OutSlot:
    """Destination of a producer's output: channels [c0, c0 + c) of a pre-allocated concat buffer.  Travels through
    autograd.Function.apply as a plain Python object (autograd never sees the buffer)."""

    def __init__(self, buf: torch.Tensor, c0: int, c: int):
        self.buf, self.c0, self.c = buf, c0, c

    def tensor(self, shape) -> Optional[torch.Tensor]:
        b, _, h, w = self.buf.shape
        if tuple(shape) != (b, self.c, h, w):
            raise RuntimeError(f"OutSlot: producer output {tuple(shape)} does not fit slot {(b, self.c, h, w)}")
        return _alias(self.buf, self.c0, self.c)


_CAT_BUFFERS = {}   # storage address -> concat buffer (lets the decoder find the buffer a skip tensor lives in)


def new_cat_buffer(like: torch.Tensor, channels: int, h: int, w: int) -> torch.Tensor:
    """A registered concat buffer.  The registry holds it strongly (the producers' outputs are storage aliases, the
    tensor object itself would die with the frame that allocated it); entries whose storage nobody else uses any more
    are dropped here, at the next allocation."""
    for key in [k for k, t in _CAT_BUFFERS.items() if torch._C._storage_Use_Count(t.untyped_storage()._cdata) <= 2]:
        del _CAT_BUFFERS[key]          # 2 = the registered tensor + the storage wrapper made for the query
    buf = torch.empty((like.shape[0], channels, h, w), dtype=torch.float32, device=like.device)
    _CAT_BUFFERS[buf.untyped_storage().data_ptr()] = buf
    return buf


def cat_buffer_of(t: torch.Tensor) -> Optional[torch.Tensor]:
    """The registered concat buffer whose memory `t` (or a detached alias of it) lives in, if any."""
    return _CAT_BUFFERS.get(t.untyped_storage().data_ptr())


# ----------------------------------------------------------------------------- raw kernels (no autograd)
def gemm_nn(A: torch.Tensor, a_batch: Tuple[int, int], a_sm: int, a_sk: int,
            X: torch.Tensor, x_off: int, x_batch: Tuple[int, int], ldx: int,
            Y: torch.Tensor, y_off: int, y_batch: Tuple[int, int], ldy: int,
            M: int, K: int, N: int, O1: int, O2: int,
            R: Optional[torch.Tensor] = None, r_off: int = 0, r_batch: Tuple[int, int] = (0, 0), ldr: int = 0,
            rowscale: Optional[torch.Tensor] = None, rs_batch: Tuple[int, int] = (0, 0),
            A3: Optional[torch.Tensor] = None, a3_kp: int = 0) -> None:
    g = _lib.GemmNN()
    if A3 is not None:
        g.A3, g.a3_kp = A3.data_ptr(), a3_kp
    g.A, g.a_s1, g.a_s2, g.a_sm, g.a_sk = A.data_ptr(), a_batch[0], a_batch[1], a_sm, a_sk
    g.X, g.x_s1, g.x_s2, g.ldx = X.data_ptr() + 4 * x_off, x_batch[0], x_batch[1], ldx
    g.Y, g.y_s1, g.y_s2, g.ldy = Y.data_ptr() + 4 * y_off, y_batch[0], y_batch[1], ldy
    if R is not None:
        g.R, g.r_s1, g.r_s2, g.ldr = R.data_ptr() + 4 * r_off, r_batch[0], r_batch[1], ldr
    if rowscale is not None:
        g.rowscale, g.rs_s1, g.rs_s2 = rowscale.data_ptr(), rs_batch[0], rs_batch[1]
    g.M, g.K, g.N, g.O1, g.O2 = M, K, N, O1, O2
    need = lib.pir_gemm_nn_ws_floats(C.byref(g)) if A3 is not None else 0
    if need:      # an underfilled deep-k product: split over k, partial sums in a scratch slot of their own
        ws = workspace(need, X.device, slot="gemm_nn")
        check(lib.pir_gemm_nn_ws(C.byref(g), ws.data_ptr(), ws.numel(), _stream()), "pir_gemm_nn_ws")
        return
    check(lib.pir_gemm_nn(C.byref(g), _stream()), "pir_gemm_nn")


def gemm_nt(X: torch.Tensor, x_off: int, x_str: Tuple[int, int, int], ldx: int,
            Y: torch.Tensor, y_off: int, y_str: Tuple[int, int, int], ldy: int,
            G: torch.Tensor, g_off: int, g_str: Tuple[int, int, int],
            M1: int, M2: int, N: int, O1: int, O2: int, BR: int,
            shift: Optional[Tuple[int, int, int, int]] = None, alpha: float = 1.0, accumulate: bool = False) -> None:
    g = _lib.GemmNT()
    g.X, g.x_s1, g.x_s2, g.x_sr, g.ldx = X.data_ptr() + 4 * x_off, x_str[0], x_str[1], x_str[2], ldx
    g.Y, g.y_s1, g.y_s2, g.y_sr, g.ldy = Y.data_ptr() + 4 * y_off, y_str[0], y_str[1], y_str[2], ldy
    g.G, g.g_so, g.g_si, g.g_sj = G.data_ptr() + 4 * g_off, g_str[0], g_str[1], g_str[2]
    g.M1, g.M2, g.N, g.O1, g.O2, g.BR = M1, M2, N, O1, O2, BR
    if shift is not None:
        g.shift_dh, g.shift_dw, g.H, g.W = shift
    if _DEFER.get(_stream(), 0) > 0:    # a piece of its own until the flush: sized for THIS call's split count
        nws = lib.pir_gemm_nt_ws_needed(C.byref(g))
    else:                               # the one reused buffer: worst case over the plans
        nws = lib.pir_gemm_nt_ws_floats(M1, M2, N, O1 * O2, BR)
    ws = workspace(nws, X.device)
    g.ws, g.ws_floats = ws.data_ptr(), ws.numel()
    g.alpha, g.accumulate = alpha, int(accumulate)
    check(lib.pir_gemm_nt(C.byref(g), _stream()), "pir_gemm_nt")


def gemm_nt_partials(X, x_off, x_str, ldx, Y, y_off, y_str, ldy, M1: int, M2: int, N: int, O1: int, O2: int, BR: int):
    """The split-K product of `gemm_nt` WITHOUT its second stage (pir_gemm_nt_partials): returns (workspace, splits) - the
    slices ws[s][o][i][j] stay in the stream's reused workspace, valid until the next call on this stream takes it; the
    consumer launched next reads and adds them (MDTA softmax kernels)."""
    g = _lib.GemmNT()
    g.X, g.x_s1, g.x_s2, g.x_sr, g.ldx = X.data_ptr() + 4 * x_off, x_str[0], x_str[1], x_str[2], ldx
    g.Y, g.y_s1, g.y_s2, g.y_sr, g.ldy = Y.data_ptr() + 4 * y_off, y_str[0], y_str[1], y_str[2], ldy
    g.G, g.g_so, g.g_si, g.g_sj = None, 0, 0, 0
    g.M1, g.M2, g.N, g.O1, g.O2, g.BR = M1, M2, N, O1, O2, BR
    st = _stream()
    ws = _WS_main(lib.pir_gemm_nt_ws_floats(M1, M2, N, O1 * O2, BR), X.device, st)
    g.ws, g.ws_floats = ws.data_ptr(), ws.numel()
    g.alpha, g.accumulate = 1.0, 0
    splits = C.c_int(0)
    check(lib.pir_gemm_nt_partials(C.byref(g), C.byref(splits), st), "pir_gemm_nt_partials")
    return ws, splits.value


def _WS_main(nfloats: int, device: torch.device, st: int) -> torch.Tensor:
    """The stream's one reused scratch buffer (never an arena piece)."""
    key = (device.index, "main", st)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = torch.empty(max(int(nfloats), 1 << 16), dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


# gram / dattn split-K slices summed by the softmax kernels instead of by a reduction launch (-140 launches per step,
# bit-identical).  OPT-IN: measured +0.4 ms at batch 32, +0.25 ms at batch 8, +0.08 ms per batch-8 inference on one box
# (three interleaved pairs): the reduction launch spreads its loads over hundreds of workgroups, the softmax kernel adds
# the slices inside the block's dependent chain.
SOFTMAX_PARTS = _os.environ.get("PIR_SOFTMAX_PARTS", "0") != "0"
NT_GROUP = _os.environ.get("PIR_NT_GROUP", "1") != "0"     # low-resolution weight gradients of a block in one launch
NT_GROUP_MAX_HW = int(_os.environ.get("PIR_NT_GROUP_MAX_HW", "1024"))


def conv1x1_wgrad_group(items) -> None:
    """[(dy, x, dw)]: dW_k = sum_b dy_k[b] x_k[b]^T for up to four 1x1 convolutions in ONE launch (pir_gemm_nt_group):
    the weight gradients of a block at the 32^2 / 16^2 levels, each of which alone has too few output tiles for the chip."""
    if _DEFER.get(_stream(), 0) <= 0:
        raise RuntimeError("conv1x1_wgrad_group: inside a deferred_reductions() scope only (one workspace piece per problem)")
    probs = (_lib.GemmNT * len(items))()
    keep = []
    for k, (dy, x, dw) in enumerate(items):
        dy, x = _planes(dy), _planes(x)
        b, cout, h, wd = dy.shape
        cin, hw = x.shape[1], h * wd
        g = probs[k]
        g.X, g.x_s1, g.x_s2, g.x_sr, g.ldx = dy.data_ptr(), 0, 0, _bs(dy), hw
        g.Y, g.y_s1, g.y_s2, g.y_sr, g.ldy = x.data_ptr(), 0, 0, _bs(x), hw
        g.G, g.g_so, g.g_si, g.g_sj = dw.data_ptr(), 0, cin, 1
        g.M1, g.M2, g.N, g.O1, g.O2, g.BR = cout, cin, hw, 1, 1, b
        g.alpha, g.accumulate = 1.0, 0
        keep += [dy, x]
    for k in range(len(items)):      # every problem's own piece, sized for the group's common split count
        ws = workspace(lib.pir_gemm_nt_group_ws_needed(probs, len(items), k), items[k][0].device, own=True)
        probs[k].ws, probs[k].ws_floats = ws.data_ptr(), ws.numel()
        keep.append(ws)
    check(lib.pir_gemm_nt_group(probs, len(items), _stream()), "pir_gemm_nt_group")


# ---- bf16x3 weight pieces (pir_split_bf16x3), cached per weight tensor OBJECT and orientation.
# Keyed weakly by the Parameter object (an address could be recycled by the allocator for another model's
# weights); valid while (storage address, torch version counter, generation) are unchanged.  The version
# counter covers load_state_dict and torch optimisers; the fused AdamW kernel writes through raw pointers,
# which torch cannot see, so promptir_amd.train calls `weights_changed()` after every step (so must any
# caller that edits weights through `.data` / raw pointers).
import weakref as _weakref

USE_X3 = _os.environ.get("PIR_X3", "1") != "0"
_SPLIT = {}   # id(tensor) -> (weakref to the tensor, {dgrad: (validity, pieces)}); identity-keyed (tensor == is elementwise)
_WEIGHT_GEN = [0]


def weights_changed() -> None:
    _WEIGHT_GEN[0] += 1


_BATCH = {"entries": [], "dirty": True, "descs": None, "blocks": None, "nblocks": 0, "ptrs": None}


def _split_args(w: torch.Tensor, dgrad: bool, taps: bool):
    cout, cin = w.shape[0], w.shape[1]
    M, K = (cin, cout) if dgrad else (cout, cin)
    if taps:   # w[cout][cin][3][3]; input gradient: transposed channels and 180-degree rotated taps
        sm, sk = (9, cin * 9) if dgrad else (cin * 9, 9)
    else:
        sm, sk = (1, cin) if dgrad else (cin, 1)
    return M, K, sm, sk


def _refresh_split_batch() -> None:
    """Re-split every registered weight with ONE launch (pir_split_bf16x3_batch) and mark all entries current."""
    import numpy as np

    # strong references for the duration of the call: building the lists below allocates, an allocation can run the
    # garbage collector, and a collected module's weights would turn their weak references to None under our feet
    live = [(e, e["w"]()) for e in _BATCH["entries"]]
    live = [(e, w) for e, w in live if w is not None]
    ents = [e for e, _ in live]
    ptrs = [(w.data_ptr(), e["buf"].data_ptr()) for e, w in live]
    if _BATCH["dirty"] or ptrs != _BATCH["ptrs"] or len(ents) != len(_BATCH["entries"]):
        if not live:
            _BATCH["entries"], _BATCH["ptrs"], _BATCH["nblocks"], _BATCH["dirty"] = [], [], 0, False
            return
        descs = (_lib.SplitDesc * len(ents))()
        blocks = []
        for i, (e, w) in enumerate(live):
            M, K, sm, sk = _split_args(w, e["dgrad"], e["taps"])
            d = descs[i]
            d.W, d.out, d.st, d.sm, d.sk = w.data_ptr(), e["buf"].data_ptr(), 1, sm, sk
            d.M, d.K, d.taps, d.flip = M, K, int(e["taps"]), int(e["taps"] and e["dgrad"])
            total = (9 if e["taps"] else 1) * M * ((K + 15) // 16 * 16)
            blocks += [(i, c) for c in range((total + 4095) // 4096)]
        dev = ents[0]["buf"].device
        raw = np.frombuffer(bytes(descs), dtype=np.uint8).copy()
        _BATCH["descs"] = torch.from_numpy(raw).to(dev)
        _BATCH["blocks"] = torch.tensor(blocks, dtype=torch.int32, device=dev).contiguous()
        _BATCH["nblocks"], _BATCH["ptrs"], _BATCH["entries"], _BATCH["dirty"] = len(blocks), ptrs, ents, False
    if not live:
        return
    check(lib.pir_split_bf16x3_batch(_BATCH["descs"].data_ptr(), _BATCH["blocks"].data_ptr(), _BATCH["nblocks"], _stream()),
          "pir_split_bf16x3_batch")
    gen = _WEIGHT_GEN[0]
    for e, w in live:
        e["slot"][e["skey"]] = ((w.data_ptr(), w._version, gen), e["buf"])


def refresh_split_weights() -> None:
    """Re-split every registered weight now (one launch).  For callers that replay a captured graph: no Python
    runs inside the replay, so the pieces must be current before it starts."""
    if _BATCH["entries"]:
        _refresh_split_batch()


def split_weights_signature():
    """Cheap fingerprint of everything the cached bf16x3 pieces depend on: the raw-pointer generation
    (`weights_changed`) plus address and torch version counter of every registered weight (load_state_dict,
    torch optimisers, dist.broadcast into a flat buffer all bump the counter).  Graph-replaying callers compare
    it with the value at their last refresh."""
    sig = _WEIGHT_GEN[0]
    for e in _BATCH["entries"]:
        w = e["w"]()
        if w is not None:
            sig = (sig * 1000003 + w.data_ptr() + 7919 * w._version) & 0xFFFFFFFFFFFFFFFF
    return sig


def _split_weight(w: torch.Tensor, dgrad: bool, taps: bool = False):
    """bf16x3 pieces of a 1x1 (taps=False) or dense 3x3 (taps=True) weight in forward / input-gradient orientation.
    The first request for a weight splits it alone and registers it; afterwards a stale entry (optimiser step)
    refreshes ALL registered weights in one launch."""
    M, K, sm, sk = _split_args(w, dgrad, taps)
    kp = (K + 15) // 16 * 16
    ver = (w.data_ptr(), w._version, _WEIGHT_GEN[0])
    key = id(w)
    rec = _SPLIT.get(key)
    if rec is None or rec[0]() is not w:
        def _gone(_r, _k=key):
            _SPLIT.pop(_k, None)
            _BATCH["dirty"] = True
        rec = (_weakref.ref(w, _gone), {})
        _SPLIT[key] = rec
    slot = rec[1]
    skey = (dgrad, taps)
    ent = slot.get(skey)
    if ent is not None and ent[0] == ver:
        return ent[1], kp
    if ent is not None and ent[1].device == w.device:
        _refresh_split_batch()
        ent = slot.get(skey)
        if ent is not None and ent[0] == ver:
            return ent[1], kp
    buf = torch.empty((27 if taps else 3) * M * kp, dtype=torch.bfloat16, device=w.device)
    if taps:
        check(lib.pir_split_bf16x3_taps(w.data_ptr(), M, K, 1, sm, sk, int(dgrad), buf.data_ptr(), _stream()),
              "pir_split_bf16x3_taps")
    else:
        check(lib.pir_split_bf16x3(w.data_ptr(), M, K, sm, sk, buf.data_ptr(), _stream()), "pir_split_bf16x3")
    slot[skey] = (ver, buf)
    _BATCH["entries"] = [e for e in _BATCH["entries"] if not (e["w"]() is w and e["skey"] == skey)]
    _BATCH["entries"].append({"w": rec[0], "dgrad": dgrad, "taps": taps, "buf": buf, "slot": slot, "skey": skey})
    _BATCH["dirty"] = True
    return buf, kp


def conv1x1_forward(x: torch.Tensor, w: torch.Tensor, residual: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[b] = W x[b] (+ residual[b]);  w is [Cout, Cin] or [Cout, Cin, 1, 1]."""
    x = _planes(x)
    b, cin, h, wd = x.shape
    cout = w.shape[0]
    hw = h * wd
    if out is None:
        out = torch.empty((b, cout, h, wd), dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = _planes(residual)
    a3, kp = _split_weight(w, dgrad=False) if USE_X3 else (None, 0)
    gemm_nn(w, (0, 0), cin, 1, x, 0, (_bs(x), 0), hw, out, 0, (_bs(out), 0), hw, cout, cin, hw, b, 1,
            R=residual, r_batch=(_bs(residual), 0) if residual is not None else (0, 0), ldr=hw, A3=a3, a3_kp=kp)
    return out


_NOT_SERVED = {}   # (kernel, shape, alignment) -> True once the library answered 1000 ("shape not served"): the fused entry
# points are tried once per shape; afterwards the caller goes straight to the unfused pair without allocating the fused
# call's outputs first.  Tuning knobs that change what is served are set before the first call.


def ln_conv1x1_forward(x: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, w: torch.Tensor, stats: bool = False):
    """y[b] = W LayerNorm_c(x[b]) with the channel LayerNorm applied on load (pir_ln_conv1x1_fwd): the normalised tensor
    is never written.  no_grad forward: nothing needs it afterwards; training (`stats`: returns (y, mean, rstd)): the weight
    gradient normalises x on load again (conv1x1_wgrad_ln) and the LayerNorm backward reads x and the statistics.  None when
    the kernel does not serve the shape - the caller then runs layernorm_forward + conv1x1_forward."""
    if not USE_X3 or ln_b is None:
        return None
    x = _planes(x)
    b, cin, h, wd = x.shape
    cout = w.shape[0]
    skey = ("ln_fwd", _lib.KNOB_EPOCH[0], b, cout, cin, h * wd, bool(stats), _bs(x) % 4, x.data_ptr() % 16)
    if _NOT_SERVED.get(skey):      # the library said 1000 for this shape before: no allocations, no weight split
        return None
    a3, kp = _split_weight(w, dgrad=False)
    y = torch.empty((b, cout, h, wd), dtype=torch.float32, device=x.device)
    mean = torch.empty((b, h * wd), dtype=torch.float32, device=x.device) if stats else None
    rstd = torch.empty_like(mean) if stats else None
    st = lib.pir_ln_conv1x1_fwd(x.data_ptr(), _bs(x), ln_w.data_ptr(), ln_b.data_ptr(), a3.data_ptr(), kp, y.data_ptr(),
                                _bs(y), _p(mean), _p(rstd), b, cout, cin, h * wd, _stream())
    if st == 1000:
        _NOT_SERVED[skey] = True
        return None
    check(st, "pir_ln_conv1x1_fwd")
    return (y, mean, rstd) if stats else y


def _wgrad_ln_ws(dy, x, b, cout, cin, hw) -> int:
    """Workspace floats of conv1x1_wgrad_ln: inside a deferral scope the exact need of the split-K kernel it runs."""
    if _DEFER.get(_stream(), 0) <= 0:
        return lib.pir_gemm_nt_ws_floats(cout, cin, hw, 1, b)
    g = _lib.GemmNT()
    g.X, g.x_sr, g.ldx = dy.data_ptr(), _bs(dy), hw
    g.Y, g.y_sr, g.ldy = x.data_ptr(), _bs(x), hw
    g.M1, g.M2, g.N, g.O1, g.O2, g.BR = cout, cin, hw, 1, 1, b
    need = lib.pir_gemm_nt_ws_needed(C.byref(g))
    # the LayerNorm-on-load kernel serves shorter pixel ranges than the plain "X private" plan: one slice per CU, two
    # for <= 4 row blocks (gemm_ntx.hip: xp_plan)
    return max(need, (512 if cout <= 128 else 256) * cout * cin)


def conv1x1_wgrad_ln(dy, x, mean, rstd, ln_w, ln_b, like, out=None):
    """dW = sum_b dy[b] LayerNorm(x[b])^T with the LayerNorm applied as the kernel stages x (pir_conv1x1_wgrad_ln); where
    that kernel does not serve the shape the normalised tensor is materialised first."""
    dy, x = _planes(dy), _planes(x)
    b, cout, h, wd = dy.shape
    cin = x.shape[1]
    dw = _grad_out(like, out)
    ws = workspace(_wgrad_ln_ws(dy, x, b, cout, cin, h * wd), x.device)
    st = lib.pir_conv1x1_wgrad_ln(dy.data_ptr(), _bs(dy), x.data_ptr(), _bs(x), mean.data_ptr(), rstd.data_ptr(),
                                  ln_w.data_ptr(), ln_b.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                  b, cout, cin, h * wd, _stream())
    if st == 1000:
        xn, _, _ = layernorm_forward(x, ln_w, ln_b)
        return conv1x1_wgrad(dy, xn, like, dw)
    check(st, "pir_conv1x1_wgrad_ln")
    return dw


def conv1x1_dgrad(dy: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx[b] = W^T dy[b]."""
    dy = _planes(dy)
    b, cout, h, wd = dy.shape
    cin = w.shape[1]
    hw = h * wd
    if out is None:
        out = torch.empty((b, cin, h, wd), dtype=torch.float32, device=dy.device)
    a3, kp = _split_weight(w, dgrad=True) if USE_X3 else (None, 0)
    gemm_nn(w, (0, 0), 1, cin, dy, 0, (_bs(dy), 0), hw, out, 0, (_bs(out), 0), hw, cin, cout, hw, b, 1, A3=a3, a3_kp=kp)
    return out


def _grad_out(like: torch.Tensor, out: Optional[torch.Tensor]) -> torch.Tensor:
    return out if out is not None else torch.empty_like(like, memory_format=torch.contiguous_format)


def conv1x1_wgrad(dy: torch.Tensor, x: torch.Tensor, like: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW = sum_b dy[b] x[b]^T, shaped like the weight."""
    dy, x = _planes(dy), _planes(x)
    b, cout, h, wd = dy.shape
    cin = x.shape[1]
    hw = h * wd
    dw = _grad_out(like, out)
    gemm_nt(dy, 0, (0, 0, _bs(dy)), hw, x, 0, (0, 0, _bs(x)), hw, dw, 0, (0, cin, 1), cout, cin, hw, 1, 1, b)
    return dw


def _conv3x3_ws(b: int, m: int, k: int, h: int, wd: int, device) -> Optional[torch.Tensor]:
    """Scratch for a dense 3x3 convolution the library would split over its stages (underfilled launches at the low-resolution
    levels: pir_conv3x3_x3_ws_floats > 0); a slot of its own - the slices are consumed by the reduction launched with them."""
    need = lib.pir_conv3x3_x3_ws_floats(b, m, k, h, wd)
    return workspace(need, device, slot="conv3x3") if need else None


def conv3x3_forward(x: torch.Tensor, w: torch.Tensor, residual: Optional[torch.Tensor] = None,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = _planes(x)
    b, cin, h, wd = x.shape
    cout = w.shape[0]
    if out is None:
        out = torch.empty((b, cout, h, wd), dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = _planes(residual)
    r_bs = _bs(residual) if residual is not None else 0
    if USE_X3:
        a3, kp = _split_weight(w, dgrad=False, taps=True)
        ws = _conv3x3_ws(b, cout, cin, h, wd, x.device)
        check(lib.pir_conv3x3_x3_ws(a3.data_ptr(), kp, x.data_ptr(), _bs(x), out.data_ptr(), _bs(out), _p(residual), r_bs,
                                    b, cout, cin, h, wd, _p(ws), ws.numel() if ws is not None else 0, _stream()), "pir_conv3x3_x3_ws")
    else:
        check(lib.pir_conv3x3(w.data_ptr(), 1, cin * 9, 9, 0, x.data_ptr(), _bs(x), out.data_ptr(), _bs(out),
                              _p(residual), r_bs, b, cout, cin, h, wd, _stream()), "pir_conv3x3")
    return out


def conv3x3_dgrad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    dy = _planes(dy)
    b, cout, h, wd = dy.shape
    cin = w.shape[1]
    out = torch.empty((b, cin, h, wd), dtype=torch.float32, device=dy.device)
    if USE_X3:
        a3, kp = _split_weight(w, dgrad=True, taps=True)
        ws = _conv3x3_ws(b, cin, cout, h, wd, dy.device)
        check(lib.pir_conv3x3_x3_ws(a3.data_ptr(), kp, dy.data_ptr(), _bs(dy), out.data_ptr(), _bs(out), None, 0,
                                    b, cin, cout, h, wd, _p(ws), ws.numel() if ws is not None else 0, _stream()), "pir_conv3x3_x3_ws(dgrad)")
    else:
        # A(tap, m=cin, k=cout) = w[k][m][8-tap]
        check(lib.pir_conv3x3(w.data_ptr(), 1, 9, cin * 9, 1, dy.data_ptr(), _bs(dy), out.data_ptr(), _bs(out),
                              None, 0, b, cin, cout, h, wd, _stream()), "pir_conv3x3(dgrad)")
    return out


def conv3x3_wgrad(dy: torch.Tensor, x: torch.Tensor, like: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    dy, x = _planes(dy), _planes(x)
    b, cout, h, wd = dy.shape
    cin = x.shape[1]
    dw = _grad_out(like, out)
    nws = lib.pir_conv3x3_wgrad_ws_floats(cout, cin, h, wd, b)
    ws = workspace(nws, x.device)
    check(lib.pir_conv3x3_wgrad(dy.data_ptr(), _bs(dy), x.data_ptr(), _bs(x), dw.data_ptr(), b, cout, cin, h, wd,
                                ws.data_ptr(), ws.numel(), 0, _stream()), "pir_conv3x3_wgrad")
    return dw


def layernorm_forward(x, weight, bias):
    x = _planes(x)
    b, c, h, w = x.shape
    y = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device)
    mean = torch.empty((b, h * w), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(lib.pir_layernorm_fwd(x.data_ptr(), _bs(x), weight.data_ptr(), _p(bias), y.data_ptr(), _bs(y),
                                mean.data_ptr(), rstd.data_ptr(), b, c, h * w, _stream()), "pir_layernorm_fwd")
    return y, mean, rstd


def layernorm_backward(dy, x, weight, with_bias, mean, rstd, dweight=None, dbias=None, dres=None):
    """dx (+ dres: gradient arriving over the residual connection), dweight, dbias."""
    dy, x = _planes(dy), _planes(x)
    if dres is not None:
        dres = _planes(dres)
    b, c, h, w = x.shape
    dx = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device)
    dweight = _grad_out(weight, dweight)
    dbias = _grad_out(weight, dbias) if with_bias else None
    nws = lib.pir_layernorm_bwd_ws_floats(b, c, h * w)
    ws = workspace(nws, x.device)
    check(lib.pir_layernorm_bwd(dy.data_ptr(), _bs(dy), x.data_ptr(), _bs(x), weight.data_ptr(), int(with_bias),
                                mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _bs(dx), _p(dres),
                                _bs(dres) if dres is not None else 0, dweight.data_ptr(),
                                _p(dbias), ws.data_ptr(), ws.numel(), b, c, h * w, _stream()), "pir_layernorm_bwd")
    return dx, dweight, dbias


def conv1x1_dgrad_ln_backward(dy, w, x, ln_w, mean, rstd, dweight=None, dbias=None, dres=None):
    """Input gradient of `conv1x1(LayerNorm(x))` in one pass (gemm_cst.hip): dx = LN'(W^T dy) + dres, dweight, dbias of the
    WithBias LayerNorm - or None where the fused kernel does not serve the shape (the caller then runs the pair)."""
    if not (USE_X3 and DGRAD_LN):
        return None
    dy, x = _planes(dy), _planes(x)
    if dres is not None:
        dres = _planes(dres)
    b, c, h, wd = x.shape
    skey = ("dgrad_ln", _lib.KNOB_EPOCH[0], b, c, dy.shape[1], h * wd, _bs(x) % 4, _bs(dy) % 4, x.data_ptr() % 16, dy.data_ptr() % 16,
            None if dres is None else (_bs(dres) % 4, dres.data_ptr() % 16), mean.data_ptr() % 16, rstd.data_ptr() % 16)
    if _NOT_SERVED.get(skey):
        return None
    a3, kp = _split_weight(w, dgrad=True)
    dx = torch.empty((b, c, h, wd), dtype=torch.float32, device=x.device)
    dweight, dbias = _grad_out(ln_w, dweight), _grad_out(ln_w, dbias)
    ws = workspace(max(512 * c, lib.pir_layernorm_bwd_ws_floats(b, c, h * wd)), x.device)
    st = lib.pir_conv1x1_dgrad_ln_bwd(dy.data_ptr(), _bs(dy), a3.data_ptr(), kp, dy.shape[1], x.data_ptr(), _bs(x),
                                      ln_w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(dres),
                                      _bs(dres) if dres is not None else 0, dx.data_ptr(), _bs(dx), dweight.data_ptr(),
                                      dbias.data_ptr(), ws.data_ptr(), ws.numel(), b, c, h * wd, _stream())
    if st == 1000:
        _NOT_SERVED[skey] = True
        return None
    check(st, "pir_conv1x1_dgrad_ln_bwd")
    return dx, dweight, dbias


def dwconv_forward(x, w, flip=False, out=None):
    x = _planes(x)
    b, c, h, wd = x.shape
    if out is None:
        out = torch.empty((b, c, h, wd), dtype=torch.float32, device=x.device)
    check(lib.pir_dwconv3x3(x.data_ptr(), _bs(x), w.data_ptr(), int(flip), out.data_ptr(), _bs(out),
                            b, c, h, wd, _stream()), "pir_dwconv3x3")
    return out


def dwconv_wgrad(dy, x, like, out=None):
    dy, x = _planes(dy), _planes(x)
    b, c, h, wd = x.shape
    dw = _grad_out(like, out)
    nws = lib.pir_dwconv3x3_wgrad_ws_floats(b, c, h, wd)
    ws = workspace(nws, x.device)
    check(lib.pir_dwconv3x3_wgrad(dy.data_ptr(), _bs(dy), x.data_ptr(), _bs(x), dw.data_ptr(), ws.data_ptr(),
                                  ws.numel(), b, c, h, wd, _stream()), "pir_dwconv3x3_wgrad")
    return dw


def dwconv_gate_forward(x, w):
    x = _planes(x)
    b, c2, h, wd = x.shape
    hid = c2 // 2
    g = torch.empty((b, hid, h, wd), dtype=torch.float32, device=x.device)
    check(lib.pir_dwconv3x3_gate(x.data_ptr(), _bs(x), w.data_ptr(), g.data_ptr(), _bs(g), b, hid, h, wd, _stream()),
          "pir_dwconv3x3_gate")
    return g


# no_grad forward: project_in -> dw3x3 -> gate without h0 in memory (pir_gdfn_fused_fwd).  OFF by default: the kernel moves
# 31 % fewer bytes per block but its filter / gate arithmetic (~1230 vector instructions per image row and wave beside 72
# MFMAs) makes it no faster than the pair it replaces at 96 channels and slower at batch 8 (profiles/r04_gdfn_fused_ab.txt)
GDFN_FUSED = _os.environ.get("PIR_GDFN_FUSED", "0") != "0"


def gdfn_fused_forward(x, ln_w, ln_b, win, wdw):
    """g = gelu(dw3x3(W_in LN(x))[:hid]) * dw3x3(W_in LN(x))[hid:] (net/model.py:94-97 behind :195) with the 2 hid-channel
    intermediate never written (pir_gdfn_fused_fwd); None where the kernel does not serve the shape."""
    if not (USE_X3 and GDFN_FUSED) or ln_b is None:
        return None
    x = _planes(x)
    b, c, h, w = x.shape
    hid = win.shape[0] // 2
    skey = ("gdfn_fused", _lib.KNOB_EPOCH[0], b, c, hid, h, w, _bs(x) % 4, x.data_ptr() % 16)
    if _NOT_SERVED.get(skey) or c not in (48, 96) or w not in (64, 128):
        return None
    a3, kp = _split_weight(win, dgrad=False)
    nbytes = int(lib.pir_gdfn_fused_ws_bytes(b, c, h, w))
    ws = workspace((nbytes + 3) // 4, x.device, slot="gdfn_fused")
    g = torch.empty((b, hid, h, w), dtype=torch.float32, device=x.device)
    st = lib.pir_gdfn_fused_fwd(x.data_ptr(), _bs(x), ln_w.data_ptr(), ln_b.data_ptr(), a3.data_ptr(), kp, wdw.data_ptr(),
                                g.data_ptr(), _bs(g), ws.data_ptr(), 4 * ws.numel(), None, None, b, c, hid, h, w, _stream())
    if st == 1000:
        _NOT_SERVED[skey] = True
        return None
    check(st, "pir_gdfn_fused_fwd")
    return g


def dwconv_gate_backward(x, w, dg):
    x, dg = _planes(x), _planes(dg)
    b, c2, h, wd = x.shape
    hid = c2 // 2
    dt = torch.empty((b, c2, h, wd), dtype=torch.float32, device=x.device)
    check(lib.pir_dwconv3x3_gate_bwd(x.data_ptr(), _bs(x), w.data_ptr(), dg.data_ptr(), _bs(dg), dt.data_ptr(),
                                     _bs(dt), b, hid, h, wd, _stream()), "pir_dwconv3x3_gate_bwd")
    return dt


def dwconv_backward(dy, x, w, dw_out=None):
    """dx = dwconv^T(dy) and dw in one pass (pir_dwconv3x3_bwd)."""
    dy, x = _planes(dy), _planes(x)
    b, c, h, wd = x.shape
    dx = torch.empty((b, c, h, wd), dtype=torch.float32, device=x.device)
    dw = _grad_out(w, dw_out)
    ws = workspace(lib.pir_dwconv3x3_bwd_ws_floats(b, c, h, wd), x.device)
    check(lib.pir_dwconv3x3_bwd(dy.data_ptr(), _bs(dy), x.data_ptr(), _bs(x), w.data_ptr(), dx.data_ptr(), _bs(dx),
                                dw.data_ptr(), ws.data_ptr(), ws.numel(), b, c, h, wd, _stream()), "pir_dwconv3x3_bwd")
    return dx, dw


def gdfn_dwconv_backward(x, w, dg, dw_out=None):
    """Backward of g = gelu(dw(x)[:hid]) * dw(x)[hid:] in one pass: returns (dx, dw)."""
    x, dg = _planes(x), _planes(dg)
    b, c2, h, wd = x.shape
    hid = c2 // 2
    dx = torch.empty((b, c2, h, wd), dtype=torch.float32, device=x.device)
    dw = _grad_out(w, dw_out)
    ws = workspace(lib.pir_gdfn_dwconv_bwd_ws_floats(b, hid, h, wd), x.device)
    check(lib.pir_gdfn_dwconv_bwd(x.data_ptr(), _bs(x), w.data_ptr(), dg.data_ptr(), _bs(dg), dx.data_ptr(), _bs(dx),
                                  dw.data_ptr(), ws.data_ptr(), ws.numel(), b, hid, h, wd, _stream()),
          "pir_gdfn_dwconv_bwd")
    return dx, dw


def reduce_partials(parts, stride, count_parts, out, count, alpha=1.0, accumulate=False):
    check(lib.pir_reduce_partials(parts.data_ptr(), stride, count_parts, alpha, int(accumulate), out.data_ptr(), count,
                                  _stream()), "pir_reduce_partials")


def copy_planes(src, dst, accumulate=False):
    """dst[b, :C] (a channel slice of a larger buffer is fine) = src[b, :C]."""
    src, = (_planes(src),)
    b, c, h, w = src.shape
    assert dst.shape == src.shape
    check(lib.pir_copy_planes(src.data_ptr(), _bs(src), dst.data_ptr(), _bs(dst), int(accumulate), b, c * h * w,
                              _stream()), "pir_copy_planes")


def dwconv_sumsq_forward(x, w, nsq):
    """y = dw3x3(x) and, from the same pass, partial sums of squares of the first `nsq` output channels
    (pir_dwconv3x3_sumsq): returns (y, sumsq [B, nparts, nsq])."""
    x = _planes(x)
    b, c, h, wd = x.shape
    y = torch.empty((b, c, h, wd), dtype=torch.float32, device=x.device)
    cap = int(lib.pir_dwconv3x3_sumsq_floats(b, nsq, h))
    sq = torch.empty(cap, dtype=torch.float32, device=x.device)
    nparts = C.c_int(0)
    check(lib.pir_dwconv3x3_sumsq(x.data_ptr(), _bs(x), w.data_ptr(), y.data_ptr(), _bs(y), sq.data_ptr(), cap, nsq,
                                  C.byref(nparts), b, c, h, wd, _stream()), "pir_dwconv3x3_sumsq")
    return y, sq[: b * nparts.value * nsq].view(b, nparts.value, nsq)


def mdta_attn_forward(qkv: torch.Tensor, temperature: torch.Tensor, heads: int, sumsq: Optional[torch.Tensor] = None):
    """attn = softmax(norm(q) norm(k)^T * temperature) per (image, head) (net/model.py:127-131).
    `sumsq` [B, nparts, 2C]: the squared q / k norms if the depthwise kernel already produced them."""
    b, c3, h, w = qkv.shape
    c_all = c3 // 3
    c = c_all // heads
    hw = h * w
    dev = qkv.device
    bs = _bs(qkv)
    if sumsq is None:
        sumsq = torch.empty((b, 1, 2 * c_all), dtype=torch.float32, device=dev)
        check(lib.pir_row_sumsq(qkv.data_ptr(), bs, sumsq.data_ptr(), b, 2 * c_all, hw, _stream()), "pir_row_sumsq")
    nparts = sumsq.shape[1]
    gram = torch.empty((b, heads, c, c), dtype=torch.float32, device=dev)
    attn = torch.empty_like(gram)
    if SOFTMAX_PARTS and USE_X3:     # the softmax kernel adds the split-K slices of q k^T itself: no reduction launch
        with tagged("mdta"):
            ws, splits = gemm_nt_partials(qkv, 0, (bs, c * hw, 0), hw, qkv, c_all * hw, (bs, c * hw, 0), hw, c, c, hw, b, heads, 1)
        check(lib.pir_mdta_softmax_fwd_parts(ws.data_ptr(), splits, sumsq.data_ptr(), nparts, temperature.data_ptr(),
                                             gram.data_ptr(), attn.data_ptr(), b, heads, c, _stream()), "pir_mdta_softmax_fwd_parts")
        return attn, gram, sumsq
    with tagged("mdta"):
        gemm_nt(qkv, 0, (bs, c * hw, 0), hw, qkv, c_all * hw, (bs, c * hw, 0), hw, gram, 0, (c * c, c, 1),
                c, c, hw, b, heads, 1)
    check(lib.pir_mdta_softmax_fwd(gram.data_ptr(), sumsq.data_ptr(), nparts, temperature.data_ptr(), attn.data_ptr(),
                                   b, heads, c, _stream()), "pir_mdta_softmax_fwd")
    return attn, gram, sumsq


@_tag_calls("mdta")
def mdta_attn_backward(dattn, qkv, temperature, heads, attn, gram, sumsq, dqkv, dtemp_out=None):
    """From dattn to dq, dk (written into the q / k thirds of `dqkv`) and dtemperature: backward through softmax,
    temperature and the two L2 normalisations (net/model.py:127-131).  `dattn`: the tensor, or (workspace, splits) - the
    split-K slices of dout v^T, which the softmax kernel then adds itself (gemm_nt_partials)."""
    b, c3, h, w = qkv.shape
    c_all = c3 // 3
    c = c_all // heads
    hw = h * w
    dev = qkv.device
    bs, qbs = _bs(qkv), c3 * hw
    dgram = torch.empty_like(attn)
    alpha_q = torch.empty((b, c_all), dtype=torch.float32, device=dev)
    alpha_k = torch.empty_like(alpha_q)
    dtemp_part = torch.empty((b, heads), dtype=torch.float32, device=dev)
    if isinstance(dattn, tuple):
        check(lib.pir_mdta_softmax_bwd_parts(dattn[0].data_ptr(), dattn[1], attn.data_ptr(), gram.data_ptr(), sumsq.data_ptr(),
                                             sumsq.shape[1], temperature.data_ptr(), dgram.data_ptr(), alpha_q.data_ptr(),
                                             alpha_k.data_ptr(), dtemp_part.data_ptr(), b, heads, c, _stream()),
              "pir_mdta_softmax_bwd_parts")
    else:
        check(lib.pir_mdta_softmax_bwd(dattn.data_ptr(), attn.data_ptr(), gram.data_ptr(), sumsq.data_ptr(), sumsq.shape[1],
                                       temperature.data_ptr(), dgram.data_ptr(), alpha_q.data_ptr(), alpha_k.data_ptr(),
                                       dtemp_part.data_ptr(), b, heads, c, _stream()), "pir_mdta_softmax_bwd")
    # dq = dG k + alpha_q * q and dk = dG^T q + alpha_k * k from ONE pass over q and k where the fused kernel serves the
    # shape (48 rows per head, whole 32-pixel blocks); 1000 = not served, nothing launched
    if MDTA_DQK:
        st = lib.pir_mdta_dqk(dgram.data_ptr(), qkv.data_ptr(), bs, c_all * hw, alpha_q.data_ptr(), alpha_k.data_ptr(),
                              dqkv.data_ptr(), qbs, c_all * hw, b, heads, c, hw, _stream())
        if st != 1000:
            check(st, "pir_mdta_dqk")
            dtemp = _grad_out(temperature, dtemp_out)
            reduce_partials(keep_until_flush(dtemp_part), heads, b, dtemp, heads)
            return dtemp
    # dq = dG k + alpha_q * q
    gemm_nn(dgram, (heads * c * c, c * c), c, 1, qkv, c_all * hw, (bs, c * hw), hw,
            dqkv, 0, (qbs, c * hw), hw, c, c, hw, b, heads,
            R=qkv, r_off=0, r_batch=(bs, c * hw), ldr=hw, rowscale=alpha_q, rs_batch=(c_all, c))
    # dk = dG^T q + alpha_k * k
    gemm_nn(dgram, (heads * c * c, c * c), 1, c, qkv, 0, (bs, c * hw), hw,
            dqkv, c_all * hw, (qbs, c * hw), hw, c, c, hw, b, heads,
            R=qkv, r_off=c_all * hw, r_batch=(bs, c * hw), ldr=hw, rowscale=alpha_k, rs_batch=(c_all, c))
    dtemp = _grad_out(temperature, dtemp_out)
    reduce_partials(keep_until_flush(dtemp_part), heads, b, dtemp, heads)
    return dtemp


@_tag_calls("mdta")
def mdta_core_forward(qkv: torch.Tensor, temperature: torch.Tensor, heads: int, sumsq: Optional[torch.Tensor] = None):
    """From qkv = dw3x3(1x1(x)) to out = softmax(norm(q) norm(k)^T * t) v (net/model.py:121-135)."""
    qkv = _planes(qkv)
    b, c3, h, w = qkv.shape
    c_all = c3 // 3
    c = c_all // heads
    hw = h * w
    bs = _bs(qkv)
    attn, gram, sumsq = mdta_attn_forward(qkv, temperature, heads, sumsq)
    out = torch.empty((b, c_all, h, w), dtype=torch.float32, device=qkv.device)
    gemm_nn(attn, (heads * c * c, c * c), c, 1, qkv, 2 * c_all * hw, (bs, c * hw), hw,
            out, 0, (c_all * hw, c * hw), hw, c, c, hw, b, heads)
    return out, attn, gram, sumsq


@_tag_calls("mdta")
def mdta_core_backward(dout, qkv, temperature, heads, attn, gram, sumsq, dtemp_out=None):
    dout, qkv = _planes(dout), _planes(qkv)
    b, c3, h, w = qkv.shape
    c_all = c3 // 3
    c = c_all // heads
    hw = h * w
    dev = qkv.device
    bs, dbs = _bs(qkv), _bs(dout)
    dqkv = torch.empty((b, c3, h, w), dtype=torch.float32, device=dev)
    qbs = c3 * hw
    # dA = dout v^T
    if SOFTMAX_PARTS and USE_X3:     # its split-K slices go straight to the softmax backward: no reduction launch, no dattn
        dattn = gemm_nt_partials(dout, 0, (dbs, c * hw, 0), hw, qkv, 2 * c_all * hw, (bs, c * hw, 0), hw, c, c, hw, b, heads, 1)
        dtemp = mdta_attn_backward(dattn, qkv, temperature, heads, attn, gram, sumsq, dqkv, dtemp_out)   # (reads the slices next)
        # dv = A^T dout :  A'(m=j, k=i) = attn[i*c + j]
        gemm_nn(attn, (heads * c * c, c * c), 1, c, dout, 0, (dbs, c * hw), hw,
                dqkv, 2 * c_all * hw, (qbs, c * hw), hw, c, c, hw, b, heads)
        return dqkv, dtemp
    dattn = torch.empty_like(attn)
    with immediate_reductions():     # the softmax backward reads dattn next
        gemm_nt(dout, 0, (dbs, c * hw, 0), hw, qkv, 2 * c_all * hw, (bs, c * hw, 0), hw, dattn, 0, (c * c, c, 1),
                c, c, hw, b, heads, 1)
    # dv = A^T dout :  A'(m=j, k=i) = attn[i*c + j]
    gemm_nn(attn, (heads * c * c, c * c), 1, c, dout, 0, (dbs, c * hw), hw,
            dqkv, 2 * c_all * hw, (qbs, c * hw), hw, c, c, hw, b, heads)
    dtemp = mdta_attn_backward(dattn, qkv, temperature, heads, attn, gram, sumsq, dqkv, dtemp_out)
    return dqkv, dtemp


# ---- `attn @ v` folded into project_out (net/model.py:133-137): per image
#        x1 = W_proj (blockdiag_h(attn_h) v) + x = W_eff v + x,   W_eff[:, h-block] = W_proj[:, h-block] attn_h   (C x C)
#      One C x C x HW GEMM instead of two per block in the forward, and in the backward
#        dv = W_eff^T dx1,  dW_eff = dx1 v^T   (two C x C x HW GEMMs instead of four: no project_out input / weight
#        gradient over the pixels, no dattn = dout v^T, no dv = attn^T dout), followed by C x C x c sized products
#        dW_proj[:, h-block] = sum_b dW_eff_b[:, h-block] attn_{b,h}^T,   dattn_{b,h} = W_proj[:, h-block]^T dW_eff_b[:, h-block].
#      `out` is never materialised (C planes less to write, read and keep for the backward).
@_tag_calls("mdta")
def mdta_fold_forward(qkv, attn, wproj, x_res, heads):
    b, c3, h, w = qkv.shape
    C = c3 // 3
    c = C // heads
    hw = h * w
    bs = _bs(qkv)
    weff = torch.empty((b, C, C), dtype=torch.float32, device=qkv.device)
    # W_eff[b][:, h*c:(h+1)*c] = W_proj[:, h*c:(h+1)*c] @ attn[b, h]        (M = C, K = c, N = c per (image, head))
    gemm_nn(wproj, (0, c), C, 1, attn, 0, (heads * c * c, c * c), c, weff, 0, (C * C, c), C, C, c, c, b, heads)
    x1 = torch.empty((b, C, h, w), dtype=torch.float32, device=qkv.device)
    x_res = _planes(x_res)
    gemm_nn(weff, (C * C, 0), C, 1, qkv, 2 * C * hw, (bs, 0), hw, x1, 0, (C * hw, 0), hw, C, C, hw, b, 1,
            R=x_res, r_batch=(_bs(x_res), 0), ldr=hw)
    return x1, weff


@_tag_calls("mdta")
def mdta_fold_backward(dx1, qkv, attn, weff, wproj, heads, dqkv, dwproj_out=None):
    """Writes dv into the v third of `dqkv`, returns (dattn, dW_proj)."""
    dx1 = _planes(dx1)
    b, c3, h, w = qkv.shape
    C = c3 // 3
    c = C // heads
    hw = h * w
    dev = qkv.device
    bs, dbs, qbs = _bs(qkv), _bs(dx1), c3 * hw
    # dv = W_eff^T dx1
    gemm_nn(weff, (C * C, 0), 1, C, dx1, 0, (dbs, 0), hw, dqkv, 2 * C * hw, (qbs, 0), hw, C, C, hw, b, 1)
    # dW_eff[b] = dx1[b] v[b]^T
    dweff = torch.empty((b, C, C), dtype=torch.float32, device=dev)
    with immediate_reductions():     # dW_proj and dattn below read dW_eff
        gemm_nt(dx1, 0, (dbs, 0, 0), hw, qkv, 2 * C * hw, (bs, 0, 0), hw, dweff, 0, (C * C, C, 1), C, C, hw, b, 1, 1)
    # dW_proj[:, h-block] = sum_b dW_eff[b][:, h-block] attn[b, h]^T      (contraction over j, then over the batch)
    dwproj = _grad_out(wproj, dwproj_out)
    gemm_nt(dweff, 0, (0, c, C * C), C, attn, 0, (0, c * c, heads * c * c), c, dwproj, 0, (c, C, 1), C, c, c, 1, heads, b)
    # dattn[b, h] = W_proj[:, h-block]^T dW_eff[b][:, h-block]            (M = c, K = C, N = c)
    dattn = torch.empty_like(attn)
    gemm_nn(wproj, (0, c), 1, C, dweff, 0, (C * C, c), C, dattn, 0, (heads * c * c, c * c), c, c, C, c, b, heads)
    return dattn, dwproj


def pixel_unshuffle(x):
    x = _planes(x)
    b, c, h2, w2 = x.shape
    h, w = h2 // 2, w2 // 2
    if h2 % 2 or w2 % 2:
        raise RuntimeError("pixel_unshuffle expects height and width to be divisible by 2")
    y = torch.empty((b, c * 4, h, w), dtype=torch.float32, device=x.device)
    check(lib.pir_pixel_unshuffle2(x.data_ptr(), _bs(x), y.data_ptr(), _bs(y), b, c, h, w, _stream()),
          "pir_pixel_unshuffle2")
    return y


def pixel_shuffle(x, out=None):
    x = _planes(x)
    b, c4, h, w = x.shape
    c = c4 // 4
    y = out if out is not None else torch.empty((b, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    check(lib.pir_pixel_shuffle2(x.data_ptr(), _bs(x), y.data_ptr(), _bs(y), b, c, h, w, _stream()),
          "pir_pixel_shuffle2")
    return y


# ----------------------------------------------------------------------------- autograd Functions
# Gradient sinks: when a Parameter carries `_grad_sink` (a view into the training engine's flat
# gradient buffer, promptir_amd/train.py) the weight-gradient kernel writes straight into it and the
# Function returns None for that input, so autograd launches no accumulation kernel of its own.
def _sink(p):
    return getattr(p, "_grad_sink", None) if p is not None else None


def _ret(grad, sink):
    return None if sink is not None else grad


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x, weight, bias)
        y, mean, rstd = layernorm_forward(x, weight, bias)
        ctx.with_bias = bias is not None
        ctx.sinks = (_sink(weight), _sink(bias))
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        sw, sb = ctx.sinks
        dx, dw, db = layernorm_backward(dy, x, weight, ctx.with_bias, mean, rstd, sw, sb)
        return dx, _ret(dw, sw), _ret(db, sb)


class Conv1x1Fn(torch.autograd.Function):
    """y = W x (+ residual).  W: [Cout, Cin, 1, 1]."""

    @staticmethod
    def forward(ctx, x, w, residual):
        _require_gpu(x, w, residual)
        ctx.save_for_backward(x, w)
        ctx.has_res = residual is not None
        ctx.sink = _sink(w)
        return conv1x1_forward(x, w, residual)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = conv1x1_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        dw = conv1x1_wgrad(dy, x, w, ctx.sink) if ctx.needs_input_grad[1] else None
        return dx, _ret(dw, ctx.sink), (dy if ctx.has_res and ctx.needs_input_grad[2] else None)


class Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, residual, out_slot=None):
        _require_gpu(x, w, residual)
        ctx.save_for_backward(x, w)
        ctx.has_res = residual is not None
        ctx.sink = _sink(w)
        out = out_slot.tensor((x.shape[0], w.shape[0], x.shape[2], x.shape[3])) if out_slot is not None else None
        return conv3x3_forward(x, w, residual, out)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = conv3x3_dgrad(dy, w) if ctx.needs_input_grad[0] else None
        dw = conv3x3_wgrad(dy, x, w, ctx.sink) if ctx.needs_input_grad[1] else None
        return dx, _ret(dw, ctx.sink), (dy if ctx.has_res and ctx.needs_input_grad[2] else None), None


class DwConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        _require_gpu(x, w)
        ctx.save_for_backward(x, w)
        ctx.sink = _sink(w)
        return dwconv_forward(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            dx, dw = dwconv_backward(dy, x, w, ctx.sink)     # one fused pass over dy and x
            return dx, _ret(dw, ctx.sink)
        dx = dwconv_forward(dy, w, flip=True) if ctx.needs_input_grad[0] else None
        dw = dwconv_wgrad(dy, x, w, ctx.sink) if ctx.needs_input_grad[1] else None
        return dx, _ret(dw, ctx.sink)


class BiasAddFn(torch.autograd.Function):
    """y += bias[c] in place on the fresh output of a bias-free convolution kernel (`bias=True`,
    net/model.py:88-92,111-113,206,294-320); backward: db = sum_{b,h,w} dy."""

    @staticmethod
    def forward(ctx, y, bias):
        _require_gpu(y, bias)
        if _planes(y) is not y:
            raise RuntimeError("BiasAddFn expects the plane layout of a convolution output")
        b, c, h, w = y.shape
        check(lib.pir_bias_add(y.data_ptr(), _bs(y), bias.data_ptr(), b, c, h * w, _stream()), "pir_bias_add")
        ctx.mark_dirty(y)
        ctx.sink = _sink(bias)
        ctx.c = c
        return y

    @staticmethod
    def backward(ctx, dy):
        db = None
        if ctx.needs_input_grad[1]:
            dyp = _planes(dy)
            b, c, h, w = dyp.shape
            db = ctx.sink if ctx.sink is not None else torch.empty((c,), dtype=torch.float32, device=dy.device)
            check(lib.pir_bias_grad(dyp.data_ptr(), _bs(dyp), db.data_ptr(), b, c, h * w, _stream()), "pir_bias_grad")
        return dy, _ret(db, ctx.sink)


class GeluGateFn(torch.autograd.Function):
    """g = gelu(t[:, :hid]) * t[:, hid:] (net/model.py:96-97) as its own node: the biased GDFN adds the depthwise
    bias between the stencil and the gate, so the fused DwConvGateFn does not apply there."""

    @staticmethod
    def forward(ctx, t):
        _require_gpu(t)
        t = _planes(t)
        b, c2, h, w = t.shape
        hid = c2 // 2
        g = torch.empty((b, hid, h, w), dtype=torch.float32, device=t.device)
        check(lib.pir_gelu_gate(t.data_ptr(), _bs(t), g.data_ptr(), _bs(g), b, hid, h * w, _stream()), "pir_gelu_gate")
        ctx.save_for_backward(t)
        return g

    @staticmethod
    def backward(ctx, dg):
        t, = ctx.saved_tensors
        dg = _planes(dg)
        b, c2, h, w = t.shape
        dt = torch.empty((b, c2, h, w), dtype=torch.float32, device=t.device)
        check(lib.pir_gelu_gate_bwd(t.data_ptr(), _bs(t), dg.data_ptr(), _bs(dg), dt.data_ptr(), _bs(dt), b, c2 // 2, h * w,
                                    _stream()), "pir_gelu_gate_bwd")
        return dt


class DwConvGateFn(torch.autograd.Function):
    """g = gelu(dw(x)[:hid]) * dw(x)[hid:]; the pre-gate tensor is recomputed in backward."""

    @staticmethod
    def forward(ctx, x, w):
        _require_gpu(x, w)
        ctx.save_for_backward(x, w)
        ctx.sink = _sink(w)
        return dwconv_gate_forward(x, w)

    @staticmethod
    def backward(ctx, dg):
        x, w = ctx.saved_tensors
        dx, dw = gdfn_dwconv_backward(x, w, dg, ctx.sink)    # gate-backward + dw^T + weight gradient fused
        return dx, _ret(dw, ctx.sink)


class MdtaCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, temperature, heads):
        _require_gpu(qkv, temperature)
        out, attn, gram, sumsq = mdta_core_forward(qkv, temperature, heads)
        ctx.heads = heads
        ctx.sink = _sink(temperature)
        ctx.save_for_backward(qkv, temperature, attn, gram, sumsq)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, temperature, attn, gram, sumsq = ctx.saved_tensors
        dqkv, dtemp = mdta_core_backward(dout, qkv, temperature, ctx.heads, attn, gram, sumsq, ctx.sink)
        return dqkv, _ret(dtemp, ctx.sink), None


USE_SIDE_STREAM = _os.environ.get("PIR_SIDE_STREAM", "1") != "0"   # default for callers outside a trainer
MDTA_FOLD = _os.environ.get("PIR_MDTA_FOLD", "1") != "0"             # fold attn @ v into project_out (TransformerBlockFn)
LN_FOLD = _os.environ.get("PIR_LN_FOLD", "1") != "0"                 # LayerNorm applied on load in the no_grad forward
LN_TRAIN = _os.environ.get("PIR_LN_TRAIN", "1") != "0"               # training forward: LayerNorm on load, statistics out, xn never written
DGRAD_LN = _os.environ.get("PIR_DGRAD_LN", "1") != "0"               # input gradient + LayerNorm backward in one kernel (gemm_cst.hip)
MDTA_DQK = _os.environ.get("PIR_MDTA_DQK", "1") != "0"               # dq and dk from one pass over q and k (mdta_dqk.hip)
MDTA_FOLD_MIN_HW = int(_os.environ.get("PIR_MDTA_FOLD_MIN_HW", "4096"))   # ... at the 64^2 / 128^2 levels (below, the
# C x C x c products it adds cost as much as the launch-bound GEMMs it removes: bench A/B, round 2)
_SIDE_STREAMS = {}
_SIDE_OVERRIDE = []   # innermost `side_streams(...)` scope wins over the module default
_CAPTURE_MAINS = set()   # main streams on which a side-stream block backward ran inside the CURRENT stream capture


class side_streams:
    """Scope in which the block backward may (or may not) put its weight-gradient GEMMs on a side stream.
    Per-caller state (a trainer enters it around its own forward/backward) instead of a process-wide switch."""

    def __init__(self, enabled: bool):
        self.enabled = bool(enabled)

    def __enter__(self):
        _SIDE_OVERRIDE.append(self.enabled)
        return self

    def __exit__(self, *exc):
        _SIDE_OVERRIDE.pop()
        return False


def _side_stream_enabled() -> bool:
    return _SIDE_OVERRIDE[-1] if _SIDE_OVERRIDE else USE_SIDE_STREAM


class _SideWgrads:
    """Weight-gradient GEMMs of one block backward on a second HIP stream: they only feed the optimiser, so they
    can overlap the input-gradient chain (complementary kernels share the CUs).  The operands are recorded on the
    side stream (the caching allocator must not recycle them early) and `join()` orders the side work before
    anything that follows on the main stream."""

    def __init__(self, device):
        self.main = torch.cuda.current_stream(device)
        self.side = None
        if _side_stream_enabled() and lib.records is None:   # the instrumented (timed) step stays on one stream
            # Side streams under ONE capturing stream are fine (the trainer's single-stream graph does it).  Under a
            # capture that spans SEVERAL part streams, each forking its own side stream, the round-3 build died with a
            # host segfault inside graph capture (each wgrad forks + joins a fourth-level stream, records
            # its operands on it - block frees of a capturing pool are deferred per recorded stream - and creates
            # two events per call; with two origin streams the capture held ~190 cross-stream edges per part).  The
            # fault was never bisected on hardware; the combination is refused instead of risked:
            if torch.cuda.is_current_stream_capturing():
                _CAPTURE_MAINS.add(self.main.cuda_stream)
                if len(_CAPTURE_MAINS) > 1:
                    raise RuntimeError(
                        "promptir_amd: weight-gradient side streams inside a stream capture that spans more than one "
                        "part stream are not supported (host crash in capture, DESIGN 4); wrap the step in "
                        "ops.side_streams(False) or capture on a single stream")
            else:
                _CAPTURE_MAINS.clear()
            key = (device.index, self.main.cuda_stream)   # one side stream per main stream (two-stream training)
            self.side = _SIDE_STREAMS.get(key)
            if self.side is None:
                self.side = _SIDE_STREAMS[key] = torch.cuda.Stream(device)
        self.used = False
        self.pending = []

    def wgrad(self, dy, x, like, sink, ln=None):
        """`ln` = (mean, rstd, weight, bias): x is the INPUT of the LayerNorm in front of the convolution (the normalised
        tensor was never written: ops.ln_conv1x1_forward) and is normalised as the kernel stages it."""
        dw = _grad_out(like, sink)
        run = (lambda: conv1x1_wgrad(dy, x, like, dw)) if ln is None else (lambda: conv1x1_wgrad_ln(dy, x, *ln, like, dw))
        if self.side is None:
            # low-resolution levels: the block's weight gradients wait for `join()` and go out as ONE grouped launch
            # (inside a deferral scope only: each problem then keeps a workspace piece of its own until the flush)
            if (ln is None and NT_GROUP and USE_X3 and dy.shape[2] * dy.shape[3] <= NT_GROUP_MAX_HW
                    and _DEFER.get(_stream(), 0) > 0):
                self.pending.append((dy, x, dw))
                return dw
            return run()
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        dy.record_stream(self.side)
        x.record_stream(self.side)
        with torch.cuda.stream(self.side):
            run()
        self.used = True
        return dw

    def join(self):
        while self.pending:
            chunk, self.pending = self.pending[:4], self.pending[4:]
            if len(chunk) == 1:
                conv1x1_wgrad(chunk[0][0], chunk[0][1], chunk[0][2], chunk[0][2])
            else:
                conv1x1_wgrad_group(chunk)
        if self.used:
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.main.wait_event(ev)


class TransformerBlockFn(torch.autograd.Function):
    """x -> x + attn(norm1(x)) -> (+ ffn(norm2(.))) (net/model.py:192-196) as ONE autograd node.

    Same kernels as the fine-grained Functions above; what the fusion buys is that the two residual
    gradient joins happen inside the LayerNorm-backward kernel (`dres`) instead of as eager adds issued by
    the autograd engine, and 2 instead of ~12 autograd nodes per block on the host.
    """

    @staticmethod
    def forward(ctx, x, n1w, n1b, temperature, wqkv, wdw1, wproj, n2w, n2b, win, wdw2, wout, heads, no_grad=False,
                out_slot=None):
        _require_gpu(x, n1w, n1b, temperature, wqkv, wdw1, wproj, n2w, n2b, win, wdw2, wout)
        # no_grad forward (inference, tiled restoration): nothing is saved, so the LayerNorms can be applied as the
        # consuming 1x1 convolution loads its activations (pir_ln_conv1x1_fwd) wherever that kernel serves the shape
        # (grad mode is always off INSIDE an autograd Function's forward, and needs_input_grad ignores it: the caller
        # passes `no_grad` = not torch.is_grad_enabled())
        infer = LN_FOLD and bool(no_grad)
        # training: the same kernel also leaves mean / rstd; the weight gradient normalises x on load again
        # (conv1x1_wgrad_ln), the LayerNorm backward reads x and the statistics: xn1 / xn2 are neither written nor saved
        train_fold = LN_TRAIN and not bool(no_grad)
        qkv0 = ln_conv1x1_forward(x, n1w, n1b, wqkv) if infer else None
        xn1 = m1 = r1 = None
        if train_fold:
            got = ln_conv1x1_forward(x, n1w, n1b, wqkv, stats=True)
            if got is not None:
                qkv0, m1, r1 = got
        if qkv0 is None:
            xn1, m1, r1 = layernorm_forward(x, n1w, n1b)
            qkv0 = conv1x1_forward(xn1, wqkv)
        qkv, sumsq = dwconv_sumsq_forward(qkv0, wdw1, 2 * x.shape[1])   # q / k norms from the stencil's own pass
        attn, gram, sumsq = mdta_attn_forward(qkv, temperature, heads, sumsq)
        fold = MDTA_FOLD and x.shape[2] * x.shape[3] >= MDTA_FOLD_MIN_HW
        if fold:     # attn @ v folded into project_out: one C x C x HW GEMM, `out` never exists
            x1, out = mdta_fold_forward(qkv, attn, wproj, x, heads)     # `out` slot keeps W_eff [B, C, C]
        else:
            out = torch.empty_like(x)
            b_, c_all, hw_ = x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
            c_ = c_all // heads
            with tagged("mdta"):
                gemm_nn(attn, (heads * c_ * c_, c_ * c_), c_, 1, qkv, 2 * c_all * hw_, (_bs(qkv), c_ * hw_), hw_,
                        out, 0, (c_all * hw_, c_ * hw_), hw_, c_, c_, hw_, b_, heads)
            x1 = conv1x1_forward(out, wproj, residual=x)
        g = gdfn_fused_forward(x1, n2w, n2b, win, wdw2) if infer else None   # no_grad: h0 never exists
        if g is not None:
            return conv1x1_forward(g, wout, residual=x1, out=out_slot.tensor(x.shape) if out_slot is not None else None)
        h0 = ln_conv1x1_forward(x1, n2w, n2b, win) if infer else None
        xn2 = m2 = r2 = None
        if train_fold:
            got = ln_conv1x1_forward(x1, n2w, n2b, win, stats=True)
            if got is not None:
                h0, m2, r2 = got
        if h0 is None:
            xn2, m2, r2 = layernorm_forward(x1, n2w, n2b)
            h0 = conv1x1_forward(xn2, win)
        g = dwconv_gate_forward(h0, wdw2)
        # `out_slot`: the block's output goes straight into its half of the concat buffer it is headed for
        x2 = conv1x1_forward(g, wout, residual=x1, out=out_slot.tensor(x.shape) if out_slot is not None else None)
        if infer:
            return x2
        ctx.heads = heads
        ctx.fold = fold
        ctx.with_bias = (n1b is not None, n2b is not None)
        ctx.sinks = tuple(_sink(p) for p in (n1w, n1b, temperature, wqkv, wdw1, wproj, n2w, n2b, win, wdw2, wout))
        ctx.save_for_backward(x, n1w, temperature, wqkv, wdw1, wproj, n2w, win, wdw2, wout,
                              xn1, m1, r1, qkv0, qkv, attn, gram, sumsq, out, x1, xn2, m2, r2, h0, g, n1b, n2b)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        (x, n1w, temperature, wqkv, wdw1, wproj, n2w, win, wdw2, wout,
         xn1, m1, r1, qkv0, qkv, attn, gram, sumsq, out, x1, xn2, m2, r2, h0, g, n1b, n2b) = ctx.saved_tensors
        s_n1w, s_n1b, s_t, s_qkv, s_dw1, s_proj, s_n2w, s_n2b, s_in, s_dw2, s_out = ctx.sinks
        # The four 1x1 weight gradients feed nothing downstream in this backward: they run on a side stream
        # beside the input-gradient chain (joined before returning).
        side = _SideWgrads(dx2.device)
        # Every reduction behind a parameter gradient of this block (four split-K weight gradients, dW_proj, two
        # depthwise weight gradients, two LayerNorm dweight / dbias pairs, dtemperature) is queued and runs as ONE
        # batched launch at the end of the block instead of ~11 launches of ~5 us spread over the chain.  (Not beside
        # side streams: the queue is per stream.)
        defer = deferred_reductions(side.side is None)
        defer.__enter__()
        try:
            return TransformerBlockFn._backward_body(ctx, dx2, side)
        finally:
            defer.__exit__(None, None, None)
            if defer.enabled:
                flush_reductions()

    @staticmethod
    def _backward_body(ctx, dx2, side):
        (x, n1w, temperature, wqkv, wdw1, wproj, n2w, win, wdw2, wout,
         xn1, m1, r1, qkv0, qkv, attn, gram, sumsq, out, x1, xn2, m2, r2, h0, g, n1b, n2b) = ctx.saved_tensors
        s_n1w, s_n1b, s_t, s_qkv, s_dw1, s_proj, s_n2w, s_n2b, s_in, s_dw2, s_out = ctx.sinks
        # ---- GDFN branch
        dg = conv1x1_dgrad(dx2, wout)
        d_wout = side.wgrad(dx2, g, wout, s_out)
        dh0, d_wdw2 = gdfn_dwconv_backward(h0, wdw2, dg, s_dw2)
        del dg
        d_win = side.wgrad(dh0, xn2, win, s_in) if xn2 is not None else side.wgrad(dh0, x1, win, s_in, ln=(m2, r2, n2w, n2b))
        fused = conv1x1_dgrad_ln_backward(dh0, win, x1, n2w, m2, r2, s_n2w, s_n2b, dres=dx2) if ctx.with_bias[1] else None
        if fused is None:
            dxn2 = conv1x1_dgrad(dh0, win)
            fused = layernorm_backward(dxn2, x1, n2w, ctx.with_bias[1], m2, r2, s_n2w, s_n2b, dres=dx2)
            del dxn2
        dx1, d_n2w, d_n2b = fused
        del dh0
        # ---- MDTA branch
        if ctx.fold:
            dqkv = torch.empty_like(qkv)
            dattn, d_wproj = mdta_fold_backward(dx1, qkv, attn, out, wproj, ctx.heads, dqkv, s_proj)   # out = W_eff
            d_temp = mdta_attn_backward(dattn, qkv, temperature, ctx.heads, attn, gram, sumsq, dqkv, s_t)
        else:
            dout = conv1x1_dgrad(dx1, wproj)
            d_wproj = side.wgrad(dx1, out, wproj, s_proj)
            dqkv, d_temp = mdta_core_backward(dout, qkv, temperature, ctx.heads, attn, gram, sumsq, s_t)
            del dout
        dqkv0, d_wdw1 = dwconv_backward(dqkv, qkv0, wdw1, s_dw1)
        del dqkv
        d_wqkv = side.wgrad(dqkv0, xn1, wqkv, s_qkv) if xn1 is not None else side.wgrad(dqkv0, x, wqkv, s_qkv, ln=(m1, r1, n1w, n1b))
        fused = conv1x1_dgrad_ln_backward(dqkv0, wqkv, x, n1w, m1, r1, s_n1w, s_n1b, dres=dx1) if ctx.with_bias[0] else None
        if fused is None:
            dxn1 = conv1x1_dgrad(dqkv0, wqkv)
            fused = layernorm_backward(dxn1, x, n1w, ctx.with_bias[0], m1, r1, s_n1w, s_n1b, dres=dx1)
            del dxn1
        dx, d_n1w, d_n1b = fused
        del dqkv0
        side.join()
        return (dx, _ret(d_n1w, s_n1w), _ret(d_n1b, s_n1b) if ctx.with_bias[0] else None, _ret(d_temp, s_t),
                _ret(d_wqkv, s_qkv), _ret(d_wdw1, s_dw1), _ret(d_wproj, s_proj),
                _ret(d_n2w, s_n2w), _ret(d_n2b, s_n2b) if ctx.with_bias[1] else None,
                _ret(d_win, s_in), _ret(d_wdw2, s_dw2), _ret(d_wout, s_out), None, None, None)


class PixelUnshuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        return pixel_unshuffle(x)

    @staticmethod
    def backward(ctx, dy):
        return pixel_shuffle(dy)


class PixelShuffleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_slot=None):
        _require_gpu(x)
        out = out_slot.tensor((x.shape[0], x.shape[1] // 4, 2 * x.shape[2], 2 * x.shape[3])) if out_slot is not None else None
        return pixel_shuffle(x, out)

    @staticmethod
    def backward(ctx, dy):
        return pixel_unshuffle(dy), None


class CatChannelsFn(torch.autograd.Function):
    """torch.cat([a, b], 1) (net/model.py:341,347,353,359,365,370) as two plane copies."""

    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a, b)
        ctx.ca = a.shape[1]
        buf = cat_buffer_of(a)
        if buf is not None and CatChannelsFn.adjacent(a, b, buf):
            # both producers already wrote their halves into the registered concat buffer: the concat is the buffer
            return _alias(buf, 0, buf.shape[1])
        out = torch.empty((a.shape[0], a.shape[1] + b.shape[1], a.shape[2], a.shape[3]), dtype=torch.float32,
                          device=a.device)
        copy_planes(a, out[:, :ctx.ca])
        copy_planes(b, out[:, ctx.ca:])
        return out

    @staticmethod
    def adjacent(a, b, buf) -> bool:
        bb, ctot, h, w = buf.shape
        hw = h * w
        same = a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr()
        full = (bb, a.shape[1] + b.shape[1], a.shape[2], a.shape[3]) == tuple(buf.shape)
        strides = tuple(a.stride()) == tuple(b.stride()) == (ctot * hw, hw, w, 1) or bb == 1 and a.is_contiguous() and b.is_contiguous()
        return bool(same and full and strides and a.storage_offset() == buf.storage_offset()
                    and b.storage_offset() == buf.storage_offset() + a.shape[1] * hw)

    @staticmethod
    def backward(ctx, dy):
        return dy[:, :ctx.ca], dy[:, ctx.ca:]


class ForkFn(torch.autograd.Function):
    """A tensor that feeds two consumers (an encoder output going to both the downsampling and the decoder's concat,
    net/model.py:326-334,341-370; a decoder output going to both its prompt block and the concat): two aliases out,
    and ONE pir_add of the two gradients in the backward - otherwise the autograd engine sums them with an ATen
    elementwise kernel of its own."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, da, db):
        if da is None or db is None:
            return da if db is None else db
        # One of the two is a fresh contiguous tensor (the dense-3x3 input gradient of the downsampling, the prompt
        # block's dx), the other usually a channel slice of the concat's gradient (free batch stride): the slice is
        # accumulated into the fresh one by the plane-copy kernel, no copy, no third buffer.  The fresh tensor is an
        # intermediate of this backward with this node as its only consumer.
        # Autograd does not promise exclusive ownership of an incoming gradient: accumulate in place only into a
        # tensor that owns its storage (no view: at batch 1 a channel slice of the concat gradient is contiguous too)
        # and is not the other operand.
        def owned(t):
            return t.is_contiguous() and t._base is None

        if not owned(da) and owned(db) and da is not db:
            da, db = db, da
        if not owned(da) or da is db or da.untyped_storage().data_ptr() == db.untyped_storage().data_ptr():
            out = torch.empty(da.shape, dtype=torch.float32, device=da.device)
            copy_planes(da, out)
            da = out
        copy_planes(db, da, accumulate=True)
        return da


def fork(x: torch.Tensor):
    return ForkFn.apply(x)


class PromptGenFn(torch.autograd.Function):
    """x -> bilinear(sum_l softmax(Linear(mean(x)))_l * P_l) (net/model.py:226-232); the 3x3 conv follows."""

    @staticmethod
    def forward(ctx, x, prompt_param, lin_w, lin_b):
        _require_gpu(x, prompt_param, lin_w, lin_b)
        x = _planes(x)
        b, c, h, w = x.shape
        _, L, D, S, _ = prompt_param.shape
        dev = x.device
        emb = torch.empty((b, c), dtype=torch.float32, device=dev)
        check(lib.pir_spatial_mean(x.data_ptr(), _bs(x), emb.data_ptr(), b, c, h * w, _stream()), "pir_spatial_mean")
        mix = torch.empty((b, L), dtype=torch.float32, device=dev)
        check(lib.pir_prompt_mix_fwd(emb.data_ptr(), lin_w.data_ptr(), lin_b.data_ptr(), mix.data_ptr(), b, c, L,
                                     _stream()), "pir_prompt_mix_fwd")
        out = torch.empty((b, D, h, w), dtype=torch.float32, device=dev)
        check(lib.pir_prompt_resize_fwd(mix.data_ptr(), prompt_param.data_ptr(), out.data_ptr(), _bs(out),
                                        b, L, D, S, h, w, _stream()), "pir_prompt_resize_fwd")
        ctx.save_for_backward(emb, mix, prompt_param, lin_w)
        ctx.x_shape = (b, c, h, w)
        ctx.sinks = (_sink(prompt_param), _sink(lin_w), _sink(lin_b))
        return out

    @staticmethod
    def backward(ctx, dout):
        emb, mix, prompt_param, lin_w = ctx.saved_tensors
        dout = _planes(dout)
        b, c, h, w = ctx.x_shape
        _, L, D, S, _ = prompt_param.shape
        dev = dout.device
        sP, sW, sB = ctx.sinks
        dP = _grad_out(prompt_param, sP)
        dmix = torch.empty_like(mix)
        nws = lib.pir_prompt_resize_bwd_ws_floats(b, L, D, S, h, w)
        ws = workspace(nws, dev)
        check(lib.pir_prompt_resize_bwd(dout.data_ptr(), _bs(dout), mix.data_ptr(), prompt_param.data_ptr(),
                                        dP.data_ptr(), dmix.data_ptr(), ws.data_ptr(), ws.numel(),
                                        b, L, D, S, h, w, _stream()), "pir_prompt_resize_bwd")
        dWl = _grad_out(lin_w, sW)
        dbl = sB if sB is not None else torch.empty((L,), dtype=torch.float32, device=dev)
        dx = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
        check(lib.pir_prompt_mix_bwd(dmix.data_ptr(), mix.data_ptr(), emb.data_ptr(), lin_w.data_ptr(),
                                     dWl.data_ptr(), dbl.data_ptr(), dx.data_ptr(), _bs(dx), 0, b, c, L, h * w,
                                     _stream()), "pir_prompt_mix_bwd")
        return dx, _ret(dP, sP), _ret(dWl, sW), _ret(dbl, sB)


class L1LossFn(torch.autograd.Function):
    """nn.L1Loss() (train.py:32,43); the gradient is produced in the same pass as the loss."""

    @staticmethod
    def forward(ctx, restored, clean, weight):
        _require_gpu(restored, clean)
        restored, clean = _contig4(restored), _contig4(clean)
        loss = torch.empty((), dtype=torch.float32, device=restored.device)
        ws = workspace(1024, restored.device, slot="loss")
        check(lib.pir_l1_loss(restored.data_ptr(), clean.data_ptr(), loss.data_ptr(), None, 1.0, float(weight),
                              ws.data_ptr(), restored.numel(), _stream()), "pir_l1_loss")
        ctx.save_for_backward(restored, clean)
        ctx.weight = float(weight)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        restored, clean = ctx.saved_tensors
        grad = torch.empty_like(restored)
        if dloss.numel() != 1:
            raise RuntimeError("L1LossFn.backward: scalar upstream gradient expected")
        check(lib.pir_l1_loss_grad(restored.data_ptr(), clean.data_ptr(), dloss.data_ptr(), ctx.weight, grad.data_ptr(),
                                   restored.numel(), _stream()), "pir_l1_loss_grad")
        return grad, None, None


def l1_loss(restored: torch.Tensor, clean: torch.Tensor, weight: float = 1.0) -> torch.Tensor:
    """`weight` * nn.L1Loss()(restored, clean): the weight (a part batch's share of the whole batch) travels by value into
    both kernels, so neither the forward nor the backward needs a device scalar of its own."""
    return L1LossFn.apply(restored, clean, weight)


_ONES = {}


def unit_gradient(device) -> torch.Tensor:
    """A persistent device scalar 1.0 to seed `loss.backward(gradient=...)`: the implicit seed of `loss.backward()` is
    an ATen fill kernel per call (and a fresh allocation inside a captured graph)."""
    key = torch.device(device).index
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones((), dtype=torch.float32, device=device)
    return t


def add_(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a += b on flat fp32 buffers (pir_add)."""
    _require_gpu(a, b)
    check(lib.pir_add(a.data_ptr(), b.data_ptr(), a.data_ptr(), a.numel(), _stream()), "pir_add")
    return a


def adamw_step(param, grad, exp_avg, exp_avg_sq, step, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
               grad_scale=1.0):
    """torch.optim.AdamW defaults as used at train.py:53, applied to flat fp32 buffers in place."""
    _require_gpu(param, grad, exp_avg, exp_avg_sq)
    check(lib.pir_adamw_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                             param.numel(), lr, betas[0], betas[1], eps, weight_decay, step, grad_scale, _stream()),
          "pir_adamw_step")
