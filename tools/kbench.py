#!/usr/bin/env python3
"""Kernel micro-benchmarks through the C ABI at the shapes of a batch-32 128x128 train step.

    python tools/kbench.py [nn|nt|dw|ln|all] [--batch 32]

Prints one line per (kernel, shape): time, algorithmic TFLOP/s or GB/s.  Interleaved rounds in one
process (cdna_hip_programming.md §5.4 rule 24), median of the rounds.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, rounds=5, inner=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / inner * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def r(*shape):
    return torch.randn(*shape, device=DEV)


LEVELS = [  # (name, C, HW side, heads)
    ("L1 enc C48 128^2", 48, 128, 1), ("L1 dec C96 128^2", 96, 128, 1), ("L2 C96 64^2", 96, 64, 2),
    ("L3 C192 32^2", 192, 32, 4), ("L4 C384 16^2", 384, 16, 8), ("noise3 C704 16^2", 704, 16, 4),
]


def bench_nn(B):
    print("== gemm_nn: 1x1 conv forward / dgrad ==")
    for name, C, S, heads in LEVELS:
        hid = int(C * 2.66)
        for tag, cin, cout, res in (("qkv", C, 3 * C, False), ("proj", C, C, True), ("ffn_in", C, 2 * hid, False),
                                    ("ffn_out", hid, C, True)):
            x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
            res_t = r(B, cout, S, S) if res else None
            out = torch.empty(B, cout, S, S, device=DEV)
            t = timeit(lambda: ops.conv1x1_forward(x, w, res_t, out=out))
            fl = 2.0 * cout * cin * S * S * B
            by = 4.0 * S * S * B * (cin + cout * (2 if res else 1))
            print(f"{name:18s} fwd  {tag:8s} M={cout:4d} K={cin:4d} N={S*S:5d}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s {by/t/1e9:7.0f} GB/s")
            dy = r(B, cout, S, S)
            dx = torch.empty(B, cin, S, S, device=DEV)
            t = timeit(lambda: ops.conv1x1_dgrad(dy, w, out=dx))
            by = 4.0 * S * S * B * (cin + cout)
            print(f"{name:18s} dgrd {tag:8s} M={cin:4d} K={cout:4d} N={S*S:5d}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s {by/t/1e9:7.0f} GB/s")


def bench_nt(B):
    print("== gemm_nt: 1x1 conv weight gradients, gram ==")
    for name, C, S, heads in LEVELS:
        hid = int(C * 2.66)
        for tag, cin, cout in (("qkv", C, 3 * C), ("proj", C, C), ("ffn_in", C, 2 * hid), ("ffn_out", hid, C)):
            x, dy, w = r(B, cin, S, S), r(B, cout, S, S), r(cout, cin, 1, 1)
            out = torch.empty_like(w)
            t = timeit(lambda: ops.conv1x1_wgrad(dy, x, w, out=out))
            fl = 2.0 * cout * cin * S * S * B
            by = 4.0 * S * S * B * (cin + cout)
            print(f"{name:18s} wgrd {tag:8s} M1={cout:4d} M2={cin:4d} N={S*S*B:7d}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s {by/t/1e9:7.0f} GB/s")
        qkv, temp = r(B, 3 * C, S, S), torch.ones(heads, 1, 1, device=DEV)
        t = timeit(lambda: ops.mdta_core_forward(qkv, temp, heads))
        c = C // heads
        fl = 4.0 * c * c * S * S * B * heads
        print(f"{name:18s} mdta core fwd (sumsq+gram+softmax+attn@v) c={c}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s")


def bench_dw(B):
    print("== depthwise stencils ==")
    for name, C, S, heads in LEVELS:
        hid = int(C * 2.66)
        x, w = r(B, 3 * C, S, S), r(3 * C, 1, 3, 3)
        y = torch.empty_like(x)
        t = timeit(lambda: ops.dwconv_forward(x, w, out=y))
        print(f"{name:18s} dw fwd   C={3*C:4d}: {t*1e6:8.1f} us {8.0*x.numel()/t/1e9:7.0f} GB/s")
        t = timeit(lambda: ops.dwconv_sumsq_forward(x, w, 2 * C))
        print(f"{name:18s} dw fwd+sumsq C={3*C:4d}: {t*1e6:8.1f} us {8.0*x.numel()/t/1e9:7.0f} GB/s")
        t = timeit(lambda: ops.dwconv_wgrad(y, x, w))
        print(f"{name:18s} dw wgrad C={3*C:4d}: {t*1e6:8.1f} us {8.0*x.numel()/t/1e9:7.0f} GB/s")
        x2, w2 = r(B, 2 * hid, S, S), r(2 * hid, 1, 3, 3)
        t = timeit(lambda: ops.dwconv_gate_forward(x2, w2))
        print(f"{name:18s} gate fwd hid={hid:4d}: {t*1e6:8.1f} us {4.0*x2.numel()*1.5/t/1e9:7.0f} GB/s")
        dg = r(B, hid, S, S)
        t = timeit(lambda: ops.dwconv_gate_backward(x2, w2, dg))
        print(f"{name:18s} gate bwd hid={hid:4d}: {t*1e6:8.1f} us {4.0*x2.numel()*2.5/t/1e9:7.0f} GB/s")
        t = timeit(lambda: ops.gdfn_dwconv_backward(x2, w2, dg))
        print(f"{name:18s} gdfn fused bwd hid={hid:4d}: {t*1e6:8.1f} us {4.0*x2.numel()*2.5/t/1e9:7.0f} GB/s (x+dg read, dx written)")
        t = timeit(lambda: ops.dwconv_backward(y, x, w))
        print(f"{name:18s} dw fused bwd C={3*C:4d}: {t*1e6:8.1f} us {12.0*x.numel()/t/1e9:7.0f} GB/s (dy+x read, dx written)")


def bench_ln(B):
    from promptir_amd import _lib
    print("== LayerNorm (knob 16 = 0: four waves per 64-pixel tile in the fused backward; 1: eight) ==")
    for name, C, S, heads in LEVELS + [("noise2 C320 32^2", 320, 32, 4)]:
      for knob in (0, 1):
        _lib.lib.pir_tune_set(16, knob)
        name_k = f"{name} k{knob}"
        bench_ln_one(name_k, B, C, S)
    _lib.lib.pir_tune_set(16, -1)


def bench_ln_one(name, B, C, S):
    if True:
        x, w, b = r(B, C, S, S), torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        t = timeit(lambda: ops.layernorm_forward(x, w, b))
        print(f"{name:18s} ln fwd: {t*1e6:8.1f} us {8.0*x.numel()/t/1e9:7.0f} GB/s")
        y, mean, rstd = ops.layernorm_forward(x, w, b)
        t = timeit(lambda: ops.layernorm_backward(y, x, w, True, mean, rstd))
        print(f"{name:18s} ln bwd: {t*1e6:8.1f} us {12.0*x.numel()/t/1e9:7.0f} GB/s (3 passes algorithmic)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="all")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    for key, fn in (("nn", bench_nn), ("nt", bench_nt), ("dw", bench_dw), ("ln", bench_ln)):
        if a.which in (key, "all"):
            fn(a.batch)
