#!/usr/bin/env python3
"""Workgroup-lifetime breakdown of gemm_nn_x3_kernel under load (abtest/libtrace.so = the library with gemm_x3.hip built
with -DX3_TRACE; run with PIR_LIB=abtest/libtrace.so).  Wave 0 of every workgroup records the shader clock at entry,
after the prologue (first stage in LDS), after the k loop, after the epilogue's stores are issued and after they are
acknowledged, plus the 100 MHz real-time clock at entry and exit."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
lib = _lib.lib
lib.pir_x3_trace_set.restype = ctypes.c_int
lib.pir_x3_trace_set.argtypes = [ctypes.c_void_p]
NWG = 1 << 18
trace = torch.zeros(NWG * 16, dtype=torch.int64, device="cuda:0")

SHAPES = [(96, 510, 128, False, "fwd"), (510, 96, 128, True, "fwd"), (96, 288, 128, False, "fwd"), (96, 96, 128, True, "fwd"),
          (96, 510, 64, False, "fwd"), (192, 1020, 32, False, "fwd"), (1020, 192, 32, True, "fwd"), (384, 2042, 16, False, "fwd")]
for cin, cout, S, res, mode in SHAPES:
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    res_t = r(B, cout, S, S) if res else None
    out = torch.empty(B, cout, S, S, device="cuda:0")
    fn = lambda: ops.conv1x1_forward(x, w, res_t, out=out)
    lib.pir_x3_trace_set(None)
    t_plain = timeit(fn)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    trace.zero_()
    lib.pir_x3_trace_set(trace.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.pir_x3_trace_set(None)
    t = trace.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 0] != 0]
    d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 4] - t[:, 0]], 1).astype(np.float64)
    life = (t[:, 6] - t[:, 5]).astype(np.float64) * 10.0          # ns
    span = (t[:, 6].max() - t[:, 5].min()) * 10.0                    # ns
    clk = d[:, 4].mean() / life.mean()                               # shader cycles per ns
    print(f"M={cout} K={cin} N={S*S} B={B} R={int(res)}: {t_plain*1e6:.1f} us untraced, traced span {span/1e3:.1f} us, "
          f"{len(t)} workgroups, mean concurrency {life.sum()/span:.0f} ({life.sum()/span/256:.2f} per CU), clock {clk:.2f} GHz")
    names = ["prologue", "k loop", "epi issue", "store ack", "total"]
    for i, nm in enumerate(names):
        v = d[:, i] / clk / 1e3
        print(f"    {nm:10s} mean {v.mean():7.2f} us   p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}")
    steps = np.maximum(t[:, 11].astype(np.float64), 1.0)
    if t[:, 11].max() > 0:
        ph = t[:, 8:11].astype(np.float64) / steps[:, None]
        print(f"    per k-step of the main loop (cycles, wave 0): loads+reads+MFMA issue {ph[:, 0].mean():7.0f}   wait+split+LDS writes {ph[:, 1].mean():7.0f}"
              f"   barrier {ph[:, 2].mean():7.0f}   (sum {ph.sum(1).mean():7.0f} = {ph.sum(1).mean()/clk/1e3:.2f} us)")
    sys.stdout.flush()
