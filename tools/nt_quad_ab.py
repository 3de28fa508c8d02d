#!/usr/bin/env python3
"""gemm_nt_x3: four-lanes-per-row stage loads (knob 14 = 1) vs fragment loads (0): equality and time on the 1x1
weight gradients and gram products of every level."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import LEVELS, r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
T = _lib.lib.pir_tune_set
tot = {0: 0.0, 1: 0.0}
for name, C, S, heads in LEVELS:
    hid = int(C * 2.66)
    for tag, cin, cout in (("qkv", C, 3 * C), ("proj", C, C), ("ffn_in", C, 2 * hid), ("ffn_out", hid, C)):
        x, dy, w = r(B, cin, S, S), r(B, cout, S, S), r(cout, cin, 1, 1)
        out = torch.empty_like(w)
        fn = lambda: ops.conv1x1_wgrad(dy, x, w, out=out)
        T(14, 0); t0 = timeit(fn); ref = out.clone()
        T(14, 1); out.zero_(); t1 = timeit(fn)
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        tot[0] += t0; tot[1] += t1
        print(f"{name:18s} wgrad {tag:8s} M1={cout:4d} M2={cin:4d} N={S*S*B:7d}: frag {t0*1e6:7.1f}  quad {t1*1e6:7.1f}  ({t1/t0:.3f})  rel diff {err:.1e}", flush=True)
    qkv, temp = r(B, 3 * C, S, S), torch.ones(heads, 1, 1, device="cuda:0")
    fn = lambda: ops.mdta_core_forward(qkv, temp, heads)
    T(14, 0); t0 = timeit(fn)
    T(14, 1); t1 = timeit(fn)
    print(f"{name:18s} mdta core fwd: frag {t0*1e6:7.1f}  quad {t1*1e6:7.1f}  ({t1/t0:.3f})", flush=True)
print({k: round(v * 1e3, 3) for k, v in tot.items()})
