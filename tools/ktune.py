#!/usr/bin/env python3
"""Sweeps the gemm_nn / gemm_nt tile configurations (pir_tune_set) over the train step's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.kbench import LEVELS, r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
lib = ops.lib


def sweep_nn():
    names = ["32x256", "64x256", "96x256", "128x128", "64x128"]
    print("== gemm_nn tile sweep (us); auto first ==")
    for name, C, S, heads in LEVELS:
        hid = int(C * 2.66)
        for tag, cin, cout in (("qkv", C, 3 * C), ("proj", C, C), ("ffn_in", C, 2 * hid), ("ffn_out", hid, C)):
            for mode in ("fwd", "dgrd"):
                if mode == "fwd":
                    x, w, out = r(B, cin, S, S), r(cout, cin, 1, 1), torch.empty(B, cout, S, S, device="cuda:0")
                    fn = lambda: ops.conv1x1_forward(x, w, None, out=out)
                    M = cout
                else:
                    x, w, out = r(B, cout, S, S), r(cout, cin, 1, 1), torch.empty(B, cin, S, S, device="cuda:0")
                    fn = lambda: ops.conv1x1_dgrad(x, w, out=out)
                    M = cin
                lib.pir_tune_set(0, -1)
                res = [timeit(fn, rounds=3, inner=2) * 1e6]
                for cfg in range(5):
                    lib.pir_tune_set(0, cfg)
                    res.append(timeit(fn, rounds=3, inner=2) * 1e6)
                lib.pir_tune_set(0, -1)
                best = min(range(5), key=lambda i: res[1 + i])
                print(f"{name:18s} {mode:4s} {tag:8s} M={M:4d} K={(cin if mode=='fwd' else cout):4d} N={S*S:5d}: auto {res[0]:7.1f} | "
                      + " ".join(f"{v:7.1f}" for v in res[1:]) + f" | best {names[best]}")


def sweep_nt():
    names = ["64x64k4", "128x64k2", "128x96k2", "128x128"]
    print("== gemm_nt tile/split sweep (us) ==")
    for name, C, S, heads in LEVELS:
        hid = int(C * 2.66)
        for tag, cin, cout in (("qkv", C, 3 * C), ("proj", C, C), ("ffn_in", C, 2 * hid), ("ffn_out", hid, C)):
            x, dy, w = r(B, cin, S, S), r(B, cout, S, S), r(cout, cin, 1, 1)
            out = torch.empty_like(w)
            fn = lambda: ops.conv1x1_wgrad(dy, x, w, out=out)
            lib.pir_tune_set(1, -1); lib.pir_tune_set(2, 0)
            auto = timeit(fn, rounds=3, inner=2) * 1e6
            best = (1e9, None)
            row = []
            for cfg in range(4):
                for splits in (8, 16, 32, 64, 128, 256, 512):
                    lib.pir_tune_set(1, cfg); lib.pir_tune_set(2, splits)
                    t = timeit(fn, rounds=2, inner=2) * 1e6
                    row.append((t, cfg, splits))
                    if t < best[0]:
                        best = (t, (cfg, splits))
            lib.pir_tune_set(1, -1); lib.pir_tune_set(2, 0)
            top = sorted(row)[:3]
            print(f"{name:18s} wgrd {tag:8s} M1={max(cin,cout):4d} M2={min(cin,cout):4d} N={S*S*B:7d}: auto {auto:7.1f} | "
                  + " ".join(f"{names[c]}/s{s}:{t:.0f}" for t, c, s in top))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("nn", "all"):
        sweep_nn()
    if which in ("nt", "all"):
        sweep_nt()
