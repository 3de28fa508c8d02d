#!/usr/bin/env python3
"""Experiment harness: a few dominant gemm_nn / gemm_nt shapes, timed through the C ABI (select the build with PIR_LIB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.kbench import r, timeit, DEV  # noqa: E402

B = 32
tag = os.environ.get("PIR_LIB", "default").split("/")[-1]
for name, cin, cout, S in (("ffn_in L1'", 96, 510, 128), ("qkv L1'", 96, 288, 128), ("ffn_in L2", 96, 510, 64),
                           ("ffn_in L3", 192, 1020, 32), ("ffn_in L4", 384, 2042, 16)):
    x, w = r(B, cin, S, S), r(cout, cin, 1, 1)
    out = torch.empty(B, cout, S, S, device=DEV)
    t = timeit(lambda: ops.conv1x1_forward(x, w, None, out=out))
    dy, dx = r(B, cout, S, S), torch.empty(B, cin, S, S, device=DEV)
    t2 = timeit(lambda: ops.conv1x1_dgrad(dy, w, out=dx))
    ow = torch.empty_like(w)
    t3 = timeit(lambda: ops.conv1x1_wgrad(dy, x, w, out=ow))
    print(f"{tag:14s} {name:12s} fwd {t*1e6:8.1f} us  dgrad {t2*1e6:8.1f} us  wgrad {t3*1e6:8.1f} us", flush=True)
