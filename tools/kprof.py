#!/usr/bin/env python3
"""A handful of representative kernels, each launched 3 times, for rocprofv3 --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.kbench import r  # noqa: E402

B = 32
DEV = "cuda:0"


def run(fn, n=3):
    for _ in range(n):
        fn()
    torch.cuda.synchronize()


# NN: compute-heavy (noise3 ffn_in fwd), small-K full-res (L1' proj fwd, ffn_in fwd)
x, w = r(B, 704, 16, 16), r(3744, 704, 1, 1); out = torch.empty(B, 3744, 16, 16, device=DEV)
run(lambda: ops.conv1x1_forward(x, w, None, out=out))
x, w = r(B, 96, 128, 128), r(510, 96, 1, 1); out = torch.empty(B, 510, 128, 128, device=DEV)
run(lambda: ops.conv1x1_forward(x, w, None, out=out))
x, w = r(B, 96, 128, 128), r(96, 96, 1, 1); out = torch.empty(B, 96, 128, 128, device=DEV); res = r(B, 96, 128, 128)
run(lambda: ops.conv1x1_forward(x, w, res, out=out))
# NT: L1' ffn_in wgrad, noise3 ffn_in wgrad
x, dy, w = r(B, 96, 128, 128), r(B, 510, 128, 128), r(510, 96, 1, 1); o = torch.empty_like(w)
run(lambda: ops.conv1x1_wgrad(dy, x, w, out=o))
x, dy, w = r(B, 704, 16, 16), r(B, 3744, 16, 16), r(3744, 704, 1, 1); o = torch.empty_like(w)
run(lambda: ops.conv1x1_wgrad(dy, x, w, out=o))
# stencils at L1'
x2, w2, dg = r(B, 510, 128, 128), r(510, 1, 3, 3), r(B, 255, 128, 128)
run(lambda: ops.gdfn_dwconv_backward(x2, w2, dg))
run(lambda: ops.dwconv_gate_forward(x2, w2))
x3, w3 = r(B, 288, 128, 128), r(288, 1, 3, 3); y3 = torch.empty_like(x3)
run(lambda: ops.dwconv_forward(x3, w3, out=y3))
run(lambda: ops.dwconv_backward(y3, x3, w3))
xl, wl, bl = r(B, 96, 128, 128), torch.ones(96, device=DEV), torch.zeros(96, device=DEV)
yl, mean, rstd = ops.layernorm_forward(xl, wl, bl)
run(lambda: ops.layernorm_forward(xl, wl, bl))
run(lambda: ops.layernorm_backward(yl, xl, wl, True, mean, rstd))
