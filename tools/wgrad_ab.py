#!/usr/bin/env python3
"""A/B of two settings of one tuning knob on the 1x1 weight gradients of a part batch (16 images by default): plain
(pir_gemm_nt) and with the LayerNorm applied on load (pir_conv1x1_wgrad_ln).  Agreement of the two results and time per call.

    B=16 KNOB=38 python tools/wgrad_ab.py           # gemm_nt_xp_kernel: half-step refill (0) against grouped loads (1)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.resident_ab import timeit, r  # noqa: E402

B = int(os.environ.get("B", "16"))
KNOB = int(os.environ.get("KNOB", "38"))
V0, V1 = int(os.environ.get("V0", "0")), int(os.environ.get("V1", "1"))
T = _lib.lib.pir_tune_set
tot = [0.0, 0.0]
print(f"knob {KNOB}: {V0} against {V1}, batch {B}")
print(f"{'shape':40s} | {'v0 us':>9s} {'v1 us':>9s} {'ratio':>6s} | {'GB/s v1':>7s} {'TF/s v1':>7s} | rel.diff")
for C, S in ((48, 128), (96, 128), (96, 64), (192, 32), (384, 16)):
    hid = int(C * 2.66)
    for tag, cin, cout, ln in (("qkv+ln", C, 3 * C, True), ("ffn_in+ln", C, 2 * hid, True), ("qkv", C, 3 * C, False),
                               ("ffn_in", C, 2 * hid, False), ("ffn_out", hid, C, False)):
        x, dy, w = r(B, cin, S, S), r(B, cout, S, S), r(cout, cin, 1, 1)
        outs = [torch.empty_like(w), torch.empty_like(w)]
        if ln:
            gam, bet = r(cin), r(cin)
            _, mean, rstd = ops.layernorm_forward(x, gam, bet)

        def run(v, out):
            T(KNOB, v)
            if ln:
                return ops.conv1x1_wgrad_ln(dy, x, mean, rstd, gam, bet, w, out=out)
            return ops.conv1x1_wgrad(dy, x, w, out=out)

        if ln and run(V0, outs[0]) is None:
            continue                        # shape not served by the fused kernel
        f0, f1 = (lambda: run(V0, outs[0])), (lambda: run(V1, outs[1]))
        f0(); f1()
        torch.cuda.synchronize()
        err = float((outs[0] - outs[1]).abs().max()) / max(float(outs[0].abs().max()), 1e-30)
        t0, t1 = timeit([f0, f1])
        tot[0] += t0; tot[1] += t1
        by = 4.0 * S * S * B * (cin + cout)
        fl = 2.0 * cin * cout * S * S * B
        print(f"C{C} {S}^2 wgrad {tag:10s} {cout:4d}x{cin:4d} | {t0*1e6:9.1f} {t1*1e6:9.1f} {t1/t0:6.2f} | {by/t1/1e9:7.0f} {fl/t1/1e12:7.1f} | {err:.1e}",
              flush=True)
# per-image products of the MDTA block (net/model.py:129 and the adjoint of :133-137): gram = q k^T per (image, head), dW_eff = dx1 v^T
for C, S, heads in ((48, 128, 1), (96, 128, 1), (96, 64, 2), (192, 32, 4), (384, 16, 8)):
    c, hw = C // heads, S * S
    qkv, dx1 = r(B, 3 * C, S, S), r(B, C, S, S)
    bs = 3 * C * hw
    for tag in ("gram", "dW_eff"):
        outs = [torch.empty(B, heads, c, c, device="cuda:0") if tag == "gram" else torch.empty(B, C, C, device="cuda:0") for _ in range(2)]

        def run(v, out):
            T(KNOB, v)
            if tag == "gram":
                ops.gemm_nt(qkv, 0, (bs, c * hw, 0), hw, qkv, C * hw, (bs, c * hw, 0), hw, out, 0, (c * c, c, 1), c, c, hw, B, heads, 1)
            else:
                ops.gemm_nt(dx1, 0, (C * hw, 0, 0), hw, qkv, 2 * C * hw, (bs, 0, 0), hw, out, 0, (C * C, C, 1), C, C, hw, B, 1, 1)

        f0, f1 = (lambda: run(V0, outs[0])), (lambda: run(V1, outs[1]))
        f0(); f1()
        torch.cuda.synchronize()
        err = float((outs[0] - outs[1]).abs().max()) / max(float(outs[0].abs().max()), 1e-30)
        t0, t1 = timeit([f0, f1])
        tot[0] += t0; tot[1] += t1
        m = c if tag == "gram" else C
        by = 4.0 * hw * B * 2 * C
        fl = 2.0 * m * m * hw * B * (heads if tag == "gram" else 1)
        print(f"C{C} {S}^2 {tag:8s} {m:4d}x{m:4d} x{B * (heads if tag == 'gram' else 1):4d}        | {t0*1e6:9.1f} {t1*1e6:9.1f} {t1/t0:6.2f} | {by/t1/1e9:7.0f} {fl/t1/1e12:7.1f} | {err:.1e}",
              flush=True)
print("sum v0 %.3f ms, v1 %.3f ms" % (tot[0] * 1e3, tot[1] * 1e3))
