#!/usr/bin/env python3
"""Per-shape roofline of one instrumented train step: every pir_gemm_nn / pir_gemm_nt / pir_conv1x1_dgrad_ln_bwd call
with its shape, measured time, and the time its own roofline allows (bf16x3 MFMA ceiling and HBM ceiling).  "fl" rows are
the input gradients fused with a LayerNorm backward (M = C, K = cout; bytes: dy, x, dres read, dx written), "lf" the forward
1x1 convolutions with the LayerNorm applied on load (pir_ln_conv1x1_fwd), "wl" the weight gradients that normalise x on load
(pir_conv1x1_wgrad_ln; M = cout, K = cin), "ng" the grouped weight gradients of a low-resolution block (pir_gemm_nt_group:
M = sum of the problems' rows, K = the shared column count, bat = problems in the group), "cw" the dense 3x3 weight
gradients (pir_conv3x3_wgrad).  Reductions queued inside a deferral scope are timed with the flush, not with the call."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from promptir_amd import ops  # noqa: E402
from promptir_amd.train import DataParallelTrainer  # noqa: E402

MFMA = 2500e12 / 6
HBM = 6.3e12   # achievable (MI355X_MICROARCH.md)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--top", type=int, default=40)
args = ap.parse_args()

dev = torch.device("cuda", 0)
net, sd = bench.build_model(dev)
tr = DataParallelTrainer(net, lr=2e-4)
x, t = bench.build_batch(args.batch, 128, 0, dev)
for _ in range(2):
    tr.train_step(x, t)
torch.cuda.synchronize()

shapes = []
raw = ops.lib._raw
orig_nn, orig_nt, orig_fl = raw.pir_gemm_nn, raw.pir_gemm_nt, raw.pir_conv1x1_dgrad_ln_bwd
orig_lf, orig_wl = raw.pir_ln_conv1x1_fwd, raw.pir_conv1x1_wgrad_ln
orig_ng, orig_cw = raw.pir_gemm_nt_group, raw.pir_conv3x3_wgrad


class Spy:
    def __init__(self, fn, kind):
        self.fn, self.kind = fn, kind

    def __call__(self, *a):
        if self.kind == "fl":
            k, (b, c, hw) = a[4], a[18:21]
            st = self.fn(*a)    # 1000 = shape not served (nothing launched, nothing timed): the pair of calls that follows is recorded
            if st != 1000:
                shapes.append((("fl", c, k, hw, b, 1, 1, 0), 2.0 * c * k * hw * b, 4.0 * b * hw * (k + 3 * c + 2)))
            return st
        if self.kind == "lf":
            b, m, k, hw = a[10:14]
            st = self.fn(*a)
            if st != 1000:
                shapes.append((("lf", m, k, hw, b, 0, 1, 0), 2.0 * m * k * hw * b, 4.0 * b * hw * (k + m + 2)))
            return st
        if self.kind == "wl":
            b, cout, cin, hw = a[11:15]
            st = self.fn(*a)
            if st != 1000:
                shapes.append((("wl", cout, cin, hw, b, 0, 0, 0), 2.0 * cout * cin * hw * b, 4.0 * b * hw * (cout + cin + 2)))
            return st
        if self.kind == "ng":      # grouped weight gradients of a low-resolution block: one row per group
            ps = list(a[0])[:a[1]]
            key = ("ng", sum(q.M1 for q in ps), ps[0].M2 if ps[0].M2 <= ps[0].M1 else ps[0].M1, ps[0].N, len(ps), ps[0].BR, 0, 0)
            shapes.append((key, sum(2.0 * q.M1 * q.M2 * q.N * q.BR for q in ps), sum(4.0 * q.BR * (q.M1 + q.M2) * q.N for q in ps)))
            return self.fn(*a)
        if self.kind == "cw":      # dense 3x3 weight gradient
            b, cout, cin, h, w = a[5:10]
            shapes.append((("cw", cout, cin, h * w, b, 0, 0, 0), 2.0 * 9 * cout * cin * h * w * b, 4.0 * b * h * w * (cout + cin)))
            return self.fn(*a)
        g = a[0]._obj
        if self.kind == "nn":
            key = ("nn", g.M, g.K, g.N, g.O1 * g.O2, bool(g.R), bool(g.A3), int(g.a_sm == 1))
            flops = 2.0 * g.M * g.K * g.N * g.O1 * g.O2
            wshared = g.a_s1 == 0 and g.a_s2 == 0
            byts = 4.0 * g.O1 * g.O2 * (g.K * g.N + g.M * g.N * (2 if g.R else 1)) + 4.0 * g.M * g.K * (1 if wshared else g.O1 * g.O2)
        else:
            key = ("nt", g.M1, g.M2, g.N, g.O1 * g.O2, g.BR, 0, 0)
            flops = 2.0 * g.M1 * g.M2 * g.N * g.O1 * g.O2 * g.BR
            byts = 4.0 * g.O1 * g.O2 * g.BR * (g.M1 + g.M2) * g.N
        shapes.append((key, flops, byts))
        return self.fn(*a)


raw.pir_gemm_nn, raw.pir_gemm_nt, raw.pir_conv1x1_dgrad_ln_bwd = Spy(orig_nn, "nn"), Spy(orig_nt, "nt"), Spy(orig_fl, "fl")
raw.pir_ln_conv1x1_fwd, raw.pir_conv1x1_wgrad_ln = Spy(orig_lf, "lf"), Spy(orig_wl, "wl")
raw.pir_gemm_nt_group, raw.pir_conv3x3_wgrad = Spy(orig_ng, "ng"), Spy(orig_cw, "cw")
ops.lib.start_timing()
tr.train_step(x, t)
recs = [r for r in ops.lib.stop_timing() if r[0].split("@")[0] in ("pir_gemm_nn", "pir_gemm_nt", "pir_conv1x1_dgrad_ln_bwd", "pir_ln_conv1x1_fwd",
                                                                     "pir_conv1x1_wgrad_ln", "pir_gemm_nt_group", "pir_conv3x3_wgrad")]
raw.pir_gemm_nn, raw.pir_gemm_nt, raw.pir_conv1x1_dgrad_ln_bwd = orig_nn, orig_nt, orig_fl
raw.pir_ln_conv1x1_fwd, raw.pir_conv1x1_wgrad_ln = orig_lf, orig_wl
raw.pir_gemm_nt_group, raw.pir_conv3x3_wgrad = orig_ng, orig_cw
assert len(recs) == len(shapes), (len(recs), len(shapes))
agg = {}
for (name, sec, _, _), (key, flops, byts) in zip(recs, shapes):
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += sec; a[2] += flops; a[3] += byts
tot = {"nn": [0.0, 0.0], "nt": [0.0, 0.0], "fl": [0.0, 0.0], "lf": [0.0, 0.0], "wl": [0.0, 0.0], "ng": [0.0, 0.0], "cw": [0.0, 0.0]}
print(f"{'kind':3} {'M':>5} {'K':>5} {'N':>6} {'bat':>4} R A3 mf | calls  time_us  bound_us  mfma_us  hbm_us  eff")
rows = []
for key, (calls, sec, flops, byts) in agg.items():
    tm, th = flops / MFMA, byts / HBM
    bound = max(tm, th)
    tot[key[0]][0] += sec; tot[key[0]][1] += bound
    rows.append((sec - bound, key, calls, sec, bound, tm, th))
for slack, key, calls, sec, bound, tm, th in sorted(rows, reverse=True)[: args.top]:
    print(f"{key[0]:3} {key[1]:5d} {key[2]:5d} {key[3]:6d} {key[4]:4d} {int(key[5]):1d} {int(key[6]):2d} {key[7]:2d} | {calls:5d} "
          f"{sec * 1e6:8.0f} {bound * 1e6:8.0f} {tm * 1e6:8.0f} {th * 1e6:8.0f} {bound / sec:5.2f}")
for k, (sec, bound) in tot.items():
    print(f"TOTAL {k}: measured {sec * 1e3:.2f} ms, roofline bound {bound * 1e3:.2f} ms, eff {bound / max(sec, 1e-12):.2f}")
