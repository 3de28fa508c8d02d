#!/usr/bin/env python3
"""Does a consumer that walks its input in the OPPOSITE image order of its producer get the tail of that tensor from
the 256 MB Infinity Cache?  (MI355X_MICROARCH.md: "a table stays resident only while the table plus every byte loaded or
stored between two uses of the same line fits in about 256 MiB".)  A producer writes T image 0 .. B-1; an ascending
consumer starts with the lines written longest ago (evicted if |T| > cache), a descending one with the newest.

Pairs timed with HIP events (consumer only), ascending vs descending image order of the consumer, the order being nothing
but the base pointer + a NEGATIVE batch stride handed to the same kernel:
    project_in GEMM (510 x 96)  ->  depthwise 3x3 + gate          T = h0, 33 MB per image
    depthwise gate              ->  project_out GEMM (96 x 255)   T = g, 17 MB per image
Usage: python tools/mall_order_probe.py [batch=16]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
c, hid, H, W = 96, 255, 128, 128
hw = H * W
torch.manual_seed(0)
x1 = torch.randn(B, c, H, W, device=dev)
w_in = torch.randn(2 * hid, c, 1, 1, device=dev) * 0.1
w_dw = torch.randn(2 * hid, 1, 3, 3, device=dev) * 0.1
w_out = torch.randn(c, hid, 1, 1, device=dev) * 0.1
lib = _lib.lib
st = torch.cuda.current_stream().cuda_stream


def gate(h0, g, desc):
    bs_h, bs_g = 2 * hid * hw, hid * hw
    if desc:
        return lib.pir_dwconv3x3_gate(h0.data_ptr() + 4 * (B - 1) * bs_h, -bs_h, w_dw.data_ptr(), g.data_ptr() + 4 * (B - 1) * bs_g, -bs_g,
                                      B, hid, H, W, st)
    return lib.pir_dwconv3x3_gate(h0.data_ptr(), bs_h, w_dw.data_ptr(), g.data_ptr(), bs_g, B, hid, H, W, st)


def pout(g, y, desc):
    a3, kp = ops._split_weight(w_out, dgrad=False)
    gm = _lib.GemmNN()
    gm.A3, gm.a3_kp = a3.data_ptr(), kp
    gm.A, gm.a_s1, gm.a_s2, gm.a_sm, gm.a_sk = w_out.data_ptr(), 0, 0, hid, 1
    sgn = -1 if desc else 1
    off_x = 4 * (B - 1) * hid * hw if desc else 0
    off_y = 4 * (B - 1) * c * hw if desc else 0
    gm.X, gm.x_s1, gm.x_s2, gm.ldx = g.data_ptr() + off_x, sgn * hid * hw, 0, hw
    gm.Y, gm.y_s1, gm.y_s2, gm.ldy = y.data_ptr() + off_y, sgn * c * hw, 0, hw
    gm.M, gm.K, gm.N, gm.O1, gm.O2 = c, hid, hw, B, 1
    return lib.pir_gemm_nn(C.byref(gm), st)


def timed(producer, consumer, reps=8):
    out = {}
    for desc in (False, True, False, True):
        ts = []
        for _ in range(reps):
            producer()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert consumer(desc) == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        out.setdefault(desc, []).append(sorted(ts)[len(ts) // 2])
    return out


h0 = ops.conv1x1_forward(x1, w_in)
g = torch.empty(B, hid, H, W, device=dev)
y = torch.empty(B, c, H, W, device=dev)
assert gate(h0, g, False) == 0
ref = g.clone()
assert gate(h0, g, True) == 0
torch.cuda.synchronize()
print("descending order: same bits as ascending:", bool(torch.equal(ref, g)))
r1 = timed(lambda: ops.conv1x1_forward(x1, w_in, out=h0), lambda d: gate(h0, g, d))
print(f"batch {B}: h0 {h0.numel() * 4 / 1e6:.0f} MB -> gate: ascending {r1[False]} us, descending {r1[True]} us")
assert pout(g, y, False) == 0
yref = y.clone()
assert pout(g, y, True) == 0
torch.cuda.synchronize()
print("project_out descending: same bits:", bool(torch.equal(yref, y)))
r2 = timed(lambda: gate(h0, g, False), lambda d: pout(g, y, d))
print(f"batch {B}: g {g.numel() * 4 / 1e6:.0f} MB -> project_out: ascending {r2[False]} us, descending {r2[True]} us")
