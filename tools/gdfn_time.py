#!/usr/bin/env python3
"""Times pir_gdfn_dwconv_bwd at the train step's shapes; A/B of the register-only wave kernel (knob 7 = 0) against
the LDS-tiled kernel (knob 7 = 1) and a sweep of its band height (knob 6), interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import _lib, ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = int(os.environ.get("B", "32"))
for name, hid, S in (("L1' hid255 128^2", 255, 128), ("L1 hid127 128^2", 127, 128), ("L2 hid255 64^2", 255, 64),
                     ("L3 hid510 32^2", 510, 32), ("L4 hid1021 16^2", 1021, 16), ("n3 hid1872 16^2", 1872, 16)):
    x, w, dg = r(B, 2 * hid, S, S), r(2 * hid, 1, 3, 3), r(B, hid, S, S)
    by = 4.0 * B * S * S * hid * 5
    res = []
    for label, off, rb in (("lds", 1, 0), ("wave", 0, 0), ("wave rb8", 0, 8), ("wave rb16", 0, 16), ("wave rb32", 0, 32),
                           ("wave rb64", 0, 64)):
        if rb > S:
            continue
        _lib.lib.pir_tune_set(7, off)
        _lib.lib.pir_tune_set(6, rb)
        t = timeit(lambda: ops.gdfn_dwconv_backward(x, w, dg))
        res.append(f"{label} {t * 1e6:7.1f} us {by / t / 1e9:5.0f} GB/s")
    _lib.lib.pir_tune_set(7, 0)
    _lib.lib.pir_tune_set(6, 0)
    print(f"B={B} {name:18s} | " + " | ".join(res), flush=True)
