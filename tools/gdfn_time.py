#!/usr/bin/env python3
"""Times pir_gdfn_dwconv_bwd at the train step's shapes (batch 32); build selected by PIR_LIB."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from promptir_amd import ops  # noqa: E402
from tools.kbench import r, timeit  # noqa: E402

B = 32
tag = os.environ.get("PIR_LIB", "default").split("/")[-1]
for name, hid, S in (("L1' hid255 128^2", 255, 128), ("L1 hid127 128^2", 127, 128), ("L2 hid255 64^2", 255, 64),
                     ("L3 hid510 32^2", 510, 32), ("L4 hid1021 16^2", 1021, 16)):
    x, w, dg = r(B, 2 * hid, S, S), r(2 * hid, 1, 3, 3), r(B, hid, S, S)
    t = timeit(lambda: ops.gdfn_dwconv_backward(x, w, dg))
    by = 4.0 * B * S * S * hid * 5
    print(f"{tag:12s} {name:18s} {t * 1e6:8.1f} us  {by / t / 1e9:7.0f} GB/s", flush=True)
