"""One-shot hardware probes kept as tests, so that a toolchain or driver update cannot silently re-break an idiom the
kernels depend on.  The probe programs live in tools/probes/ and are built by __graft_entry__.build()."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "probes", "store_soffset.bin")


def test_probe_idiom_is_the_library_idiom():
    """CPU: the probe exercises the same one-instruction add the kernels use for store offsets (wide_tiles.h)."""
    lib = open(os.path.join(ROOT, "promptir_amd", "csrc", "wide_tiles.h")).read()
    probe = open(os.path.join(ROOT, "tools", "probes", "store_soffset.hip")).read()
    pat = r'asm volatile\("v_add_u32 %0, %1, %2" : "=v"\(r\) : "s"\(row_off\), "v"\(lane_off\)\);'
    assert re.search(pat, lib) and re.search(pat, probe)
    # every 16-byte store of a persistent GEMM tile goes through it; none passes a non-zero scalar offset operand
    for src in ("gemm_res.hip", "gemm_cst.hip"):
        text = open(os.path.join(ROOT, "promptir_amd", "csrc", src)).read()
        stores = re.findall(r"raw_buffer_store_b128\(([^;]*)\);", text)
        assert stores, src
        for args in stores:
            assert args.rstrip().endswith(", 0, 0"), (src, args)      # soffset = 0, aux = 0


@pytest.mark.gpu
def test_buffer_store_offsets_on_gfx950():
    """(A) per-lane offset + wave-uniform row term added in a VGPR: stores inside num_records land, stores beyond it are
    dropped - what gemm_res.hip / gemm_cst.hip rely on to clip rows beyond M.  (B) the row term in the scalar-offset
    operand: recorded, not relied on (round 3 saw wrong results with it on stores)."""
    assert os.path.exists(PROBE), "tools/probes/store_soffset.bin missing: run __graft_entry__.build()"
    res = subprocess.run([PROBE], capture_output=True, text=True, timeout=120)
    line = res.stdout.strip().splitlines()[-1]
    out = json.loads(line)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "probe_store_soffset.json"), "w") as f:
        f.write(line + "\n")
    assert "error" not in out, out
    assert out["vgpr_sum_store_in_range_ok"] == 1 and out["vgpr_sum_store_clipped_ok"] == 1, out
    assert out["vgpr_sum_load_ok"] == 1, out
    assert res.returncode == 0
