#!/bin/bash
# counter passes over the fused GDFN forward at one shape (batch 8, 96 channels, 128 x 128): tools/gdfn_fused_pmc.sh OUTDIR
set -u
OUT=$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cat > /tmp/gf_one.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from promptir_amd import ops
ops.GDFN_FUSED = True
dev = torch.device("cuda", 0)
b, c, h, w = 8, 96, 128, 128
hid = 255
x = torch.randn(b, c, h, w, device=dev)
lw, lb = torch.ones(c, device=dev), torch.zeros(c, device=dev)
win = torch.randn(2 * hid, c, 1, 1, device=dev) * 0.1
wdw = torch.randn(2 * hid, 1, 3, 3, device=dev) * 0.1
for _ in range(6):
    ops.gdfn_fused_forward(x, lw, lb, win, wdw)
    ops.dwconv_gate_forward(ops.ln_conv1x1_forward(x, lw, lb, win), wdw)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 /tmp/gf_one.py "$ROOT" > "$ROOT/$OUT/$name.log" 2>&1; }
run p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
run p2 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
run p3 --pmc FETCH_SIZE
run p4 --pmc WRITE_SIZE
python3 - "$ROOT/$OUT" <<'PY'
import csv, glob, os, re, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
dur = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
        a = agg[(k, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for path in glob.glob(os.path.join(out, "p1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:60]
        d = dur[k]; d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for k in sorted(dur, key=lambda k: -dur[k][1]):
    if dur[k][0] < 3: continue
    print(f"{k:60s} x{dur[k][0]:3d} mean {dur[k][1] / dur[k][0]:8.1f} us")
    for (kk, c), (n, v) in sorted(agg.items()):
        if kk == k: print(f"      {c:28s} {v / n:16.0f}")
PY
rm -rf "$ROOT/$OUT"/p1 "$ROOT/$OUT"/p2 "$ROOT/$OUT"/p3 "$ROOT/$OUT"/p4
