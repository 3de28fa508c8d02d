# counters of the LayerNorm-on-load B-stationary kernel against the plain one (one shape of tools/ln_fwd_ab.py)
set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r03_pmc_lnfwd; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ONLY=${ONLY:-"C96 128^2 ffn_in"}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -- python3 $ROOT/tools/ln_fwd_ab.py --batch 16 --only "$ONLY" > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/sq2 -- python3 $ROOT/tools/ln_fwd_ab.py --batch 16 --only "$ONLY" > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT --output-format csv -d $OUT/sq3 -- python3 $ROOT/tools/ln_fwd_ab.py --batch 16 --only "$ONLY" > $OUT/sq3.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, os, re
def kname(n):
    m = re.search(r"(\w+(<[^>]*>)?)\(", n.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else n[:60]
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_pmc_lnfwd"
agg, dur = {}, {}
for sub in ("sq", "sq2", "sq3"):
    for path in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = kname(row["Kernel_Name"])
            a = agg.setdefault(k, {}).setdefault(row["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
for path in glob.glob(f"{out}/sq/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = kname(row["Kernel_Name"])
        dur.setdefault(k, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
for k in agg:
    if "gemm" in k or "ln_fwd" in k:
        print(k, "mean us %.1f" % (sum(dur.get(k, [0])) / max(len(dur.get(k, [0])), 1)), {c: round(v[1] / v[0]) for c, v in sorted(agg[k].items())})
PY
rm -rf $OUT/sq $OUT/sq2 $OUT/sq3
