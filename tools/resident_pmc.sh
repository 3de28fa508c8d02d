set -u
SHAPE=${SHAPE:-"510 96 128"}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r03_pmc_res; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for mode in ${MODES:-0 1}; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq_$mode -- python3 $ROOT/tools/one_gemm.py $SHAPE --res $mode --iters 6 > $OUT/sq_$mode.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d $OUT/grbm_$mode -- python3 $ROOT/tools/one_gemm.py $SHAPE --res $mode --iters 6 > $OUT/grbm_$mode.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$mode -- python3 $ROOT/tools/one_gemm.py $SHAPE --res $mode --iters 6 > $OUT/fetch_$mode.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$mode -- python3 $ROOT/tools/one_gemm.py $SHAPE --res $mode --iters 6 > $OUT/write_$mode.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/sq2_$mode -- python3 $ROOT/tools/one_gemm.py $SHAPE --res $mode --iters 6 > $OUT/sq2_$mode.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_pmc_res"
for mode in (0, 1):
    agg = {}
    for sub in ("sq", "grbm", "fetch", "write", "sq2"):
        for path in glob.glob(f"{out}/{sub}_{mode}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                k = row["Kernel_Name"]
                if "gemm_nn" not in k: continue
                a = agg.setdefault(row["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
    dur = []
    for path in glob.glob(f"{out}/sq_{mode}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "gemm_nn" in row["Kernel_Name"]: dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    print("mode", mode, "mean us", sum(dur) / max(len(dur), 1), {k: round(v[1] / v[0], 1) for k, v in sorted(agg.items())})
PY
rm -rf $OUT/sq_* $OUT/grbm_* $OUT/fetch_* $OUT/write_* $OUT/sq2_*
