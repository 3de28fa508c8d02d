# FETCH_SIZE / WRITE_SIZE / duration of the weight-gradient kernels of given shapes, for several knob settings.
#   SHAPES="510 96 128 --ln;288 96 128 --ln" KNOBS="38=0;38=1" bash tools/wgrad_pmc.sh
set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/wgrad_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
IFS=';' read -ra SH <<< "${SHAPES:-510 96 128 --ln;288 96 128 --ln;255 96 128;254 48 128 --ln}"
IFS=';' read -ra KN <<< "${KNOBS:-38=0;38=1}"
si=0
for shape in "${SH[@]}"; do
  ki=0
  for kn in "${KN[@]}"; do
    tag=s${si}_k${ki}
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$tag -- python3 $ROOT/tools/one_wgrad.py $shape --knob $kn > $OUT/fetch_$tag.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$tag -- python3 $ROOT/tools/one_wgrad.py $shape --knob $kn > $OUT/write_$tag.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/sq_$tag -- python3 $ROOT/tools/one_wgrad.py $shape --knob $kn > $OUT/sq_$tag.log 2>&1
    echo "$tag|$shape|$kn" >> $OUT/index.txt
    ki=$((ki+1))
  done
  si=$((si+1))
done
cd $ROOT
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/wgrad_pmc"
lines = []
for ent in open(out + "/index.txt"):
    tag, shape, kn = ent.strip().split("|")
    agg, dur = {}, {}
    for sub in ("fetch", "write", "sq"):
        for path in glob.glob(f"{out}/{sub}_{tag}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                k = row["Kernel_Name"]
                if "gemm_nt" not in k: continue
                a = agg.setdefault((k[:70], row["Counter_Name"]), [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
    for path in glob.glob(f"{out}/fetch_{tag}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if "gemm_nt" in k: dur.setdefault(k[:70], []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    for k, d in dur.items():
        c = {cn: v[1] / v[0] for (kk, cn), v in agg.items() if kk == k}
        f, w = c.get("FETCH_SIZE", 0) * 1024, c.get("WRITE_SIZE", 0) * 1024
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1) / 4
        lines.append(f"{shape:22s} {kn:8s} {k[28:70]:42s} us {sum(d)/len(d):8.1f}  FETCHx2 MB {2*f/1e6:8.1f}  WRITE MB {w/1e6:7.1f}  mfma {mf:.2f}")
print("\n".join(lines))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
PY
rm -rf $OUT/fetch_* $OUT/write_* $OUT/sq_* $OUT/index.txt
