# FETCH_SIZE / duration / MFMA busy of the grouped low-resolution weight gradients under several knob settings.
set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/group_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
IFS=';' read -ra SH <<< "${SHAPES:-192 32;384 16}"
IFS=';' read -ra KN <<< "${KNOBS:-34=0;34=1}"
rm -f $OUT/index.txt
si=0
for shape in "${SH[@]}"; do
  ki=0
  for kn in "${KN[@]}"; do
    tag=s${si}_k${ki}
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$tag -- python3 $ROOT/tools/one_group.py $shape --knob $kn > $OUT/fetch_$tag.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq_$tag -- python3 $ROOT/tools/one_group.py $shape --knob $kn > $OUT/sq_$tag.log 2>&1
    tail -1 $OUT/fetch_$tag.log
    echo "$tag|$shape|$kn" >> $OUT/index.txt
    ki=$((ki+1))
  done
  si=$((si+1))
done
cd $ROOT
python3 - <<'PY'
import csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/group_pmc"
for ent in open(out + "/index.txt"):
    tag, shape, kn = ent.strip().split("|")
    agg, dur = {}, {}
    for sub in ("fetch", "sq"):
        for path in glob.glob(f"{out}/{sub}_{tag}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                k = row["Kernel_Name"]
                if "gemm_nt" not in k: continue
                a = agg.setdefault((k[:80], row["Counter_Name"]), [0, 0.0]); a[0] += 1; a[1] += float(row["Counter_Value"])
    for path in glob.glob(f"{out}/fetch_{tag}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"]
            if "gemm_nt" in k: dur.setdefault(k[:80], []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    for k, d in dur.items():
        c = {cn: v[1] / v[0] for (kk, cn), v in agg.items() if kk == k}
        f = c.get("FETCH_SIZE", 0) * 1024
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1)
        wt = c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
        print(f"{shape:8s} {kn:10s} {k[28:80]:52s} us {sorted(d)[len(d)//2]:8.1f}  FETCHx2 MB {2*f/1e6:8.1f}  mfma(raw ratio) {mf:.2f} wait {wt:.2f}")
PY
rm -rf $OUT/fetch_* $OUT/sq_*
