# kernel launches of the graph step that leave the chip under-filled: fewer than 256 workgroups and longer than 25 us
set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r03_underfilled; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs --config5 0 > $OUT/run.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, os, re
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r03_underfilled"
agg = {}
for path in glob.glob(f"{out}/tr/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        n = row["Kernel_Name"].replace("(anonymous namespace)::", "")
        m = re.search(r"(\w+(<[^>]*>)?)\(", n)
        k = m.group(1) if m else n[:60]
        wgs = (int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"])) // max(1, int(row["Workgroup_Size_X"]) * int(row["Workgroup_Size_Y"]) * int(row["Workgroup_Size_Z"]))
        dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        a = agg.setdefault((k, wgs), [0, 0.0]); a[0] += 1; a[1] += dur
rows = [(v[1], k, w, v[0]) for (k, w), v in agg.items() if w < 256 and v[1] / v[0] > 25.0]
tot = sum(v[1] for v in agg.values())
print(f"total kernel time in trace {tot/1e3:.1f} ms")
for t, k, w, n in sorted(rows, reverse=True)[:40]:
    print(f"{k[:90]:90s} wgs {w:4d} calls {n:5d} mean {t/n:7.1f} us total {t/1e3:7.2f} ms")
PY
rm -rf $OUT/tr
