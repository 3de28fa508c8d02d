#!/bin/bash
# tools/env_ab_multi.sh OUTDIR ROUNDS "ENV1" "ENV2" ...   same-box interleaved bench runs, one line per run
set -u
OUT=$1; R=$2; shift 2
mkdir -p $OUT
for r in $(seq 1 $R); do
  i=0
  for E in "$@"; do
    i=$((i+1))
    env $E timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs > $OUT/run_${i}_$r.log 2>&1
    python - "$OUT/run_${i}_$r.log" "$E" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit(0)
d = json.loads(l[-1])
c5 = d.get("config5") or {}
print("%-70s b32 %7.3f ms (median %7.3f)  b8 %7.3f ms" % (sys.argv[2], d["ms_per_step"], d["ms_per_step_median"], c5.get("ms_per_step", 0)))
PY
  done
done
