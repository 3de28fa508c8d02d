#!/bin/bash
# Same-box A/B of bench.py under two environments, interleaved:  tools/env_ab.sh OUTDIR "ENV_A" "ENV_B" [rounds] [bench args...]
# (an empty environment: pass "X=0")
set -u
OUT=$1; A=$2; B=$3; R=${4:-2}; shift 4 || shift $#
mkdir -p $OUT
for r in $(seq 1 $R); do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    env $E timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-legs "$@" > $OUT/run_${tag}_$r.log 2>&1
    python - "$OUT/run_${tag}_$r.log" "$tag" "$E" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], sys.argv[3], "FAILED"); sys.exit(0)
d = json.loads(l[-1])
c5 = d.get("config5") or {}
print(sys.argv[2], sys.argv[3], "patches/s", d["value"], "ms", d["ms_per_step"], "median", d["ms_per_step_median"], "config5 ms", c5.get("ms_per_step"))
PY
  done
done
