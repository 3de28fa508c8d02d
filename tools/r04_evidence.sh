#!/bin/bash
# Round-4 evidence set from ONE box: bench line, kernel stats (eager + graph), counter passes, per-shape roofline, launch census.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r04_prof
mkdir -p "$ROOT/$OUT"
cd "$ROOT"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2>&1; echo "bench rc $?"
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench_line.json
bash tools/collect_profiles.sh $OUT > $OUT/collect.log 2>&1; echo "collect rc $?"
bash tools/census.sh $OUT/census X=0 > $OUT/census.log 2>&1; echo "census rc $?"
ls -la $OUT | head -40
tail -5 $OUT/summary.txt
