#!/bin/bash
# tools/census.sh OUTDIR [extra env...]: per-step launch census (batch 32 and batch 8) of the graph step from a kernel trace
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
for B in 32 8; do
  env "$@" timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/trace_b$B" -- python3 "$ROOT/bench.py" --batch $B --steps 5 --warmup 1 --no-cpu-baseline --no-legs --config5 0 > "$ROOT/$OUT/trace_b$B.log" 2>&1
  python3 "$ROOT/tools/step_kernels.py" "$ROOT/$OUT/trace_b$B" batch$B > "$ROOT/$OUT/step_launches_b$B.json" 2> "$ROOT/$OUT/step_launches_b$B.err"
  rm -rf "$ROOT/$OUT/trace_b$B"
done
