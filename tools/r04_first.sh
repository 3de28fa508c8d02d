#!/bin/bash
# Round-4 first GPU call: tests, bench line, per-step launch census (batch 32 and batch 8), FETCH_SIZE calibration.
set -u
OUT=gpurun_out/r04a
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd "$ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/status.txt
tail -5 $OUT/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > $OUT/bench.log 2>&1; echo "bench rc $?" | tee -a $OUT/status.txt
tail -c 1500 $OUT/bench.log
cd /tmp && export TMPDIR=/tmp
for B in 32 8; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$ROOT/$OUT/trace_b$B" -- python3 "$ROOT/bench.py" --batch $B --steps 5 --warmup 1 --no-cpu-baseline --no-legs --config5 0 > "$ROOT/$OUT/trace_b$B.log" 2>&1
  echo "trace b$B rc $?" | tee -a "$ROOT/$OUT/status.txt"
  python3 "$ROOT/tools/step_kernels.py" "$ROOT/$OUT/trace_b$B" batch$B > "$ROOT/$OUT/step_launches_b$B.json" 2> "$ROOT/$OUT/step_launches_b$B.err"
  rm -rf "$ROOT/$OUT/trace_b$B"
done
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/calib_fetch" -- "$ROOT/tools/probes/fetch_calib.bin" > "$ROOT/$OUT/calib_fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/calib_write" -- "$ROOT/tools/probes/fetch_calib.bin" > "$ROOT/$OUT/calib_write.log" 2>&1
python3 "$ROOT/tools/fetch_calib_summary.py" "$ROOT/$OUT/calib_fetch" "$ROOT/$OUT/calib_write" > "$ROOT/$OUT/fetch_calib.txt" 2>&1
rm -rf "$ROOT/$OUT/calib_fetch" "$ROOT/$OUT/calib_write"
cat "$ROOT/$OUT/fetch_calib.txt"
cat "$ROOT/$OUT/status.txt"
