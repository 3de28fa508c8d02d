#!/bin/bash
# Evidence for one build, collected on the GPU box (run through gpurun): rocprofv3 kernel-trace statistics of the bench
# (serialized eager step and the default graph step) and counter passes (one --pmc set per run, kernel trace only) of
# the serialized eager step.  Summaries land in $OUT (gpurun_out/...); copy what is to be judged into profiles/.
#   tools/collect_profiles.sh gpurun_out/r02_prof          (BATCH=8 for BASELINE config 5's per-GPU batch)
set -u
OUT=${1:-gpurun_out/prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
EAGER="PIR_GRAPH=0 PIR_SIDE_STREAM=0 PIR_MICRO_STREAMS=1"
run() {  # name, env string, rocprof args..., -- bench args
  local name=$1 envs=$2; shift 2
  env $envs rocprofv3 "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/bench.py" --batch "${BATCH:-32}" --steps "${STEPS:-3}" --warmup 1 --no-cpu-baseline --no-legs --config5 0 \
      > "$ROOT/$OUT/$name.log" 2>&1 || echo "$name failed" >> "$ROOT/$OUT/errors.txt"
}
STEPS=5 run trace_eager "$EAGER" --kernel-trace --stats
STEPS=5 run trace_graph "PIR_NOOP=1" --kernel-trace --stats
STEPS=1 run pmc_mfma "$EAGER" --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES
STEPS=1 run pmc_wait "$EAGER" --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
STEPS=1 run pmc_grbm "$EAGER" --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY
STEPS=1 run pmc_fetch "$EAGER" --kernel-trace --pmc FETCH_SIZE
STEPS=1 run pmc_write "$EAGER" --kernel-trace --pmc WRITE_SIZE
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2> "$OUT/summary.err"
python3 tools/step_roofline.py --batch "${BATCH:-32}" --top 80 > "$OUT/gemm_shape_roofline.txt" 2> "$OUT/gemm_shape_roofline.err"
# keep the per-kernel statistics, drop the raw per-dispatch CSVs (tens of MB: gpurun merges at most 64 MiB back)
for t in trace_eager trace_graph; do
  f=$(find "$OUT/$t" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${t}_kernel_stats.csv"
done
rm -rf "$OUT"/trace_eager "$OUT"/trace_graph "$OUT"/pmc_mfma "$OUT"/pmc_wait "$OUT"/pmc_grbm "$OUT"/pmc_fetch "$OUT"/pmc_write
ls -la "$OUT"
