# batch-8 train step (BASELINE config 5's per-GPU workload) with one and two part-batch streams
for k in 2 1 2 1; do
  PIR_MICRO_STREAMS=$k python bench.py --batch 8 --no-legs --config5 0 --no-cpu-baseline 2>/dev/null > /tmp/kb.json
  python - "$k" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/kb.json") if l.startswith("{")][-1])
print("streams", sys.argv[1], "batch8", d["value"], d["ms_per_step"], flush=True)
PY
done
