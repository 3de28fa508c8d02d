for spec in "16 1" "16 2" "32 1" "32 2" "8 1" "4 1"; do
  set -- $spec
  PIR_MICRO_STREAMS=$2 python bench.py --batch $1 --no-legs --config5 0 --no-cpu-baseline 2>/dev/null > /tmp/kb.json
  python - "$spec" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/kb.json") if l.startswith("{")][-1])
print("batch/streams", sys.argv[1], d["value"], d["ms_per_step"], flush=True)
PY
done
