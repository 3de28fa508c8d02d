for e in "X=0" "PIR_INFER_STREAMS=4" "X=0" "PIR_INFER_STREAMS=4"; do
  env $e python bench.py --steps 3 --warmup 1 --config5 0 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > /tmp/kb.json
  python - "$e" <<'PY'
import json, sys
d = json.load(open("/tmp/kb.json"))
print(sys.argv[1], "inference", d["inference"]["value"], d["inference"]["ms_per_batch"], "tiled ms", d["tiled_512"]["value"], flush=True)
PY
done
