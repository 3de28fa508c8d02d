#!/bin/bash
# bench.py (timed step + legs, no CPU baseline) under the libraries given as arguments ("-" = the product library), one line per run
for lib in "$@"; do
  if [ "$lib" = "-" ]; then envs=""; else envs="PIR_LIB=$PWD/$lib"; fi
  env $envs python bench.py --no-cpu-baseline 2>/dev/null > /tmp/kb.json
  python - "$lib" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/kb.json") if l.startswith("{")][-1])
print(sys.argv[1], "train", d["value"], d["ms_per_step"], "| config5", d["config5"]["value"], "| inference", d["inference"]["value"], d["inference"]["ms_per_batch"],
      "| tiled", d["tiled_512"]["value"], flush=True)
PY
done
