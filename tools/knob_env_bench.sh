#!/bin/bash
# bench.py under a list of environment settings ("NAME=VALUE" words, one run each; "-" = defaults), one line per run
for setting in "$@"; do
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs python bench.py --no-legs --config5 0 --no-cpu-baseline 2>/dev/null > /tmp/kb.json
  python - "$setting" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/kb.json") if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["ms_per_step_median"], flush=True)
PY
done
