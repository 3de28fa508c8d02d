#!/bin/bash
# bench.py (batch 32 line + the batch-8 config-5 leg) under a list of environment settings ("NAME=VALUE", "-" = defaults)
for setting in "$@"; do
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs python bench.py --no-legs --config5 1 --no-cpu-baseline 2>/dev/null > /tmp/kb.json
  python - "$setting" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/kb.json") if l.startswith("{")][-1])
c5 = d.get("config5") or {}
print(sys.argv[1], "b32:", d["value"], d["ms_per_step"], d["ms_per_step_median"], "| b8 ms:", c5.get("ms_per_step"), c5.get("patches_per_s_per_gpu", c5.get("value")), flush=True)
PY
done
