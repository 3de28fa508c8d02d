python bench.py --batch 8 --no-legs --config5 0 --no-cpu-baseline > gpurun_out/r03_b8_a.json 2>/dev/null
PIR_MICRO_STREAMS=1 python bench.py --batch 8 --no-legs --config5 0 --no-cpu-baseline > gpurun_out/r03_b8_b.json 2>/dev/null
python - <<PY
import json
for f in ("a","b"):
    d=json.loads([l for l in open(f"gpurun_out/r03_b8_{f}.json") if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["config"]["execution"])
    fam=d["roofline"]["families_ms"]
    print({k:v for k,v in fam.items()}, sum(fam.values()))
    print("launches nn", d["roofline"]["launches_per_step"])
PY
