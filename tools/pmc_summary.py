#!/usr/bin/env python3
"""Per-kernel evidence table from the passes of tools/collect_profiles.sh.

For every kernel (short name, template arguments kept): launches, mean duration (kernel trace of the serialized eager
bench), share of kernel time, MFMA-pipe busy fraction, VALU / LDS instruction-active fractions, TA busy, and HBM bytes
per launch.  Writes <out>/traffic.json (per-family bytes per launch) for bench.py's `roofline.traffic`.

Conventions (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 bf16 MFMA) summed over all
SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_ACTIVE_INST_* count quad-cycles of wave time; FETCH_SIZE /
WRITE_SIZE are KB, FETCH_SIZE reads half of a wide coalesced stream on gfx950 (the x2 column)."""
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
SIMDS = 1024


def short(name):
    k = name.replace("(anonymous namespace)::", "").replace("void ", "")
    k = re.sub(r"\(.*$", "", k)
    return k[:110]


def read_counters(sub):
    agg = {}
    for path in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            a = agg.setdefault((short(row["Kernel_Name"]), row["Counter_Name"]), [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return agg


def read_trace(sub):
    agg = {}
    for path in glob.glob(os.path.join(out, sub, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            a = agg.setdefault(short(row["Kernel_Name"]), [0, 0.0])
            a[0] += 1
            a[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    return agg


trace = read_trace("trace_eager")
cnt = {}
for sub in ("pmc_mfma", "pmc_wait", "pmc_grbm", "pmc_fetch", "pmc_write"):
    cnt.update(read_counters(sub))


def mean(k, c):
    v = cnt.get((k, c))
    return v[1] / v[0] if v else None


total_us = sum(v[1] for v in trace.values())
rows = []
for k, (n, us) in sorted(trace.items(), key=lambda kv: -kv[1][1]):
    gui = mean(k, "GRBM_GUI_ACTIVE")
    cyc = gui / 8.0 if gui else None                      # active cycles of one XCD during an average launch
    mfma = mean(k, "SQ_VALU_MFMA_BUSY_CYCLES")
    valu, lds, ta = mean(k, "SQ_ACTIVE_INST_VALU"), mean(k, "SQ_ACTIVE_INST_LDS"), mean(k, "GRBM_TA_BUSY")
    fetch, write = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
    rows.append({
        "kernel": k, "launches": n, "mean_us": us / n, "share": us / total_us,
        "mfma_busy": mfma / (SIMDS * cyc) if mfma is not None and cyc else None,
        "valu_active": 4.0 * valu / (SIMDS * cyc) if valu is not None and cyc else None,
        "lds_active": 4.0 * lds / (SIMDS * cyc) if lds is not None and cyc else None,
        "ta_busy": ta / gui if ta is not None and gui else None,
        "fetch_kb_raw": fetch, "write_kb": write,
        "hbm_gbs": ((2.0 * fetch + write) * 1024 / (us / n * 1e-6) / 1e9) if fetch is not None and write is not None else None,
    })

f = lambda v, p="%.2f": "   -" if v is None else p % v
print(f"# serialized eager bench, kernel time per step family table; total kernel time in trace {total_us / 1e3:.1f} ms")
print(f"{'kernel':110s} {'calls':>6s} {'mean_us':>9s} {'share':>6s} {'mfma':>5s} {'valu':>5s} {'lds':>5s} {'ta':>5s} {'fetchKB':>9s} {'writeKB':>9s} {'GB/s(2F+W)':>10s}")
for r in rows:
    if r["share"] < 0.0005:
        continue
    print(f"{r['kernel']:110s} {r['launches']:6d} {r['mean_us']:9.1f} {r['share'] * 100:5.1f}% {f(r['mfma_busy'])} {f(r['valu_active'])} "
          f"{f(r['lds_active'])} {f(r['ta_busy'])} {f(r['fetch_kb_raw'], '%9.0f')} {f(r['write_kb'], '%9.0f')} {f(r['hbm_gbs'], '%10.0f')}")

# family totals for bench.py: kernels behind pir_gemm_nn = gemm_nn_x3_kernel / gemm_nn_kernel without the CONV flag
def template_args(k):
    m = re.search(r"<(.*)>", k)
    return [a.strip() for a in m.group(1).split(",")] if m else []


def is_conv(k):
    """gemm_nn_x3_kernel<TM, TN, WM, WN, A_MFAST, A_PRE, CONV, BREG>: the CONV flag by POSITION (index 6; defaulted
    trailing arguments are printed by the demangler, but do not rely on it)."""
    a = template_args(k)
    return len(a) > 6 and a[6] == "true"


fam = {}
for r in rows:
    k = r["kernel"]
    # the gemm_nn family of bench.py's roofline: every kernel behind pir_gemm_nn, pir_ln_conv1x1_fwd (B-stationary kernel with
    # the LayerNorm applied on load) and pir_conv1x1_dgrad_ln_bwd (C-stationary kernel with the LayerNorm backward in its tail)
    if (k.startswith("gemm_nn_x3_kernel") and not is_conv(k)) or k.startswith("gemm_nn_res_kernel") or k.startswith("gemm_nn_bst_kernel") \
            or k.startswith("gemm_nn_cst_kernel"):
        name = "pir_gemm_nn"
    elif k.startswith("gemm_nn_x3_kernel"):
        name = "pir_conv3x3_x3"
    elif k.startswith("gemm_nt_x3_kernel") or k.startswith("gemm_nt_x3_group_kernel") or k.startswith("gemm_nt_xp_kernel") \
            or k.startswith("nt_reduce") or k.startswith("reduce_one_kernel<3>") or k.startswith("reduce_one_kernel<4>"):
        name = "pir_gemm_nt"     # split-K products and their second stages (reduce_batch.hip kinds 3, 4 = the nt forms)
    else:
        name = k.split("<")[0]
    a = fam.setdefault(name, {"launches": 0, "us": 0.0, "fetch_kb_raw": 0.0, "write_kb": 0.0, "mfma_cycles": 0.0, "cycles": 0.0})
    a["launches"] += r["launches"]
    a["us"] += r["mean_us"] * r["launches"]
    if r["fetch_kb_raw"] is not None and r["write_kb"] is not None:
        a["fetch_kb_raw"] += r["fetch_kb_raw"] * r["launches"]
        a["write_kb"] += r["write_kb"] * r["launches"]
    if r["mfma_busy"] is not None:
        a["mfma_cycles"] += r["mfma_busy"] * r["mean_us"] * r["launches"]
        a["cycles"] += r["mean_us"] * r["launches"]
summary = {}
for name, a in fam.items():
    summary[name] = {"launches_in_trace": a["launches"], "mean_us": a["us"] / a["launches"],
                     "fetch_kb_per_launch_raw": a["fetch_kb_raw"] / a["launches"], "write_kb_per_launch": a["write_kb"] / a["launches"],
                     "mfma_busy_time_weighted": a["mfma_cycles"] / a["cycles"] if a["cycles"] else None}
# consistency with the bench's own count: launches of the pir_gemm_nn family in the trace must be
# (steps + warmup + 2) x roofline.launches_per_step of the traced run (the parity-gate pass and the instrumented step
# are full steps too).  A mismatch means the classification above no longer matches the kernels: fail loudly.
log = os.path.join(out, "trace_eager.log")
if os.path.exists(log):
    line = [l for l in open(log) if l.startswith("{") and '"roofline"' in l]
    if line:
        b = json.loads(line[-1])
        per_step = b["roofline"]["launches_per_step"]
        steps_in_trace = b["steps"] + b["warmup"] + 2
        got = summary.get("pir_gemm_nn", {}).get("launches_in_trace", 0)
        if got != per_step * steps_in_trace:
            sys.exit(f"pmc_summary: pir_gemm_nn has {got} launches in the trace, bench.py counted {per_step} per step x "
                     f"{steps_in_trace} steps = {per_step * steps_in_trace}: family classification is off")
        for a in summary.values():
            a["steps_in_trace"] = steps_in_trace
# whole step: every kernel's calibrated bytes x launches over the steps in the trace
steps_in_trace = next((a["steps_in_trace"] for a in summary.values() if "steps_in_trace" in a), None)
tot_bytes = sum((2.0 * r["fetch_kb_raw"] + r["write_kb"]) * 1024 * r["launches"] for r in rows
                if r["fetch_kb_raw"] is not None and r["write_kb"] is not None)
step_bytes = tot_bytes / steps_in_trace if steps_in_trace else None
for a in summary.values():
    a["fetch_factor"] = 2.0
json.dump({"source": "tools/collect_profiles.sh + tools/pmc_summary.py (rocprofv3 --pmc, separate passes, serialized eager bench)",
           "note": "traffic = 2 x FETCH_SIZE + WRITE_SIZE; the factor 2 is calibrated for 4-, 8- and 16-byte-per-lane streaming "
                   "reads and the 64-byte-per-row stage loads of the split-K kernels alike (profiles/r04_fetch_calibration.txt: "
                   "FETCH_SIZE x 1024 / bytes = 0.500 for every pattern; WRITE_SIZE = 1.000)",
           "step_bytes": step_bytes, "steps_in_trace": steps_in_trace,
           "families": summary}, open(os.path.join(out, "traffic.json"), "w"), indent=1)
if step_bytes:
    print(f"\n# whole step: {step_bytes / 1e9:.1f} GB per step (2 x FETCH_SIZE + WRITE_SIZE over every kernel, {steps_in_trace} steps in the trace)")
print("\n# families (time-weighted)")
for name, a in sorted(summary.items(), key=lambda kv: -kv[1]["mean_us"] * kv[1]["launches_in_trace"]):
    print(f"{name:28s} launches {a['launches_in_trace']:6d} mean {a['mean_us']:8.1f} us  fetch {a['fetch_kb_per_launch_raw']:10.0f} KB raw  write {a['write_kb_per_launch']:10.0f} KB"
          f"  mfma busy {f(a['mfma_busy_time_weighted'])}")
