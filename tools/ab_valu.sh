#!/bin/bash
# usage: tools/ab_valu.sh TAG LIB...   (on the GPU box)
# The A/B VERDICT r04 (next #2) asks for: the headline launch of several builds of librsbwt.so on ONE box, each
# (1) un-profiled: python3 bench.py (20 timed steps; the library's own HIP events around the search kernel), then
# (2) one rocprofv3 --pmc pass with SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE.
# Writes gpurun_out/ab_valu_<TAG>.jsonl: one line per build.
set -uo pipefail
tag="$1"; shift
out="$PWD/gpurun_out/ab_valu_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
ARGS="--cpu-sample 0 --no-single-check --no-second-mix ${BENCH_ARGS:-}"
: > "$out.jsonl"
for lib in "$@"; do
  name="$(basename "$lib" .so)"
  export RSBWT_LIB="$lib"
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 $ARGS > "$out/$name.bench.json" 2> "$out/$name.bench.err" || { echo "bench FAILED: $lib"; tail -3 "$out/$name.bench.err"; continue; }
  d="$out/$name.pmc"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -f csv -d "$d" -o p -- python3 bench.py --steps 3 --warmup 1 $ARGS > "$d.log" 2>&1 || { echo "pmc FAILED: $lib"; tail -3 "$d.log"; }
  python3 - "$out/$name.bench.json" "$d/p_counter_collection.csv" "$name" >> "$out.jsonl" <<'PY'
import csv, sys, json, collections, os
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
b = json.loads(line)
r = b["roofline"]
rec = {"build": sys.argv[3], "kernel_ms": r["kernel_ms"], "ms_per_step": b["ms_per_step"], "frac": r["frac"], "value": b["value"],
       "passes": r.get("phase_stamps", {}).get("passes"), "cycles_per_pass": r.get("phase_stamps", {}).get("cycles_per_pass")}
if os.path.exists(sys.argv[2]):
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(sys.argv[2])):
        if "search_lines_kernel<false, false, false, 0>" in row["Kernel_Name"] or "search_solo_kernel<false, false, false, false, false, false>" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    rec["pmc_mean_per_launch"] = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
    p = rec["pmc_mean_per_launch"]
    if "SQ_INSTS_VALU" in p and "GRBM_GUI_ACTIVE" in p:
        # wave-instructions x 4 cycles over the 1,024 SIMDs' cycles (GRBM_GUI_ACTIVE sums the 8 XCDs)
        rec["valu_busy_if_4_cycles_each"] = p["SQ_INSTS_VALU"] * 4.0 / (1024.0 * p["GRBM_GUI_ACTIVE"] / 8.0)
print(json.dumps(rec))
PY
  tail -1 "$out.jsonl"
done
