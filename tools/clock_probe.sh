#!/bin/bash
# usage: tools/clock_probe.sh: shader clock during the search kernel at several occupancies
# (GRBM_GUI_ACTIVE cycles / kernel duration), to tell power capping from a saturated unit
set -uo pipefail
out="$PWD/gpurun_out/clock_probe"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in 2 3 4 5; do
  export RSBWT_WAVE_WGS_PER_CU=$w
  d="$out/w$w"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -f csv -d "$d" -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > "$d.log" 2>&1 || { echo "failed w=$w"; continue; }
  python3 - "$d" "$w" <<'PY'
import csv, sys, collections
d, w = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(d + "/p_kernel_trace.csv")):
    if "search_wave_kernel<false, false, true, true, false" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(d + "/p_counter_collection.csv")):
    if r["Dispatch_Id"] in dur:
        agg[r["Counter_Name"]].append((float(r["Counter_Value"]), dur[r["Dispatch_Id"]]))
for k, v in sorted(agg.items()):
    c = sum(x for x, _ in v) / len(v); t = sum(y for _, y in v) / len(v)
    print(f"WGS={w} {k:18s} {c:.5g} per launch, {t/1e6:.3f} ms, {c/t:.3f} per ns")
PY
done
