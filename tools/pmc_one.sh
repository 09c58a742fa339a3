#!/bin/bash
# usage: tools/pmc_one.sh TAG "KERNEL NAME SUBSTRING" "COUNTERS..." -- bench.py args
# One rocprofv3 --pmc pass of bench.py; prints the per-launch mean of every counter for the kernels whose
# name contains the substring.  Output also in gpurun_out/pmc_one_<TAG>.txt
set -euo pipefail
tag="$1"; kname="$2"; counters="$3"; shift 3; [ "${1:-}" = "--" ] && shift
out="$PWD/gpurun_out/pmc_one_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $counters -f csv -d "$out" -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 "$@" > "$out.log" 2>&1 || { tail -5 "$out.log"; exit 1; }
python3 - "$out/p_counter_collection.csv" "$kname" <<'PY' | tee "$out.txt"
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:36s} {sum(v)/len(v):.6g}   ({len(v)} launches)")
PY
