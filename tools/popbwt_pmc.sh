#!/bin/bash
# Runs on the GPU box (via gpurun): kernel stats + two --pmc passes of tools/popbwt_bench.py's fused search on a VALID
# population BWT (8 shards built on the spot).  The raw traces are large (the builder launches thousands of kernels):
# each pass is condensed on the box into gpurun_out/popbwt_pmc_<tag>/summary_*.json (per-launch means of the search
# kernel's counters) and its raw directory removed.   usage: tools/popbwt_pmc.sh TAG [popbwt_bench args]
set -euo pipefail
tag="${1:-r04}"; shift || true
out="gpurun_out/popbwt_pmc_${tag}"
raw="/tmp/popbwt_pmc_raw_$$"
mkdir -p "$out" "$raw"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python3 tools/popbwt_bench.py --symbols-per-shard 3e9 --steps 5 $*"
condense() {  # $1 = raw dir of a pass, $2 = name
python3 - "$1" "$out/summary_$2.json" <<'PY'
import collections, csv, glob, json, sys
root, dst = sys.argv[1], sys.argv[2]
res = {}
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "search_lines_kernel<false, false, false, 0>" in r["Kernel_Name"] or "search_solo_kernel<false, false, false, false, false, false>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = sorted(v)
        res[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v), "median": v[len(v) // 2], "max": v[-1]}
for f in glob.glob(root + "/**/*_kernel_stats.csv", recursive=True):
    res["kernel_stats"] = [r for r in csv.DictReader(open(f)) if "search_" in r["Name"] or "pack_" in r["Name"]][:12]
json.dump(res, open(dst, "w"), indent=1)
print(dst, {k: (v if k == "kernel_stats" else round(v["median"])) for k, v in res.items() if k != "kernel_stats"})
PY
}
timeout -k 10 500 rocprofv3 --kernel-trace --stats -f csv -d "$raw/stats" -o p -- $B > "$out/stats.log" 2>&1 || { tail -5 "$out/stats.log"; exit 1; }
condense "$raw/stats" stats; rm -rf "$raw/stats"
for c in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_HIT_sum" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  n="$(echo $c | cut -d' ' -f1)"
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c -f csv -d "$raw/$n" -o p -- $B > "$out/pmc_$n.log" 2>&1 || { tail -5 "$out/pmc_$n.log"; exit 1; }
  condense "$raw/$n" "$n"; rm -rf "$raw/$n"
done
tail -c 3000 "$out/stats.log" > "$out/bench_line.txt"; rm -f "$out"/*.log
echo "popbwt_pmc $tag done"
