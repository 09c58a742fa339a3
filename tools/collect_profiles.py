#!/usr/bin/env python3
"""Condenses gpurun_out/profile_<tag>/ (tools/profile_round.sh) into the tracked profiles/ files:
  profiles/<tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_pmc_search_kernel.json  mean per-launch counters of the search kernel
  profiles/pmc_traffic.json              HBM bytes per search launch for bench.py's roofline.traffic
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KB, collected
in separate passes; on gfx950 FETCH_SIZE tallies a 128-B read request at 64 B, so it is doubled
(calibrated on this access pattern with tools/gather_bench: 128-B lines read 0.50x, see
profiles/r01_gather_microbench.txt)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
R = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20000000000
Q = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10000000
S = int(sys.argv[4]) if len(sys.argv) > 4 else 8
STREAM = sys.argv[5] if len(sys.argv) > 5 else "pop"
MIX = sys.argv[6] if len(sys.argv) > 6 else "population"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_source_sha: the figure is only valid for the sources it was measured on)
src = os.path.join(root, "gpurun_out", f"profile_{tag}")
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "stats", "p_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
# the kernel the fused query launches ran on: one lane per search behind deep tables (round 5: also with several shards
# per launch), lane pairs otherwise -- whichever the trace shows
_names = open(os.path.join(src, "stats", "p_kernel_stats.csv")).read()
KERNEL = ("search_solo_kernel<false, false, false, false, false, false>" if "search_solo_kernel<false, false, false, false, false, false>" in _names
          else "search_lines_kernel<false, false, false, 0>")
counters = {}
for f in glob.glob(os.path.join(src, "pmc_*", "p_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        counters[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
json.dump({"command": "rocprofv3 --kernel-trace --pmc <one counter set per pass> -f csv -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-single-check --no-second-mix",
           "kernel": "rsb::" + KERNEL + " (the bench's fused query launches; the k-mer tables' own launches run as <..., 1>)",
           "counters": counters},
          open(os.path.join(dst, f"{tag}_pmc_search_kernel.json"), "w"), indent=1)
fetch_kb = counters["FETCH_SIZE"]["mean_per_launch"]
write_kb = counters["WRITE_SIZE"]["mean_per_launch"]
traffic = (2 * fetch_kb + write_kb) * 1024
json.dump({"kernel_source_sha": bench.kernel_source_sha(), "run_bytes_per_shard": R, "queries_per_batch": Q,
           "shards_per_gpu": S, "k": 31, "stream": STREAM, "mix": MIX, "hbm_bytes_per_launch": traffic,
           "from": f"profiles/{tag}_pmc_search_kernel.json", "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
           "rule": "(2*FETCH_SIZE + WRITE_SIZE) KB * 1024; gfx950 FETCH_SIZE counts 128-B read requests at 64 B"},
          open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(f"{tag}: traffic {traffic / 1e9:.2f} GB per launch; wrote profiles/{tag}_*")
