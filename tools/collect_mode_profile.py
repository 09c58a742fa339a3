#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<mode>_<tag>/ (tools/profile_mode.sh) into profiles/<tag>_<mode>_kernel_stats.csv and
profiles/<tag>_<mode>_pmc.json: per-launch means of every counter for the kernels whose name contains one of the
substrings given (default: the walk kernels and the search kernel).
usage: tools/collect_mode_profile.py TAG MODE [kernel substring ...]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, mode = sys.argv[1], sys.argv[2]
subs = sys.argv[3:] or ["extract_prefix_wave_kernel", "extract_postfix_wave_kernel", "search_lines_kernel<false, false, false, 0>", "search_solo_kernel"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{mode}_{tag}")
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "stats", "p_kernel_stats.csv"), os.path.join(dst, f"{tag}_{mode}_kernel_stats.csv"))
out = {}
for f in glob.glob(os.path.join(src, "pmc_*", "p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for s in subs:
            if s in r["Kernel_Name"]:
                agg[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for s, cs in agg.items():
        for k, v in cs.items():
            out.setdefault(s, {})[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
json.dump({"command": f"rocprofv3 --kernel-trace --pmc <one counter set per pass> -f csv -- python3 bench.py --mode {mode} --steps 3 --warmup 1 ...",
           "kernels": out}, open(os.path.join(dst, f"{tag}_{mode}_pmc.json"), "w"), indent=1)
print(json.dumps({s: {k: round(v["mean_per_launch"]) for k, v in cs.items()} for s, cs in out.items()}, indent=1))
