#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<mode>_<tag>/ (tools/profile_mode.sh) into profiles/<tag>_<mode>_kernel_stats.csv and
profiles/<tag>_<mode>_pmc.json: per-launch means of every counter for the kernels whose name contains one of the
substrings given (default: the walk kernels and the search kernel).
Also profiles/pmc_traffic_modes.json[MODE]: HBM bytes per step of the kernels bench.py's roofline.kernel_ms covers in that mode
(and of every kernel of the step), from the FETCH_SIZE / WRITE_SIZE passes corrected as collect_profiles.py does, keyed by the
SHA of csrc/ and the sizes of the run -- bench.py reports it as roofline.traffic when they match.
usage: tools/collect_mode_profile.py TAG MODE [kernel substring ...]   (env STEPS: steps + warmup of the profiled run, default 4)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, mode = sys.argv[1], sys.argv[2]
subs = sys.argv[3:] or ["extract_prefix_wave_kernel", "extract_postfix_wave_kernel", "search_solo_kernel<false, false, false, true, false, true>",
                        "search_solo_kernel<false, false, false, false, true, false>"]
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

# ---- HBM traffic per step for bench.py's roofline.traffic
sys.path.insert(0, root)
import bench  # noqa: E402
steps = int(os.environ.get("STEPS", "4"))
# launches of the covered kernels in the profiled run: the walk kernels of --mode extract run once more in the bench's counting
# step under the same names (the 1-mismatch search kernels' counting instantiations have names of their own)
cov_launches = steps + (1 if mode == "extract" else 0)
covered = {"1mm": ["search_solo_kernel<false, false, false, true, false, true>",   # the walk of the k-mers (search_solo.h, WALK)
                   "search_solo_kernel<false, false, false, false, true, false>"],  # the worklist search (WL)
           "extract": ["extract_prefix_wave_kernel", "move_prefix", "extract_postfix_wave_kernel"]}[mode]
# the step's other kernels: they also run in the bench's one counting step (whose search kernels have names of their own)
others = {"1mm": ["wl_", "hit_", "pack_dense_kernel", "search_init_tiled_kernel"], "extract": []}[mode]
tot = {"FETCH_SIZE": [0.0, 0.0], "WRITE_SIZE": [0.0, 0.0]}
for f in glob.glob(os.path.join(src, "pmc_*", "p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in tot:
            if any(c in r["Kernel_Name"] for c in covered):
                tot[r["Counter_Name"]][0] += float(r["Counter_Value"])
            if any(c in r["Kernel_Name"] for c in others):
                tot[r["Counter_Name"]][1] += float(r["Counter_Value"])
p_modes = os.path.join(dst, "pmc_traffic_modes.json")
modes = json.load(open(p_modes)) if os.path.exists(p_modes) else {}
args = dict(a.split("=", 1) for a in os.environ.get("RUN_ARGS", "").split() if "=" in a)
modes[mode] = {"source_sha": bench.mode_source_sha(), "run_bytes_per_shard": int(float(args.get("runs", 2e10))),
               "shards_per_gpu": int(args.get("shards", 8)), "units_per_batch": int(float(args.get("units", 4e5 if mode == "1mm" else 2e6))),
               "steps_profiled": steps,
               "hbm_bytes_per_step_covered_kernels": (2 * tot["FETCH_SIZE"][0] + tot["WRITE_SIZE"][0]) * 1024 / cov_launches,
               "hbm_bytes_per_step_all_kernels": (2 * tot["FETCH_SIZE"][0] + tot["WRITE_SIZE"][0]) * 1024 / cov_launches
                                                 + (2 * tot["FETCH_SIZE"][1] + tot["WRITE_SIZE"][1]) * 1024 / (steps + (1 if mode == "1mm" else 0)),
               "covered_kernels": covered, "from": f"profiles/{tag}_{mode}_pmc.json",
               "rule": "(2*FETCH_SIZE + WRITE_SIZE) KB * 1024 summed over the kernels' dispatches / steps; gfx950 FETCH_SIZE counts 128-B read requests at 64 B"}
json.dump(modes, open(p_modes, "w"), indent=1)
print("pmc_traffic_modes:", json.dumps(modes[mode], indent=1))
