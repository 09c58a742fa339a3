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
                        "search_solo_kernel<false, false, false, false, true, false>", "wl_table_entries_kernel"]
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
# Every covered kernel runs ONCE per step: its bytes per step = its bytes over the profiled run / its own number of
# dispatches there (the walk kernels of --mode extract and the 1-mismatch table pre-pass also run in the bench's one
# counting step under the same names; the 1-mismatch search kernels' counting instantiations have names of their own).
covered = {"1mm": ["search_solo_kernel<false, false, false, true, false, true>",   # the walk of the k-mers (search_solo.h, WALK)
                   "wl_table_entries_kernel",                                        # the table-part variants' entries read ahead (round 5)
                   "search_solo_kernel<false, false, false, false, true, false>"],  # the worklist search (WL)
           "extract": ["extract_prefix_wave_kernel", "move_prefix", "extract_postfix_wave_kernel"]}[mode]
# the step's other kernels
others = {"1mm": ["wl_init", "wl_own", "wl_branch", "hit_", "pack_dense_kernel", "search_init_tiled_kernel"], "extract": []}[mode]
per = {"FETCH_SIZE": collections.defaultdict(lambda: [0.0, 0]), "WRITE_SIZE": collections.defaultdict(lambda: [0.0, 0])}
for f in glob.glob(os.path.join(src, "pmc_*", "p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in per:
            for c in covered + others:
                if c in r["Kernel_Name"]:
                    per[r["Counter_Name"]][c][0] += float(r["Counter_Value"])
                    per[r["Counter_Name"]][c][1] += 1
                    break
def per_step(names):
    b = 0.0
    for c in names:
        f, w = per["FETCH_SIZE"][c], per["WRITE_SIZE"][c]
        if f[1]:
            b += 2 * f[0] / f[1] * 1024
        if w[1]:
            b += w[0] / w[1] * 1024
    return b
tot_cov, tot_oth = per_step(covered), per_step(others)
p_modes = os.path.join(dst, "pmc_traffic_modes.json")
modes = json.load(open(p_modes)) if os.path.exists(p_modes) else {}
args = dict(a.split("=", 1) for a in os.environ.get("RUN_ARGS", "").split() if "=" in a)
modes[mode] = {"source_sha": bench.mode_source_sha(), "run_bytes_per_shard": int(float(args.get("runs", 2e10))),
               "shards_per_gpu": int(args.get("shards", 8)), "units_per_batch": int(float(args.get("units", 4e5 if mode == "1mm" else 2e6))),
               "steps_profiled": steps,
               "hbm_bytes_per_step_covered_kernels": tot_cov,
               "hbm_bytes_per_step_all_kernels": tot_cov + tot_oth,
               "per_kernel_bytes_per_launch": {c: (2 * per["FETCH_SIZE"][c][0] / max(per["FETCH_SIZE"][c][1], 1) + per["WRITE_SIZE"][c][0] / max(per["WRITE_SIZE"][c][1], 1)) * 1024
                                               for c in covered + others if per["FETCH_SIZE"][c][1] or per["WRITE_SIZE"][c][1]},
               "covered_kernels": covered, "from": f"profiles/{tag}_{mode}_pmc.json",
               "rule": "(2*FETCH_SIZE + WRITE_SIZE) KB * 1024 summed over the kernels' dispatches / steps; gfx950 FETCH_SIZE counts 128-B read requests at 64 B"}
json.dump(modes, open(p_modes, "w"), indent=1)
print("pmc_traffic_modes:", json.dumps(modes[mode], indent=1))
