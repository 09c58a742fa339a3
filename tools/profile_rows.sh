#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of tools/bench_rows.py
# (the f2 / f3 rows) under gpurun_out/prof_rows_<tag>/.   usage: tools/profile_rows.sh TAG [bench_rows args]
set -euo pipefail
tag="${1:-r02}"; shift || true
args="${*:-2e10 40000 2000000 1}"
out="gpurun_out/prof_rows_${tag}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d "$out/stats" -o p -- python3 tools/bench_rows.py $args > "$out/stats.log" 2>&1 || { tail -5 "$out/stats.log"; exit 1; }
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
         "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  d="$out/pmc_$(echo $c | cut -d' ' -f1)"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -f csv -d "$d" -o p -- python3 tools/bench_rows.py $args > "$d.log" 2>&1 || { tail -5 "$d.log"; exit 1; }
done
find "$out" -name "*kernel_stats.csv" | head -3
echo "profile_rows $tag done"
