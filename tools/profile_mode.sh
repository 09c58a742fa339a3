#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of `bench.py --mode MODE` (1mm | extract)
# under gpurun_out/prof_MODE_<tag>/.   usage: tools/profile_mode.sh TAG MODE [bench args]
set -euo pipefail
tag="${1:-r03}"; mode="${2:-extract}"; shift 2 || true
out="gpurun_out/prof_${mode}_${tag}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python3 bench.py --mode $mode --steps 3 --warmup 1 $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d "$out/stats" -o p -- $B > "$out/stats.log" 2>&1 || { tail -5 "$out/stats.log"; exit 1; }
for c in "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  d="$out/pmc_$(echo $c | cut -d' ' -f1)"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -f csv -d "$d" -o p -- $B > "$d.log" 2>&1 || { tail -5 "$d.log"; exit 1; }
done
echo "profile_mode $tag $mode done"
