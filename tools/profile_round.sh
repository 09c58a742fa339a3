#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of the
# default bench.py workload, written under gpurun_out/profile_<tag>/ (copy what should be judged
# into profiles/ afterwards with tools/collect_profiles.py).
set -euo pipefail
tag="${1:-r03}"
out="gpurun_out/profile_${tag}"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-single-check --no-second-mix ${BENCH_ARGS:-}"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d "$out/stats" -o p -- $B > "$out/stats.log" 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
         "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_READ_sum"; do
  d="$out/pmc_$(echo $c | cut -d' ' -f1)"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -f csv -d "$d" -o p -- $B > "$d.log" 2>&1 || exit 1
done
echo "profile_round $tag done"
