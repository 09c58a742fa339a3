#!/bin/bash
# usage: tools/pmc_ab.sh TAG LIB...: the same PMC passes for several builds of librsbwt.so (A/B
# comparisons of kernel variants); writes gpurun_out/pmc_ab_<TAG>.txt
set -uo pipefail
tag="$1"; shift
out="$PWD/gpurun_out/pmc_ab_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python3 bench.py --steps 3 --warmup 1 --cpu-sample 0"
i=0
for lib in "$@"; do
  export RSBWT_LIB="$lib"
  for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1)); d="$out/p$i"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -f csv -d "$d" -o p -- $B > "$d.log" 2>&1 || { echo "FAILED: $lib: $c" >> "$out.txt"; continue; }
    python3 - "$d/p_counter_collection.csv" "$lib" >> "$out.txt" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "search_lines_kernel<false, false, false, 0>" in r["Kernel_Name"] or "search_solo_kernel<false, false, false, false, false, false>" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{sys.argv[2].split('/')[-1]:16s} {k:36s} {sum(v)/len(v):.6g}")
PY
  done
done
cat "$out.txt"
