#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ... [-- extra bench args]: runs the bench once per value.
# Every run keeps its stderr (gpurun_out/sweep_<VAR>_<value>.err); a run that fails is reported as
# FAILED with its exit status instead of a stale or empty line being summarised.
set -uo pipefail
var="$1"; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "${1:-}" == "--" ] && shift
mkdir -p gpurun_out
for v in "${vals[@]}"; do
  out="gpurun_out/sweep_${var}_${v}"
  rm -f "$out.json"
  if env "$var=$v" timeout -k 10 600 python bench.py --cpu-sample 0 "$@" > "$out.json" 2> "$out.err"; then
    echo "$var=$v $(python tools/bench_brief.py "$out.json")"
  else
    rc=$?
    echo "$var=$v FAILED (exit $rc): $(tail -1 "$out.err")"
    [ $rc -ge 124 ] && { echo "stopping the sweep: a run was killed"; exit $rc; }
  fi
done
