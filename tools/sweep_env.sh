#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ... [-- extra bench args]: runs the bench once per value
var="$1"; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
  env "$var=$v" timeout -k 10 300 python bench.py --cpu-sample 0 "$@" 2>/dev/null | tail -1 > /tmp/sweep.json
  echo "$var=$v $(python tools/bench_brief.py /tmp/sweep.json)"
done
