#!/bin/bash
# usage: tools/build_variant.sh NAME [hipcc flags...]: builds tools/bin/librsbwt_NAME.so with extra flags
# (e.g. -DRSB_MIN_WGS_PER_CU=5) for A/B runs: RSBWT_LIB=tools/bin/librsbwt_NAME.so python bench.py ...
set -euo pipefail
name="$1"; shift
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$here/readserver_amd/csrc"
mkdir -p "$here/tools/bin"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-function -I"$here/include" \
  "$src/kernels.hip" "$src/search_lines.hip" "$src/extract_lines.hip" "$src/mm1_worklist.hip" "$src/build_lines.hip" "$src/capi.hip" "$src/sets.hip" "$src/bwt_file.cpp" \
  "$src/bpi2.cpp" "$src/synth.cpp" "$src/service_slice.cpp" "$src/service_loop.cpp" "$src/layout_host.cpp" \
  -ldl -lpthread "$@" -o "$here/tools/bin/librsbwt_$name.so"
echo "built tools/bin/librsbwt_$name.so"
