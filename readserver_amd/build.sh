#!/bin/bash
# Builds readserver_amd/lib/librsbwt.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
src="$here/csrc"
mkdir -p "$here/lib"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-function \
  -I"$here/../include" \
  "$src/kernels.hip" "$src/search_lines.hip" "$src/extract_lines.hip" "$src/mm1_worklist.hip" "$src/build_lines.hip" "$src/capi.hip" "$src/sets.hip" "$src/bwt_file.cpp" "$src/bpi2.cpp" "$src/synth.cpp" "$src/service_slice.cpp" "$src/service_loop.cpp" "$src/layout_host.cpp" \
  -ldl -lpthread -o "$here/lib/librsbwt.so" "$@"
# index_rlebwt: the twin of the reference's src/util/index_rlebwt.cpp (writes "<bwt>.bpi2"); host code only
"${CXX:-g++}" -O2 -std=c++17 -Wall -I"$here/../include" "$src/index_rlebwt_main.cpp" "$src/bpi2.cpp" "$src/bwt_file.cpp" \
  -o "$here/lib/index_rlebwt"
# rsbwt_service: the twin of the reference's `service` process for the count path (libzmq is bound at run
# time by librsbwt.so: nothing to add here on a box that has it)
"${CXX:-g++}" -O2 -std=c++17 -Wall -I"$here/../include" "$src/service_main.cpp" -L"$here/lib" -lrsbwt \
  -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -o "$here/lib/rsbwt_service"
