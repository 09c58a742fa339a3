#!/bin/bash
# Builds readserver_amd/lib/librsbwt.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
src="$here/csrc"
mkdir -p "$here/lib"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
exec "$HIPCC" -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-function \
  -I"$here/../include" \
  "$src/kernels.hip" "$src/search_wave.hip" "$src/slots.hip" "$src/index.hip" "$src/capi.hip" "$src/bwt_file.cpp" "$src/synth.cpp" "$src/service_slice.cpp" \
  -o "$here/lib/librsbwt.so" "$@"
