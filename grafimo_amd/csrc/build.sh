#!/usr/bin/env bash
# Builds libgrafimo_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# -ffp-contract=off: the p-value DP must round the product before the add (no FMA).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fvisibility=hidden
       -Wall -Wextra -Wno-unused-parameter -I"$root/include")
"$HIPCC" "${FLAGS[@]}" -c "$here/grafimo_hip.hip" -o "$here/grafimo_hip.o" "$@"
"$HIPCC" "${FLAGS[@]}" -c "$here/graph_extract.hip" -o "$here/graph_extract.o"
"$HIPCC" "${FLAGS[@]}" -c "$here/tsv_ingest.cpp" -o "$here/tsv_ingest.o"
"$HIPCC" "${FLAGS[@]}" -c "$here/vcf_ingest.cpp" -o "$here/vcf_ingest.o"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$here/libgrafimo_hip.so" \
    "$here/grafimo_hip.o" "$here/graph_extract.o" "$here/tsv_ingest.o" "$here/vcf_ingest.o" -lpthread -lz
echo "built $here/libgrafimo_hip.so"
