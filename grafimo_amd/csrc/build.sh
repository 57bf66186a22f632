#!/usr/bin/env bash
# Builds libgrafimo_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# -ffp-contract=off: the p-value DP must round the product before the add (no FMA).
# The translation units compile side by side (JOBS, default 8).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fvisibility=hidden
       -Wall -Wextra -Wno-unused-parameter -I"$root/include")
pids=()
cc() { "$HIPCC" "${FLAGS[@]}" "$@" & pids+=($!); }
cc -c "$here/grafimo_hip.hip" -o "$here/grafimo_hip.o"
quad=()
for gm in 0:1 1:1 2:1 3:1 0:2 1:2 0:3 1:3; do
    cc -Wno-unused-function -DGFM_QUAD_GROUP=${gm%:*} -DGFM_QUAD_MM=${gm#*:} -c "$here/score_quad_tu.hip" \
        -o "$here/score_quad_g${gm%:*}_m${gm#*:}.o"
    quad+=("$here/score_quad_g${gm%:*}_m${gm#*:}.o")
done
cc -c "$here/graph_extract.hip" -o "$here/graph_extract.o"
cc -c "$here/stream_calib.hip" -o "$here/stream_calib.o"
cc -c "$here/region_reduce.hip" -o "$here/region_reduce.o"
cc -c "$here/tsv_ingest.cpp" -o "$here/tsv_ingest.o"
cc -c "$here/vcf_ingest.cpp" -o "$here/vcf_ingest.o"
cc -c "$here/scan_stream.cpp" -o "$here/scan_stream.o"
cc -c "$here/gfm_workers.cpp" -o "$here/gfm_workers.o"
cc -c "$here/graph_tsv_writer.cpp" -o "$here/graph_tsv_writer.o"
cc -c "$here/hit_table.cpp" -o "$here/hit_table.o"
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$here/libgrafimo_hip.so" \
    "$here/grafimo_hip.o" "${quad[@]}" "$here/graph_extract.o" "$here/stream_calib.o" "$here/region_reduce.o" "$here/tsv_ingest.o" \
    "$here/vcf_ingest.o" "$here/scan_stream.o" "$here/gfm_workers.o" "$here/graph_tsv_writer.o" "$here/hit_table.o" -lpthread -lz
echo "built $here/libgrafimo_hip.so"
