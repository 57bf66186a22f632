#!/usr/bin/env bash
# Development aid: builds one-width variants of libgrafimo_hip.so into lab/ (seconds each) so that
# kernel experiments can be timed side by side on one GPU box:
#   scripts/lab_build.sh <tag> [-DGFM_...]...        ->  lab/libgfm_<tag>.so   (width: $LAB_W, default 19)
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
tag="$1"; shift
W="${LAB_W:-19}"
src="${LAB_SRC:-$root/grafimo_amd/csrc}"     # LAB_SRC: another copy of csrc/ (e.g. git archive of HEAD) for A/B runs
obj="$root/grafimo_amd/csrc"
mkdir -p "$root/lab"
CC=(/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fvisibility=hidden
    -I"$root/include" -DGFM_ONLY_W="$W" "$@")
"${CC[@]}" -c "$src/grafimo_hip.hip" -o "$root/lab/gfm_$tag.o" &
# the graph kernels only when a flag concerns them (-DGFM_LAB: the fused kernels' timers and "parts switched off" bits, which
# the product library does not contain): a minute of compile time
gx="$obj/graph_extract.o"
if [[ " $* " == *" -DGFM_LAB "* || " $* " == *"-DGFM_GRAPH_"* ]]; then
    gx="$root/lab/gfm_${tag}_graph.o"
    "${CC[@]}" -c "$src/graph_extract.hip" -o "$gx" &
fi
for gm in 0:1 1:1 2:1 3:1 0:2 1:2 0:3 1:3; do
    "${CC[@]}" -DGFM_QUAD_GROUP=${gm%:*} -DGFM_QUAD_MM=${gm#*:} -c "$src/score_quad_tu.hip" \
        -o "$root/lab/gfm_${tag}_g${gm%:*}_m${gm#*:}.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/lab/libgfm_$tag.so" "$root/lab/gfm_$tag.o" \
    "$root/lab"/gfm_${tag}_g?_m?.o "$gx" "$obj/stream_calib.o" "$obj/region_reduce.o" "$obj/tsv_ingest.o" "$obj/vcf_ingest.o" \
    "$obj/scan_stream.o" "$obj/gfm_workers.o" "$obj/graph_tsv_writer.o" "$obj/hit_table.o" -lpthread -lz
rm -f "$root/lab/gfm_$tag.o" "$root/lab"/gfm_${tag}_g?_m?.o "$root/lab/gfm_${tag}_graph.o"
echo "built lab/libgfm_$tag.so"
