#!/usr/bin/env bash
# per-kernel times of the fused extraction -> scoring path:  scripts/fused_prof.sh <tag>   (gpurun_out/<tag>/)
set -u
tag="${1:-r04prof}"
root="$GRAFT_REPO_ROOT"
out="$root/gpurun_out/$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 "$root/scripts/fused_prof.py" > "$out/fused_prof.log" 2>&1
f=$(ls -t "$out"/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$out/fused_kernel_stats.csv"
rm -rf "$out/prof"
cat "$out/fused_kernel_stats.csv" | cut -c1-200
