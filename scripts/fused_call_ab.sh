#!/usr/bin/env bash
# Development aid (GPU box): gfm_graph_score as a whole (bench.py's extract block: fused_ms, emit_ms, graph_score_kernel alone) under
# environment variants, side by side on one box.   scripts/fused_call_ab.sh <tag> "VAR=VAL ..." "VAR=VAL ..." ...   ("-" = no variable)
root="$GRAFT_REPO_ROOT"; tag="$1"; shift
out="$root/gpurun_out/$tag"; mkdir -p "$out"
for spec in "$@"; do
  ( [ "$spec" != "-" ] && export $spec
    timeout 300 python3 "$root/scripts/extract_bench.py" 2> "$out/err.txt" | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
print(f'{sys.argv[1]:40s} fused_ms {d[\"fused_ms\"]:.4f}  graph_score_kernel {r.get(\"kernel_us\", 0):.1f} us  extract_plus_score_ms {d[\"extract_plus_score_ms\"]:.3f}  emit_ms {d[\"emit_ms\"]:.3f} (host enqueue {d[\"emit_host_enqueue_ms\"]:.3f})')
" "$spec" )
done 2>&1 | tee "$out/summary.txt"
