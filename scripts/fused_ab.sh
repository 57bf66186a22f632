#!/usr/bin/env bash
# Development aid (GPU box): graph_score_kernel's time under variant builds of the library, side by side on one box.
#   scripts/fused_ab.sh <tag> <lib-or-"product">[:ENV=VAL,...] ...
# e.g.  scripts/fused_ab.sh ab1 product lab/libgfm_w6.so product:GRAFIMO_FUSED_WAVES=12
root="$GRAFT_REPO_ROOT"; tag="$1"; shift
out="$root/gpurun_out/$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  lib="${spec%%:*}"; envs=""; [[ "$spec" == *:* ]] && envs="${spec#*:}"
  unset GRAFIMO_HIP_LIB GRAFIMO_FUSED_WAVES
  [ "$lib" != "product" ] && export GRAFIMO_HIP_LIB="$root/$lib"
  IFS=',' read -ra kv <<< "$envs"; for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
  name=$(echo "$spec" | tr '/:=,' '____')
  timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p_$name" -- python3 "$root/scripts/fused_prof.py" > "$out/log_$name.txt" 2>&1
  f=$(ls -t "$out"/p_$name/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" "$spec" <<'PY'
import csv, sys
try:
    rows = {r["Name"].split("namespace)::")[1].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "graph_" in r["Name"]}
    print(f"{sys.argv[2]:50s} " + ", ".join(f"{k} {v:.1f}" for k, v in rows.items() if k.startswith(("graph_score_kernel", "graph_del_score", "graph_heavy"))))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  rm -rf "$out/p_$name"
done 2>&1 | tee "$out/summary.txt"
