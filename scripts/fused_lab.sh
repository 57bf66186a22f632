#!/usr/bin/env bash
# what the deletion kernel of the fused path spends its time on: the same profile with parts switched off
root="$GRAFT_REPO_ROOT"; out="$root/gpurun_out/${1:-r04lab}"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for dbg in ${FUSED_LAB_MODES:-0 1 2 4}; do
  export GRAFIMO_FUSED_DEBUG=$dbg
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p$dbg" -- python3 "$root/scripts/fused_prof.py" > "$out/log$dbg.txt" 2>&1
  f=$(ls -t "$out"/p$dbg/*/*kernel_stats.csv 2>/dev/null | head -1)
  python3 - "$f" "$dbg" <<'PY'
import csv, sys
rows = {r["Name"].split("::")[1].split("(")[0]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "graph_" in r["Name"]}
print(f"debug={sys.argv[2]}: " + ", ".join(f"{k} {v:.1f} us" for k, v in rows.items() if k.startswith("graph_") and "count_kernel" != k[-12:] or k.startswith("graph_del")))
PY
  rm -rf "$out/p$dbg"
done
